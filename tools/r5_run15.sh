#!/bin/bash
# unrolled tri glimpse loop: parity + c3 / c4 A/B against the hoisted loop (CTI_NO_UNROLLED_LOOP=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_15; mkdir -p $O
python -m pytest tests/test_fusions_gpu.py tests/test_models_gpu.py tests/test_bf16_io_gpu.py -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
for i in 1 2; do
  python bench.py --config c3 2>$O/bench_c3.err | tail -1 > $O/bench_c3_$i.json
  CTI_NO_UNROLLED_LOOP=1 python bench.py --config c3 2>/dev/null | tail -1 > $O/bench_c3_nounroll_$i.json
  python bench.py --config c4 2>$O/bench_c4.err | tail -1 > $O/bench_c4_$i.json
  CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_serial_$i.json
done
cat $O/summary.txt; grep -E "passed|failed|unrolled vs" $O/tests.log | tail -12
for f in $O/bench_c*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4))"; done
