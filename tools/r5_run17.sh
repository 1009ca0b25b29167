#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_17; mkdir -p $O
python tools/find_wn_callers.py > $O/wn_callers.txt 2>&1
python -m pytest tests/test_fusions_gpu.py tests/test_models_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
for i in 1 2; do
  python bench.py --config c4 2>$O/bench_c4.err | tail -1 > $O/bench_c4_$i.json
  CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_serial_$i.json
done
cat $O/summary.txt; tail -2 $O/tests.log; grep -v amdgpu $O/wn_callers.txt | tail -20
for f in $O/bench_c*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4))"; done
