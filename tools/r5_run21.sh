#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_21; mkdir -p $O
python -m pytest tests/test_unrolled_ops_gpu.py tests/test_bf16_io_gpu.py tests/test_fusions_gpu.py tests/test_parity_gpu.py tests/test_edge_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
timeout 300 python tools/bench_pools.py 30 2>/dev/null | grep kernel > $O/hbm_kernels.jsonl
for i in 1 2; do python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_$i.json; CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_serial_$i.json; done
cat $O/summary.txt; tail -2 $O/tests.log; grep -E "bi_pool" $O/hbm_kernels.jsonl | cut -c1-170
for f in $O/bench_c*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4))"; done
