#!/bin/bash
# GRU step with the h operand loaded straight into registers (CTI_GRU_DIRECTA=0 = the LDS-ring form of rounds 3-4): parity, then c3 / c4 on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_22; mkdir -p $O
python -m pytest tests/test_models_gpu.py tests/test_parity_gpu.py tests/test_edge_gpu.py tests/test_backward_gpu.py -q -m gpu -k "gru or GRU or model or embedding or question or ffoe or mc_" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
for i in 1 2; do
  python bench.py --config c3 2>/dev/null | tail -1 > $O/bench_c3_da_$i.json
  CTI_GRU_DIRECTA=0 python bench.py --config c3 2>/dev/null | tail -1 > $O/bench_c3_ring_$i.json
  python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_da_$i.json
  CTI_GRU_DIRECTA=0 python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_ring_$i.json
done
CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_serial_da.json
CTI_GRU_DIRECTA=0 CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_serial_ring.json
cat $O/summary.txt; tail -3 $O/tests.log
for f in $O/bench_c*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4))"; done
