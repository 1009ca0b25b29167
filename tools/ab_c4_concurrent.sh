#!/bin/bash
# c3 / c4 model forwards: serial vs sibling streams (c4), with the model-level auxiliary-stream sections on / off
cd $GRAFT_REPO_ROOT
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3), d.get('parity_of_timed_forward',{}).get('every_row_vs_bf16x3_forward'))"; }
for i in 1 2; do
CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 2>&1 | p "c4 serial"
python bench.py --config c4 2>&1 | p "c4 concurrent"
CTI_BENCH_C4_ORDER=ban_first python bench.py --config c4 2>&1 | p "c4 concurrent ban_first"
CTI_NO_AUX_STREAM=1 python bench.py --config c4 2>&1 | p "c4 concurrent no_aux"
python bench.py --config c3 2>&1 | p "c3"
CTI_NO_AUX_STREAM=1 python bench.py --config c3 2>&1 | p "c3 no_aux"
done
