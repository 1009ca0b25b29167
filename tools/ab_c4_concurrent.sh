#!/bin/bash
# c3 / c4 model forwards A/B: hoisted glimpse loops on / off, sibling streams on / off
cd $GRAFT_REPO_ROOT
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3), d.get('parity_of_timed_forward'))"; }
for i in 1 2; do
python bench.py --config c4 2>&1 | p "c4"
CTI_NO_HOISTED_LOOP=1 python bench.py --config c4 2>&1 | p "c4 literal-loop"
CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 2>&1 | p "c4 serial"
python bench.py --config c3 2>&1 | p "c3"
CTI_NO_HOISTED_LOOP=1 python bench.py --config c3 2>&1 | p "c3 literal-loop"
done
