#!/bin/bash
# guard-on vs guard-off A/B of the configs[1] step on one box, alternating runs: bash tools/guard_ab.sh > gpurun_out/guard_ab.txt
for i in 1 2 3; do
  for g in 0 1; do
    echo -n "CTI_F6_GUARD_ABLATE=$g  "
    CTI_F6_GUARD_ABLATE=$g python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-subrecords --no-fp32-exact 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.4f  samples/s %.0f' % (d['ms_per_step'], d['value']))"
  done
done
