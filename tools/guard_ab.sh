#!/bin/bash
# guard A/B of the configs[1] step on one box, alternating runs: bash tools/guard_ab.sh > gpurun_out/guard_ab.txt
#   late   = ONE scan behind the join (round-4 default)      early = the round-3 placement (three scans spread over both streams)      off = no guard kernels
run() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-subrecords --no-fp32-exact 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.4f  samples/s %.0f' % (d['ms_per_step'], d['value']))"; }
for i in 1 2 3; do
  echo -n "late   "; CTI_F6_GUARD_LATE=1 run
  echo -n "early  "; CTI_F6_GUARD_LATE=0 run
  echo -n "off    "; CTI_F6_GUARD_ABLATE=1 run
done
