#!/bin/bash
# round 5, GPU call 4: bf16 activation path (kernels + models), model tests, c3 / c4 bench before/after, M-build ablations
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_4; mkdir -p $O
python -m pytest tests/test_bf16_io_gpu.py -q -s -m gpu > $O/bf16io.log 2>&1; echo "bf16io rc=$?" >> $O/summary.txt
python -m pytest tests/test_fusions_gpu.py tests/test_abi.py tests/test_models_gpu.py -q -m gpu > $O/fusions.log 2>&1; echo "fusions/models rc=$?" >> $O/summary.txt
for c in c3 c4; do
  python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$?" >> $O/summary.txt
  CTI_BENCH_V_FP32=1 python bench.py --config $c > $O/bench_${c}_v32.json 2> $O/bench_${c}_v32.err
done
CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 > $O/bench_c4_serial.json 2>/dev/null
python tools/tune_mbuild_f6.py run 4 > $O/mbuild_ablation.txt 2>&1
cat $O/summary.txt
tail -15 $O/bf16io.log; tail -8 $O/fusions.log
for f in $O/bench_c3.json $O/bench_c3_v32.json $O/bench_c4.json $O/bench_c4_v32.json $O/bench_c4_serial.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), d['parity_of_timed_forward'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat $O/mbuild_ablation.txt | tail -12; tail -3 $O/bench_c4.err
