#!/bin/bash
# round 5, GPU call 3: guard tests at the calibrated thresholds, bf16 activation path (kernels + models), c3 / c4 bench before/after, replication tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_3; mkdir -p $O
python -m pytest tests/test_accuracy_envelope_gpu.py -q -s -m gpu > $O/envelope.log 2>&1; echo "envelope rc=$?" >> $O/summary.txt
python -m pytest tests/test_bf16_io_gpu.py -q -s -m gpu > $O/bf16io.log 2>&1; echo "bf16io rc=$?" >> $O/summary.txt
python -m pytest tests/test_fusions_gpu.py tests/test_abi.py tests/test_models_gpu.py -q -m gpu -x > $O/fusions.log 2>&1; echo "fusions/models rc=$?" >> $O/summary.txt
for c in c3 c4; do
  python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$?" >> $O/summary.txt
  CTI_BENCH_V_FP32=1 python bench.py --config $c > $O/bench_${c}_v32.json 2> $O/bench_${c}_v32.err
done
CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 > $O/bench_c4_serial.json 2>/dev/null
cat $O/summary.txt
grep -a "sweep\|localised\|poison mode\|cancelling eps\|passed\|failed" $O/envelope.log | tail -40
tail -15 $O/bf16io.log; tail -5 $O/fusions.log
for f in $O/bench_c3.json $O/bench_c3_v32.json $O/bench_c4.json $O/bench_c4_v32.json $O/bench_c4_serial.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), d['parity_of_timed_forward'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
