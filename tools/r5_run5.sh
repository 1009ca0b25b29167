#!/bin/bash
# round 5, GPU call 5: the unrolled BAN glimpse loop (tests + c4 bench, concurrent and serial, with the knob off beside it) + per-kernel stats of the models
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_5; mkdir -p $O
python -m pytest tests/test_fusions_gpu.py tests/test_models_gpu.py tests/test_bf16_io_gpu.py -q -s -m gpu -k "unrolled or hoisted or replay or graph or bf16 or full_batch" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
for i in 1 2; do
python bench.py --config c4 > $O/bench_c4_$i.json 2> $O/bench_c4.err; echo "bench c4 rc=$?" >> $O/summary.txt
CTI_NO_UNROLLED_LOOP=1 python bench.py --config c4 > $O/bench_c4_nounroll_$i.json 2>/dev/null
CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 > $O/bench_c4_serial_$i.json 2>/dev/null
CTI_NO_UNROLLED_LOOP=1 CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 > $O/bench_c4_serial_nounroll_$i.json 2>/dev/null
done
python bench.py --config c3 > $O/bench_c3.json 2>/dev/null
bash tools/prof_models.sh > $O/prof_models.log 2>&1; cp gpurun_out/pc_c4/summary.txt $O/model_c4_kernel_stats.txt; cp gpurun_out/pc_c3/summary.txt $O/model_c3_kernel_stats.txt
cat $O/summary.txt; grep -a "unrolled vs\|passed\|failed\|Error" $O/tests.log | tail -12
for f in $O/bench_c4_1.json $O/bench_c4_nounroll_1.json $O/bench_c4_2.json $O/bench_c4_nounroll_2.json $O/bench_c4_serial_1.json $O/bench_c4_serial_nounroll_1.json $O/bench_c4_serial_2.json $O/bench_c4_serial_nounroll_2.json $O/bench_c3.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), {k:(round(v,5) if isinstance(v,float) else v) for k,v in d['parity_of_timed_forward'].items() if k not in ('vs','rows','tol')})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
head -25 $O/model_c4_kernel_stats.txt | cut -c1-200
