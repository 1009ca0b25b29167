#!/bin/bash
# round 5, GPU call 6: unrolled BAN loop v2 (independent addend loads, own K split of the raw-partials products) + the whole GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_6; mkdir -p $O
for i in 1 2; do
python bench.py --config c4 > $O/bench_c4_$i.json 2> $O/bench_c4.err; echo "bench c4 rc=$?" >> $O/summary.txt
CTI_NO_UNROLLED_LOOP=1 python bench.py --config c4 > $O/bench_c4_nounroll_$i.json 2>/dev/null
CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 > $O/bench_c4_serial_$i.json 2>/dev/null
done
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
bash tools/prof_models.sh > $O/prof_models.log 2>&1; cp gpurun_out/pc_c4/summary.txt $O/model_c4_kernel_stats.txt; cp gpurun_out/pc_c3/summary.txt $O/model_c3_kernel_stats.txt
cat $O/summary.txt; tail -4 $O/pytest_gpu.log
for f in $O/bench_c4_1.json $O/bench_c4_nounroll_1.json $O/bench_c4_2.json $O/bench_c4_nounroll_2.json $O/bench_c4_serial_1.json $O/bench_c4_serial_2.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), {k:(round(v,5) if isinstance(v,float) else v) for k,v in d['parity_of_timed_forward'].items() if k not in ('vs','rows','tol')})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
head -16 $O/model_c4_kernel_stats.txt | cut -c1-200
