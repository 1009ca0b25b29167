#!/bin/bash
# round 5, GPU call 8: unrolled BAN loop v3 (K-concatenated products: <= 17 addends per pool), a-side epilogue store ablation (coalesced H stores)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_8; mkdir -p $O
python -m pytest tests/test_fusions_gpu.py tests/test_models_gpu.py -q -s -m gpu -k "unrolled or hoisted or replay or graph or full_batch" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
for i in 1 2; do
python bench.py --config c4 > $O/bench_c4_$i.json 2> $O/bench_c4.err; echo "bench c4 rc=$?" >> $O/summary.txt
CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 > $O/bench_c4_serial_$i.json 2>/dev/null
done
( echo "# rank-net shape (512 x 801024 x 512)"; python tools/tune_f16f6_planes.py run 4; echo "# Tucker shape (512 x 801024 x 300)"; CTI_TUNE_K=300 python tools/tune_f16f6_planes.py run 4 ) > $O/aside_epilogue_ablation.txt 2>&1
bash tools/trace_models.sh
python tools/print_forward_timeline.py $(ls gpurun_out/pc_c4/*kernel_trace.csv gpurun_out/pc_c4/*/*kernel_trace.csv 2>/dev/null | head -1) > $O/model_c4_timeline.txt 2>&1
find gpurun_out/pc_c3 gpurun_out/pc_c4 -name "*kernel_trace.csv" -delete
cat $O/summary.txt; grep -a "unrolled vs\|passed\|failed\|Error" $O/tests.log | tail -8
for f in $O/bench_c4_1.json $O/bench_c4_2.json $O/bench_c4_serial_1.json $O/bench_c4_serial_2.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), {k:(round(v,5) if isinstance(v,float) else v) for k,v in d['parity_of_timed_forward'].items() if k not in ('vs','rows','tol')})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat $O/aside_epilogue_ablation.txt | grep -v amdgpu
sed -n 28,60p $O/model_c4_timeline.txt | cut -c1-120
