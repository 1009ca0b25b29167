#!/bin/bash
# round 5, GPU call 9: a-side epilogue with LDS-staged H stores + bias in LDS: parity tests, kernel A/B, headline bench + timeline
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_9; mkdir -p $O
python -m pytest tests/test_f16f6_gpu.py tests/test_c2_gpu.py tests/test_range_guard_gpu.py tests/test_parity_gpu.py tests/test_abi.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
( echo "# rank-net shape (512 x 801024 x 512)"; python tools/tune_f16f6_planes.py run 4; echo "# Tucker shape (512 x 801024 x 300)"; CTI_TUNE_K=300 python tools/tune_f16f6_planes.py run 4 ) > $O/aside_hstage_ab.txt 2>&1
for i in 1 2; do python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-fp32-exact --no-subrecords > $O/bench_$i.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt; done
bash tools/trace_step.sh; cp gpurun_out/step_trace/timeline.txt $O/step_timeline.txt
cat $O/summary.txt; tail -4 $O/tests.log; grep -v amdgpu $O/aside_hstage_ab.txt
for i in 1 2; do python -c "
import json
d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), d['kernel_ms'])"; done
cat $O/step_timeline.txt
