#!/bin/bash
# round 5, GPU call 7: timeline of the c4 forward (serial models, eager) with the unrolled BAN loop; pool / addend microbench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_7; mkdir -p $O
bash tools/trace_models.sh
python tools/print_forward_timeline.py $(ls gpurun_out/pc_c4/*kernel_trace.csv gpurun_out/pc_c4/*/*kernel_trace.csv 2>/dev/null | head -1) > $O/model_c4_timeline.txt 2>&1
python tools/print_forward_timeline.py $(ls gpurun_out/pc_c3/*kernel_trace.csv gpurun_out/pc_c3/*/*kernel_trace.csv 2>/dev/null | head -1) > $O/model_c3_timeline.txt 2>&1
find gpurun_out/pc_c3 gpurun_out/pc_c4 -name "*kernel_trace.csv" -delete
python tools/bench_pools.py 30 > $O/hbm_kernels.jsonl 2> $O/hbm_kernels.err
cat $O/model_c4_timeline.txt | cut -c1-130
grep -a "round 5\|bi_pool\|rows_equal" $O/hbm_kernels.jsonl | cut -c1-250
