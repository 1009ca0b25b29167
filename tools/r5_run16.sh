#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_16; mkdir -p $O
python -m pytest tests/test_fusions_gpu.py -q -m gpu -s -k "unrolled or hoisted" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
bash tools/prof_models.sh > $O/prof_models.log 2>&1; cp gpurun_out/pc_c3/summary.txt $O/model_c3_kernel_stats.txt; cp gpurun_out/pc_c4/summary.txt $O/model_c4_kernel_stats.txt
bash tools/trace_models.sh > $O/trace_models.log 2>&1
python tools/print_forward_timeline.py gpurun_out/pc_c3/m_kernel_trace.csv > $O/model_c3_timeline.txt 2>&1; python tools/print_forward_timeline.py gpurun_out/pc_c4/m_kernel_trace.csv > $O/model_c4_timeline.txt 2>&1
find gpurun_out/pc_c3 gpurun_out/pc_c4 -name "*kernel_trace.csv" -delete 2>/dev/null
cat $O/summary.txt; grep -E "passed|failed|after the update" $O/tests.log | tail -14; tail -1 $O/model_c3_timeline.txt; tail -1 $O/model_c4_timeline.txt
