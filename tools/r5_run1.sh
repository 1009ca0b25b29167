#!/bin/bash
# round 5, GPU call 1: guard calibration (new stratified estimate + thresholds), replication fix, baseline headline + step timeline
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_1; mkdir -p $O
python -m pytest tests/test_accuracy_envelope_gpu.py -x -q -s -m gpu > $O/envelope.log 2>&1; echo "envelope rc=$?" >> $O/summary.txt
python -m pytest tests/test_range_guard_gpu.py tests/test_fusions_gpu.py tests/test_abi.py tests/test_c2_gpu.py -q -m gpu -k "guard or replicat or poison or private or c2 or range" > $O/guard_rep.log 2>&1; echo "guard_rep rc=$?" >> $O/summary.txt
python tools/rho_of_bench_inputs.py > $O/rho_bench.txt 2>&1
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-fp32-exact --no-subrecords > $O/bench_short.json 2> $O/bench_short.err; echo "bench rc=$?" >> $O/summary.txt
bash tools/trace_step.sh; cp gpurun_out/step_trace/timeline.txt $O/timeline.txt
tail -3 $O/envelope.log $O/guard_rep.log; cat $O/summary.txt; tail -c 600 $O/bench_short.json
