#!/bin/bash
# in-step durations of the a-side products, staged vs direct H stores: two step traces + 5 alternating bench pairs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_14; mkdir -p $O
bash tools/trace_step.sh > /dev/null 2>&1; cp gpurun_out/step_trace/timeline.txt $O/step_timeline_stage.txt
CTI_HIP_LIB=$PWD/tools/variants/libcti_hip_nostage.so bash tools/trace_step.sh > /dev/null 2>&1; cp gpurun_out/step_trace/timeline.txt $O/step_timeline_direct.txt
for i in 1 2 3 4 5; do
  CTI_HIP_LIB=$PWD/tools/variants/libcti_hip_nostage.so python bench.py --no-cpu-baseline --no-fp32-exact --no-subrecords 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('direct  ms_per_step %.4f  samples/s %.0f' % (d['ms_per_step'], d['value']))" >> $O/step_ab.txt
  python bench.py --no-cpu-baseline --no-fp32-exact --no-subrecords 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stage   ms_per_step %.4f  samples/s %.0f' % (d['ms_per_step'], d['value']))" >> $O/step_ab.txt
done
for f in stage direct; do echo "== $f"; grep -E "gemm_f16f6_kernel|quantize_rows|mbuild|step length" $O/step_timeline_$f.txt | cut -c1-90; done; cat $O/step_ab.txt
