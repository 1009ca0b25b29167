export CTI_NO_AUX_STREAM=1
bash tools/trace_step.sh > /dev/null 2>&1; cat gpurun_out/step_trace/timeline.txt
