# duration of the M build ALONE on the chip per library variant (CTI_F6_JOIN=r: the main stream joins in front of the rank nets' product): bash tools/ab_kq.sh <variant> ...
for v in "$@"; do
  CTI_HIP_LIB=$GRAFT_REPO_ROOT/iccv19_vqa-cti_amd/lib/variants/libcti_hip_$v.so CTI_F6_JOIN=r bash tools/trace_step.sh > /dev/null 2>&1
  echo "$v: $(grep 'mbuild' gpurun_out/step_trace/timeline.txt | cut -c1-60)"
done
