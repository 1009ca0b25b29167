# A/B of library variants on the headline step (one box): bash tools/ab_headline_variants.sh <variant> ...   (iccv19_vqa-cti_amd/lib/variants/libcti_hip_<variant>.so)
for v in "$@"; do
  CTI_HIP_LIB=iccv19_vqa-cti_amd/lib/variants/libcti_hip_$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-exact --no-subrecords 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['value']), 'samples/s', round(d['ms_per_step'],3), 'ms; mode-3', round(d['roofline']['launch_ms'],3), 'ms', d['kernel_ms'])"
done
