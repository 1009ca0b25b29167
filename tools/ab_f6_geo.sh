# needs a library built with -DCTI_F6_WITH_HALF_GEO (CTI_HIP_LIB=<that .so>): the product build has no half geometry since round 5
cd $GRAFT_REPO_ROOT
for g in default half; do
  if [ $g = half ]; then export CTI_F6_GEO=half; else unset CTI_F6_GEO; fi
  python bench.py --no-cpu-baseline --no-fp32-exact 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$g', round(d['value']), round(d['ms_per_step'],3), [(k['kernel'][:60], round(k['ms'],3)) for k in d['roofline_kernels']['kernels']])"
done
