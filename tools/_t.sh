for rep in 1 2; do for cfg in c3 c4; do for lib in "" sksb2; do
  if [ -n "$lib" ]; then export CTI_HIP_LIB=$GRAFT_REPO_ROOT/iccv19_vqa-cti_amd/lib/variants/libcti_hip_$lib.so; else unset CTI_HIP_LIB; fi
  python bench.py --config $cfg --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg ${lib:-shipped}', round(d['value']), 'samples/s', round(d['ms_per_step']*1e3,1), 'us')"
done; done; done
