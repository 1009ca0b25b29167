for rep in 1 2; do for lib in band0 "" band4; do
  if [ -n "$lib" ]; then export CTI_HIP_LIB=$GRAFT_REPO_ROOT/iccv19_vqa-cti_amd/lib/variants/libcti_hip_$lib.so; else unset CTI_HIP_LIB; fi
  python tools/bench_gemm16.py 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('${lib:-band8(shipped)}', {k: (v['rows_bf16'][0], v['rows_bf16_whole_tiles'][0], v['vendor'][0]) for k,v in d['shapes'].items()})" 2>&1 | cut -c1-400
done; done
