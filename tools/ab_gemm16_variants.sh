for v in "$@"; do echo "== $v"; CTI_HIP_LIB=iccv19_vqa-cti_amd/lib/variants/libcti_hip_$v.so python tools/bench_gemm16.py 20 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['shapes']
for k,v in d.items(): print(k, {a:b[0] for a,b in v.items() if a.startswith('rows_bf16') or a=='vendor'})
"; done
