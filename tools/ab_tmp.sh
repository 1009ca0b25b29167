V=iccv19_vqa-cti_amd/lib/variants
for r in 1 2; do
for v in tp0 cu c4 cuc4; do echo "$v $(CTI_HIP_LIB=$PWD/$V/libcti_hip_$v.so python tools/bench_pools.py 30 2>/dev/null | grep 'tri_pool' | grep '"A": 3' | cut -c50-75)"; done; done
