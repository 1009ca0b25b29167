export CTI_HIP_LIB=$GRAFT_REPO_ROOT/iccv19_vqa-cti_amd/lib/variants/libcti_hip_rsabl1.so
python3 tools/bench_gemm_small.py 3 2>&1 | tail -5 | cut -c1-400
