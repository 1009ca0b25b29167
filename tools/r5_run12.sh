#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_12; mkdir -p $O
python -m pytest tests/test_f16f6_gpu.py tests/test_c2_gpu.py tests/test_range_guard_gpu.py tests/test_parity_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "f16f6 tests (staging off) rc=$?" >> $O/summary.txt
python -m pytest tests/test_bf16_io_gpu.py tests/test_fusions_gpu.py -q -m gpu > $O/tests_pools.log 2>&1; echo "pool tests rc=$?" >> $O/summary.txt
timeout 300 python tools/bench_pools.py 30 2>/dev/null | grep kernel > $O/hbm_kernels.jsonl
( echo "# rank-net shape (512 x 801024 x 512)"; python tools/tune_f16f6_planes.py run 4; echo "# Tucker shape (512 x 801024 x 300)"; CTI_TUNE_K=300 python tools/tune_f16f6_planes.py run 4 ) > $O/aside_hstage_ab.txt 2>&1
cat $O/summary.txt; tail -4 $O/tests.log; tail -4 $O/tests_pools.log; grep -E "pool_shift|bi_logits" $O/hbm_kernels.jsonl | cut -c1-160; grep -v amdgpu $O/aside_hstage_ab.txt
