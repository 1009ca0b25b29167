#!/bin/bash
# staged H stores (fixed: predicated stores instead of the trash row) -- parity + kernel A/B + whole-step A/B; tri pool bf16 rows as dword loads
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_13; mkdir -p $O
python -m pytest tests/test_f16f6_gpu.py tests/test_c2_gpu.py tests/test_range_guard_gpu.py tests/test_parity_gpu.py tests/test_bf16_io_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "tests (staging on) rc=$?" >> $O/summary.txt
timeout 300 python tools/bench_pools.py 30 2>/dev/null | grep kernel > $O/hbm_kernels.jsonl
( echo "# rank-net shape (512 x 801024 x 512)"; python tools/tune_f16f6_planes.py run 4; echo "# Tucker shape (512 x 801024 x 300)"; CTI_TUNE_K=300 python tools/tune_f16f6_planes.py run 4 ) > $O/aside_hstage_ab.txt 2>&1
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-fp32-exact --no-subrecords 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stage   ms_per_step %.4f  samples/s %.0f' % (d['ms_per_step'], d['value']))" >> $O/step_ab.txt
  CTI_HIP_LIB=$PWD/tools/variants/libcti_hip_nostage.so python bench.py --no-cpu-baseline --no-fp32-exact --no-subrecords 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('direct  ms_per_step %.4f  samples/s %.0f' % (d['ms_per_step'], d['value']))" >> $O/step_ab.txt
done
cat $O/summary.txt; tail -4 $O/tests.log; grep -E "tri_pool_shift" $O/hbm_kernels.jsonl | cut -c1-120; grep -v amdgpu $O/aside_hstage_ab.txt; cat $O/step_ab.txt
