#!/bin/bash
# GRU step kernel with a 24-KiB ring in the plain-bf16 mode (co-resident with the projection GEMM's workgroups) vs the 64-KiB ring (tools/variants/libcti_hip_gru64k.so)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_24; mkdir -p $O
python -m pytest tests/test_models_gpu.py tests/test_parity_gpu.py tests/test_edge_gpu.py tests/test_fusions_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
OLD=$PWD/tools/variants/libcti_hip_gru64k.so
for i in 1 2; do
  python bench.py --config c3 2>/dev/null | tail -1 > $O/bench_c3_new_$i.json
  CTI_HIP_LIB=$OLD python bench.py --config c3 2>/dev/null | tail -1 > $O/bench_c3_old_$i.json
  python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_new_$i.json
  CTI_HIP_LIB=$OLD python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_old_$i.json
  CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_serial_new_$i.json
  CTI_HIP_LIB=$OLD CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_serial_old_$i.json
done
bash tools/trace_models.sh > $O/trace_models.log 2>&1
python tools/print_forward_timeline.py gpurun_out/pc_c4/m_kernel_trace.csv > $O/model_c4_timeline.txt 2>&1; python tools/print_forward_timeline.py gpurun_out/pc_c3/m_kernel_trace.csv > $O/model_c3_timeline.txt 2>&1
find gpurun_out/pc_c3 gpurun_out/pc_c4 -name "*kernel_trace.csv" -delete 2>/dev/null
cat $O/summary.txt; tail -2 $O/tests.log
for f in $O/bench_c*.json; do python -c "
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4))
except Exception as e: print('$f'.split('/')[-1], 'FAILED', e)"; done
grep -E "gru_step_fused|gemm16_planes_kernel<2|launches" $O/model_c4_timeline.txt | head -8 | cut -c1-110
