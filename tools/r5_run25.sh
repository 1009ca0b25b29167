#!/bin/bash
# can the GRU chain run in the shadow of the image projection?  288 x 192 GEMM tile (222 registers, 128 KiB LDS: CTI_GEMM16_TILE=1) + 24-KiB GRU ring (variant library)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_25; mkdir -p $O
V=$PWD/tools/variants/libcti_hip_grucompact.so
for i in 1 2; do
  python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_base_$i.json
  CTI_GEMM16_TILE=1 python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_wide_$i.json
  CTI_HIP_LIB=$V CTI_GEMM16_TILE=1 python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_wide_compact_$i.json
  CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_serial_base_$i.json
  CTI_HIP_LIB=$V CTI_GEMM16_TILE=1 CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4_serial_wide_compact_$i.json
done
export CTI_HIP_LIB=$V CTI_GEMM16_TILE=1
bash tools/trace_models.sh > $O/trace_models.log 2>&1
python tools/print_forward_timeline.py gpurun_out/pc_c4/m_kernel_trace.csv > $O/model_c4_timeline_wide_compact.txt 2>&1
find gpurun_out/pc_c3 gpurun_out/pc_c4 -name "*kernel_trace.csv" -delete 2>/dev/null
for f in $O/bench_c*.json; do python -c "
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4))
except Exception as e: print('$f'.split('/')[-1], 'FAILED', e)"; done
grep -E "gru_step_fused|gemm16_planes_kernel<2|launches" $O/model_c4_timeline_wide_compact.txt | head -9 | cut -c1-110
