#!/bin/bash
# builds the one-wave-per-SIMD variants (mb_gemm16y.hip) into tools/mb/bin
cd "$(dirname "$0")"
mkdir -p bin
b() { name=$1; shift; hipcc -O3 --offload-arch=gfx950 "$@" -o bin/$name mb_gemm16y.hip || exit 1; }
b g16y &
b g16y_abl1 -DMB_ABL=1 &
b g16y_abl2 -DMB_ABL=2 &
b g16y_abl4 -DMB_ABL=4 &
wait
b g16y_abl8 -DMB_ABL=8 &
b g16y_abl9 -DMB_ABL=9 &
hipcc -O3 --offload-arch=gfx950 -o bin/g16_ns4 mb_gemm16.hip &
wait
ls -la bin | grep g16
