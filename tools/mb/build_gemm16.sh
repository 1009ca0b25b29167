#!/bin/bash
# builds the mb_gemm16 variants into tools/mb/bin (travels to the GPU box with the snapshot)
cd "$(dirname "$0")"
mkdir -p bin
b() { name=$1; shift; hipcc -O3 --offload-arch=gfx950 "$@" -o bin/$name mb_gemm16.hip || exit 1; }
b g16_ns4 &
b g16_ns5 -DNS_RING=5 &
b g16_ns3 -DNS_RING=3 &
b g16_nostag -DMB_STAGGER=0 &
wait
b g16_noprio -DMB_SETPRIO=0 &
b g16_abl1 -DMB_ABL=1 &
b g16_abl2 -DMB_ABL=2 &
b g16_abl4 -DMB_ABL=4 &
wait
b g16_abl8 -DMB_ABL=8 &
b g16_abl9 -DMB_ABL=9 &
wait
ls -la bin
