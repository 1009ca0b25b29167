#!/bin/bash
cd "$(dirname "$0")"
mkdir -p bin
b() { name=$1; shift; hipcc -O3 --offload-arch=gfx950 "$@" -o bin/$name mb_f16f6s.hip || exit 1; }
n=0
for a in 0 1 2 4 13 14 10 6 30 22 18 16; do b f6s_abl$a -DMB_ABL=$a & n=$((n+1)); if [ $((n%4)) = 0 ]; then wait; fi; done
wait
ls bin | grep f6s
