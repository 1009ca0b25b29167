#!/bin/bash
cd "$(dirname "$0")/bin"
echo "== correctness"
for shp in "3 40 30 96" "2 1008 3129 512"; do timeout 300 ./f6s_abl0 $shp 2 | tail -3 | tr '\n' ' '; echo; done
echo "== timing (mode-3 shape); ABL bits: 1 no DMA, 2 no MFMA, 4 no stores, 8 no frag reads, 16 no vmcnt waits"
for a in 0 1 2 4 13 14 10 6 30 22 18 16; do timeout 600 ./f6s_abl$a 256 1008 3129 512 10 | tail -1; done
