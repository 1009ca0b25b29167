#!/bin/bash
# on the GPU box: correctness (K a multiple of 64), then the timing table beside the two-groups kernel
cd "$(dirname "$0")/bin"
echo "== correctness"
for shp in "256 256 64" "256 256 128" "1000 777 320" "300 3129 512" "1008 520 512" "513 257 192" "2048 2048 2048"; do
  timeout 120 ./g16y $shp 2 | tail -2 | tr '\n' ' '; echo
done
echo "== timing"
for shp in "4096 4096 4096" "8192 8192 8192" "9216 3072 2048" "9216 11264 2048" "16384 3328 512"; do
  for v in g16_ns4 g16y g16y_abl1 g16y_abl2 g16y_abl4 g16y_abl8 g16y_abl9; do
    timeout 300 ./$v $shp 20 | tail -1
  done
done
