#!/bin/bash
# on the GPU box: correctness at ragged shapes, then the timing table of every variant, then the vendor yardstick
cd "$(dirname "$0")/bin"
echo "== correctness (ns4 / ns5 / ns3 / nostag)"
for v in g16_ns4 g16_ns5 g16_ns3 g16_nostag; do
  for shp in "256 256 32" "256 256 64" "1000 777 320" "300 3129 512" "1008 520 512" "513 257 96"; do
    timeout 120 ./$v $shp 2 | tail -2 | tr '\n' ' '; echo
  done
done
echo "== timing"
for shp in "4096 4096 4096" "8192 8192 8192" "9216 3072 2048" "9216 11264 2048" "16384 3328 512"; do
  for v in g16_ns4 g16_ns5 g16_ns3 g16_nostag g16_noprio g16_abl1 g16_abl2 g16_abl4 g16_abl8 g16_abl9; do
    timeout 300 ./$v $shp 20 | tail -1
  done
done
echo "== vendor"
cd ../../.. && python tools/ref_gemm_rate.py 30
