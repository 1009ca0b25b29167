#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/mb/bin
O=$GRAFT_REPO_ROOT/gpurun_out/r5_18; mkdir -p $O
{
echo "== correctness"
for v in g16_ns4 g16_dmac g16_dmac_noprio; do
  for shp in "256 256 32" "256 256 64" "256 256 96" "1000 777 320" "300 3129 512" "1008 520 512" "513 257 96" "2048 2048 160"; do
    timeout 120 ./$v $shp 2 | tail -2 | tr '\n' ' '; echo
  done
done
echo "== timing"
for shp in "4096 4096 4096" "8192 8192 8192" "9216 3072 2048" "9216 8192 2048" "3584 8192 1024" "2304 2048 2048"; do
  for v in g16_ns4 g16_dmac g16_dmac_noprio g16_ns4 g16_dmac; do
    timeout 300 ./$v $shp 20 | tail -1
  done
done
} > $O/mb_gemm16_dmac.txt 2>&1
grep -c mismatch $O/mb_gemm16_dmac.txt; grep -E "check|TFLOP" $O/mb_gemm16_dmac.txt | cut -c1-150
