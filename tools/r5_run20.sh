#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/mb/bin
O=$GRAFT_REPO_ROOT/gpurun_out/r5_20; mkdir -p $O
{
echo "== correctness"
for v in g16_dmac2 g16_dmac3; do
  for shp in "256 256 32" "256 256 64" "256 256 96" "256 256 128" "1000 777 320" "300 3129 512" "1008 520 512" "513 257 96" "2048 2048 160" "4096 4096 1024"; do
    echo -n "$v: "; timeout 120 ./$v $shp 2 | tail -2 | tr '\n' ' '; echo
  done
done
echo "== timing"
for shp in "4096 4096 4096" "9216 3072 2048" "9216 8192 2048" "3584 8192 1024"; do
  for v in g16_ns4 g16_dmac2 g16_dmac3 g16_ns4 g16_dmac2 g16_dmac3; do
    echo -n "$v: "; timeout 300 ./$v $shp 20 | tail -1
  done
done
} > $O/mb_gemm16_dmac23.txt 2>&1
grep -c " 0 mismatches" $O/mb_gemm16_dmac23.txt; grep -E "mismatch" $O/mb_gemm16_dmac23.txt | grep -v " 0 mismatches" | head; sed -n '/== timing/,$p' $O/mb_gemm16_dmac23.txt | cut -c1-130
