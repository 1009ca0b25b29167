#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/mb/bin
O=$GRAFT_REPO_ROOT/gpurun_out/r5_19; mkdir -p $O
{
for shp in "4096 4096 4096" "9216 3072 2048" "9216 8192 2048"; do
  for v in g16_ns4 g16_abl16 g16_abl1 g16_abl8 g16_abl9 g16_abl2 g16_ns4 g16_abl16; do
    echo -n "$v: "; timeout 300 ./$v $shp 20 | tail -1
  done
done
} > $O/mb_gemm16_abl.txt 2>&1
cat $O/mb_gemm16_abl.txt | cut -c1-150
