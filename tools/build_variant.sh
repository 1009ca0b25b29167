#!/bin/bash
# A/B builds of ONE source of libcti_hip.so: tools/build_variant.sh <name> <file.hip> [-Dflag ...] -> iccv19_vqa-cti_amd/lib/variants/libcti_hip_<name>.so
# (git-ignored, travels with gpurun; load it with CTI_HIP_LIB=<path>).  The other objects come from /tmp/objs (built on first use, 8 at a time).
set -e
root=$(cd "$(dirname "$0")/.." && pwd); csrc=$root/iccv19_vqa-cti_amd/csrc; objs=/tmp/objs; out=$root/iccv19_vqa-cti_amd/lib/variants
name=$1; file=$2; shift 2
mkdir -p $objs $out
for f in $csrc/*.hip; do b=$(basename $f .hip); [ $objs/$b.o -nt $f ] || echo $f; done | xargs -r -P 8 -I{} sh -c "hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-result -c {} -o $objs/\$(basename {} .hip).o"
b=$(basename $file .hip); r=${REPLACES:-$b}          # REPLACES=<source name>: the object this file stands in for (default: its own name)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-result "$@" -c $file -o $objs/${b}__$name.o
others=$(ls $objs/*.o | grep -v "__" | grep -v "/$r.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libcti_hip_$name.so $others $objs/${b}__$name.o
echo $out/libcti_hip_$name.so
