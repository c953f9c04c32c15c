#!/bin/bash
# build a variant of the two U-in-LDS translation units with extra flags into build_exp/lib<tag>.so (the other objects are the product's)
# usage: tools/dev/build_ubl_variant.sh <tag> -DEICOS_FAC_DEPTH=4 ...
set -e
tag=$1; shift
cd "$(dirname "$0")/../../eicos_amd/csrc"
make -s -j8 >/dev/null
out=../../build_exp/obj_$tag; mkdir -p $out
F="-O3 -std=c++17 -fPIC -Wall -Wno-unused-parameter -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp $*"
for u in kernels_ubl256 kernels_ubl512; do /opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c $u.hip -o $out/$u.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o ../../build_exp/lib$tag.so kernels.o kernels_t128.o kernels_t512.o kernels_ldsres.o kernels_w2.o $out/kernels_ubl256.o $out/kernels_ubl512.o api.o multi.o symbolic.o plans.o tiles.o
echo built build_exp/lib$tag.so
