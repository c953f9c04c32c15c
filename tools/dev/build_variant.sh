#!/bin/bash
# build a variant of the library with extra compiler flags into build_exp/lib<tag>.so (A/B runs on one GPU box; EICOS_AMD_LIB selects the library)
# usage: tools/dev/build_variant.sh <tag> -DEICOS_FAC_DEPTH=3 ...        (flags reach the five device translation units; the host objects are the product's)
set -e
tag=$1; shift
cd "$(dirname "$0")/../../eicos_amd/csrc"
make -s -j6 api.o multi.o symbolic.o plans.o tiles.o >/dev/null
out=../../build_exp/obj_$tag; mkdir -p $out
F="-O3 -std=c++17 -fPIC -Wall -Wno-unused-parameter ${FPC:--ffp-contract=off} $*"
for u in kernels kernels_t128 kernels_t512 kernels_ldsres kernels_w2; do /opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c $u.hip -o $out/$u.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o ../../build_exp/lib$tag.so $out/kernels.o $out/kernels_t128.o $out/kernels_t512.o $out/kernels_ldsres.o $out/kernels_w2.o api.o multi.o symbolic.o plans.o tiles.o
echo built build_exp/lib$tag.so
