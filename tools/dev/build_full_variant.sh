#!/bin/bash
# build a variant of the WHOLE library (device units AND host objects) with extra flags into build_exp/lib<tag>.so
# usage: tools/dev/build_full_variant.sh <tag> -DEICOS_IDXB=1 ...
set -e
tag=$1; shift
cd "$(dirname "$0")/../../eicos_amd/csrc"
out=../../build_exp/obj_$tag; mkdir -p $out
F="-O3 -std=c++17 -fPIC -Wall -Wno-unused-parameter -ffp-contract=off $*"
for u in kernels kernels_t128 kernels_t512 kernels_ldsres kernels_w2 kernels_ubl256 kernels_ubl512; do /opt/rocm/bin/hipcc --offload-arch=gfx950 $F -mllvm -amdgpu-sched-strategy=max-ilp -c $u.hip -o $out/$u.o & done
for u in api multi symbolic plans tiles; do /opt/rocm/bin/hipcc $F -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -pthread -c $u.cpp -o $out/$u.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o ../../build_exp/lib$tag.so $out/*.o
echo built build_exp/lib$tag.so
