#!/bin/bash
# build a variant of the library with extra -D flags into build_exp/lib<tag>.so (A/B runs on one GPU box)
# usage: tools/dev/build_variant.sh <tag> -DEICOS_FAC_DEPTH=3 ...
set -e
tag=$1; shift
cd "$(dirname "$0")/../../eicos_amd/csrc"
out=../../build_exp/obj_$tag; mkdir -p $out
F="-O3 -std=c++17 -fPIC -Wall -Wno-unused-parameter ${FPC:--ffp-contract=off} $*"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c kernels.hip -o $out/kernels.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c kernels_ldsres.hip -o $out/kernels_ldsres.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c kernels_w2.hip -o $out/kernels_w2.o &
/opt/rocm/bin/hipcc $F -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c api.cpp -o $out/api.o
/opt/rocm/bin/hipcc $F -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -pthread -c multi.cpp -o $out/multi.o
/opt/rocm/bin/hipcc $F -x c++ -c symbolic.cpp -o $out/symbolic.o
/opt/rocm/bin/hipcc $F -x c++ -c plans.cpp -o $out/plans.o
/opt/rocm/bin/hipcc $F -x c++ -c tiles.cpp -o $out/tiles.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o ../../build_exp/lib$tag.so $out/kernels.o $out/kernels_ldsres.o $out/kernels_w2.o $out/api.o $out/multi.o $out/symbolic.o $out/plans.o $out/tiles.o
echo built build_exp/lib$tag.so
