#!/bin/bash
# builds build_exp/sweep_bench (gfx950); the baseline TU includes the whole kernels.hip and takes ~2 minutes the first time
set -e
cd "$(dirname "$0")/../../.."
mkdir -p build_exp/sb
F="-O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950"
H=/opt/rocm/bin/hipcc
if [ ! -f build_exp/sb/base.o ] || [ eicos_amd/csrc/kernels.hip -nt build_exp/sb/base.o ] || [ tools/dev/sweep_bench/bench_base.hip -nt build_exp/sb/base.o ]; then
  $H $F -c tools/dev/sweep_bench/bench_base.hip -o build_exp/sb/base.o &
fi
$H $F -c tools/dev/sweep_bench/bench_main.hip -o build_exp/sb/main.o
$H -O3 -std=c++17 -x c++ -c eicos_amd/csrc/symbolic.cpp -o build_exp/sb/symbolic.o
$H -O3 -std=c++17 -x c++ -c eicos_amd/csrc/plans.cpp -o build_exp/sb/plans.o
wait
$H --offload-arch=gfx950 -o build_exp/sweep_bench build_exp/sb/main.o build_exp/sb/base.o build_exp/sb/symbolic.o build_exp/sb/plans.o
echo built build_exp/sweep_bench
