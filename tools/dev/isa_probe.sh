#!/bin/bash
# ISA of ONE stage function of kernels.hip in seconds (the full file takes two minutes):
#   tools/dev/isa_probe.sh 'stage_factor<256, 1, true, true>(ps, (gdbl_p)w)' > /tmp/x.s      (ps, a, b: int, w, i: double *; WPE=2 for the 256-VGPR budget)
#   WPE=2 tools/dev/isa_probe.sh 'a = kkt_post<256, true>(ps, (gdbl_p)i, (gdbl_p)w, a); w[0] = a' | grep -E 'NumVgprs|ScratchSize'
set -e
d=$(mktemp -d)
cat > $d/p.hip <<EOT
#define EICOS_ISA_PROBE 1
#include "$(cd "$(dirname "$0")/../.." && pwd)/eicos_amd/csrc/kernels.hip"
namespace eicos { __global__ __launch_bounds__(${T:-256}, ${WPE:-3}) void probe(int ps, double *w, double *i, int a, int b) { $1; } }
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off --cuda-device-only -S -o - $d/p.hip 2>&1
rm -rf $d
