#!/bin/bash
# Builds a MEASUREMENT variant of the library in which one idempotent stage of k_solve runs R times per call -- for
# tools/dev/r4_stage_traffic.sh (HBM traffic and time of ONE stage under the real mixed load = (variant - product) / (R - 1)).
# The product sources are not touched: a patched copy of eicos_amd/csrc is compiled under build_exp/.
#   usage: tools/dev/build_stage_repeat.sh factor 5   -> build_exp/libfacrep5.so   (stage_factor: reads K, writes U / L / D: idempotent)
#          tools/dev/build_stage_repeat.sh solve 3    -> build_exp/libksrep3.so    (kkt_solve, one right-hand side: reads the rhs, writes dx/dy/dz)
#          tools/dev/build_stage_repeat.sh tfactor 3 / dsolve 3 -> libtfacrep3.so / libdksrep3.so: the tile factorisation / the dual right-hand-side solve
set -e
what=$1; R=$2
root="$(cd "$(dirname "$0")/../.." && pwd)"
case $what in factor) tag=facrep$R;; solve) tag=ksrep$R;; tfactor) tag=tfacrep$R;; dsolve) tag=dksrep$R;; *) echo "factor | solve | tfactor | dsolve"; exit 2;; esac
src=$root/build_exp/src_$tag; rm -rf $src; mkdir -p $src; cp $root/eicos_amd/csrc/*.hip $root/eicos_amd/csrc/*.hpp $root/eicos_amd/csrc/*.cpp $src/
sed -i 's#"../../include/eicos_amd.h"#"'$root'/include/eicos_amd.h"#' $src/api.cpp $src/multi.cpp
python3 - "$src/kernels.hip" "$what" "$R" <<'PY'
import sys
p, what, R = sys.argv[1], sys.argv[2], sys.argv[3]
s = open(p).read()
if what == "factor":
    old = "            if (P.tile != 1) { if (P.fac_defer) stage_factor<T, NLDS, I16, true>(ps, W); else stage_factor<T, NLDS, I16, false>(ps, W); }"
elif what == "tfactor":  # the tile path's factorisation (dense-front / the hybrid's top block)
    old = "            if (P.tile) stage_factor_tiles<T, NLDS>(ps, I, W, iter);"
elif what == "dsolve":   # the dual right-hand-side solve
    old = "                kkt_solve<T, 1, I16, 2, true>(ps, I, I, W, stage, 3);\n"
else:
    old = "            kkt_solve<T, NLDS, I16, 1>(ps, I, I, W, stage, 1);\n"
assert s.count(old) == 1, "the call site moved: update tools/dev/build_stage_repeat.sh"
s = s.replace(old, "            for (int rep_ = 0; rep_ < %s; rep_++)\n%s" % (R, old))
open(p, "w").write(s)
PY
cd $src
F="-O3 -std=c++17 -fPIC -Wall -Wno-unused-parameter -ffp-contract=off"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c kernels.hip -o kernels.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c kernels_ldsres.hip -o kernels_ldsres.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c kernels_w2.hip -o kernels_w2.o &
/opt/rocm/bin/hipcc $F -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c api.cpp -o api.o
/opt/rocm/bin/hipcc $F -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -pthread -c multi.cpp -o multi.o
for f in symbolic plans tiles; do /opt/rocm/bin/hipcc $F -x c++ -c $f.cpp -o $f.o; done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o $root/build_exp/lib$tag.so kernels.o kernels_ldsres.o kernels_w2.o api.o multi.o symbolic.o plans.o tiles.o
rm -rf $src
echo built build_exp/lib$tag.so
