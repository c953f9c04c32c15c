# same-box comparison of N library variants: bash tools/dev/r4_abn.sh "libA.so libB.so ..." "pattern batch soc" ...   (libraries under build_exp/; "cur" = the tree's library)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
libs=$1; shift
{ for rep in 1 2; do
for w in "$@"; do
for l in $libs; do
if [ $l = cur ]; then echo -n "cur        "; python tools/dev/r4_phases.py $w | head -1
else echo -n "$l "; EICOS_AMD_LIB=$PWD/build_exp/$l python tools/dev/r4_phases.py $w | head -1; fi
done; done; done
} > gpurun_out/r4_abn.log 2>&1
grep -v "Exception ignored\|BrokenPipe" gpurun_out/r4_abn.log | cut -c1-230
