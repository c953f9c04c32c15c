# round 4: the build whose stage functions save no callee-saved registers, against build_exp/libbase.so: bit-level fingerprints of 40 problem / variant runs, then the suite
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/dev/hash_outputs.py gpurun_out/hash_new.json > gpurun_out/hash_new.log 2>&1
EICOS_AMD_LIB=$PWD/build_exp/libbase.so python tools/dev/hash_outputs.py gpurun_out/hash_base.json > gpurun_out/hash_base.log 2>&1
python - <<'PY'
import json
a=json.load(open("gpurun_out/hash_new.json")); b=json.load(open("gpurun_out/hash_base.json"))
diff=[k for k in a if a[k]!=b.get(k)]
print("fingerprints:", len(a), "runs; differing:", diff)
PY
bash tools/dev/r4_suite.sh
