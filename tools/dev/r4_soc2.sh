export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ timeout 900 python -m pytest tests -m gpu -x -q -k "multi_gpu or device_list or soc or random_socp or big_cone or fatal_seeds or update_scalings or every_kernel" 2>&1 | tail -5
for rep in 1 2; do
python tools/dev/r4_phases.py MPC02 1024 1 | head -1
EICOS_AMD_LIB=$PWD/build_exp/libbase.so python tools/dev/r4_phases.py MPC02 1024 1 | head -1
done
python tools/dev/r4_phases.py MPC02 512 1
python tools/dev/hash_outputs.py gpurun_out/r4_hash_new.json > /dev/null 2>&1
EICOS_AMD_LIB=$PWD/build_exp/libbase.so python tools/dev/hash_outputs.py gpurun_out/r4_hash_base.json > /dev/null 2>&1
python - <<'PY'
import json
a=json.load(open('gpurun_out/r4_hash_new.json')); b=json.load(open('gpurun_out/r4_hash_base.json'))
diff=[k for k in a if a[k]!=b.get(k)]
print('hash_outputs: %d problems, differing: %s' % (len(a), diff))
PY
} > gpurun_out/r4_soc2.log 2>&1
cat gpurun_out/r4_soc2.log | cut -c1-300
