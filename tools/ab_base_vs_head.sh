set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "splitk or partials or split" 2>&1 | tail -3
for rep in 1 2; do
for lib in dgq_amd/csrc/libdgq_hip_base.so dgq_amd/csrc/libdgq_hip.so; do
  DGQ_HIP_LIB=$GRAFT_REPO_ROOT/$lib python bench.py --no-cpu-baseline --no-roofline > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
  python - $lib <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/ab_tmp.json') if l.startswith('{')][-1])
print("%-40s %.2f steps/s %.3f ms  %s" % (sys.argv[1], d['value'], d['ms_per_step'], d['windows']['ms_per_step_all']))
PY
done; done
