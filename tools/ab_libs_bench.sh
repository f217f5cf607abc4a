#!/bin/bash
# bench.py once per library build, inside one gpurun call: tools/ab_libs_bench.sh lib1.so lib2.so ...  ("-" = the in-tree library); REPS=2 repeats the round
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-1}); do
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset DGQ_HIP_LIB; else export DGQ_HIP_LIB=$GRAFT_REPO_ROOT/$lib; fi
  python bench.py --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
  python - "$lib" <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/ab_tmp.json') if l.startswith('{')][-1])
print("%-44s %.2f %s  %.3f ms  %s" % (sys.argv[1], d['value'], d['unit'], d['ms_per_step'], d['windows']['ms_per_step_all']))
PY
done; done
