#!/bin/bash
# A/B of the step under environment switches: tools/ab_bench.sh "NAME=VAL ..." "NAME=VAL ..." ... (one bench.py run per
# argument, no CPU leg); prints value / ms_per_step / roofline of each.  Run through gpurun.
set -e
mkdir -p gpurun_out
i=0
for envs in "$@"; do
  i=$((i+1))
  out=gpurun_out/ab_$i.json
  env $envs python bench.py --no-cpu-baseline ${BENCH_ARGS:-} > $out 2> gpurun_out/ab_$i.err || { tail -5 gpurun_out/ab_$i.err; exit 1; }
  python - "$envs" $out <<'PY'
import json, sys
line = [l for l in open(sys.argv[2]) if l.startswith("{")][-1]
d = json.loads(line)
r = d.get("roofline") or {}
print("%-60s %7.2f steps/s  %6.3f ms  gemm %.3f ms/step frac %.4f launches %s" % (
    sys.argv[1] or "(default)", d["value"], d["ms_per_step"], r.get("kernel_ms_per_step", 0), r.get("frac", 0), r.get("launches_per_step")))
PY
done
