#!/bin/bash
# Round-6 evidence in one gpurun call (tools/final_evidence.sh + tools/profile_step.sh + the other configurations' bench lines):
#   tools/run_evidence_r06.sh <tag> <commit>      -> gpurun_out/<tag>/*, gpurun_out/<tag>_*   (copy what is to be judged into profiles/)
TAG=${1:-r06}; COMMIT=${2:-unknown}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O
bash tools/final_evidence.sh $TAG $COMMIT > gpurun_out/${TAG}_evidence.log 2>&1 || { tail -20 gpurun_out/${TAG}_evidence.log; exit 1; }
tail -3 gpurun_out/${TAG}_evidence.log
bash tools/profile_step.sh $TAG > gpurun_out/${TAG}_profile_step.log 2>&1 || { tail -20 gpurun_out/${TAG}_profile_step.log; exit 1; }
python tools/kernel_classes.py gpurun_out/${TAG}_bench_kernel_stats_per_step.csv gpurun_out/${TAG}_step_kernel_classes.json | tail -3
for cfg in c3 c4 c5; do
  python bench.py --config $cfg --no-cpu-baseline > $O/bench_line_$cfg.json 2> $O/bench_$cfg.err || { tail -5 $O/bench_$cfg.err; exit 1; }
  python - $O/bench_line_$cfg.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["config"]["config_id"], d["value"], d["unit"], d["ms_per_step"], "ms; gemm frac", (d.get("roofline") or {}).get("frac"), "step frac", (d.get("roofline_step") or {}).get("frac_of_binding_roofs"))
PY
done
python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_line_bf16.json 2> $O/bench_bf16.err || { tail -5 $O/bench_bf16.err; exit 1; }
python bench.py --no-cpu-baseline > $O/bench_line_fp32_beside_bf16.json 2> $O/bench_fp32b.err
python bench.py --prompts-per-gpu 4 --no-cpu-baseline > $O/bench_line_c2_prompts4.json 2> $O/bench_p4.err || { tail -5 $O/bench_p4.err; exit 1; }
for f in bf16 fp32_beside_bf16 c2_prompts4; do python - $O/bench_line_$f.json $f <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], d["value"], d["unit"], d["ms_per_step"], "ms; gemm frac", (d.get("roofline") or {}).get("frac"))
PY
done
HEADLINE=1 python tools/bench_gemm.py > $O/gemm_headline_shapes.txt 2>&1; tail -12 $O/gemm_headline_shapes.txt
