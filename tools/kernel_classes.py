"""Kernel-class shares of a step from a per-step kernel table (tools/prof_summary.py output) -> JSON that bench.py attaches to
its line as `non_hip_kernels` — only for the configuration it was profiled on, stamped with the commit it was measured at.
usage: python tools/kernel_classes.py <per_step.csv> <out.json> [config=c2] [dtype=fp32]"""
import csv, json, subprocess, sys

src, out = sys.argv[1], sys.argv[2]
config = sys.argv[3] if len(sys.argv) > 3 else "c2"
dtype = sys.argv[4] if len(sys.argv) > 4 else "fp32"
CLASSES = (("dgq_gemm", ("gemm_wxa8_kernel", "gemm_big_kernel", "gemm_panel_kernel", "gemm_convq_kernel", "conv_rowsum_kernel", "splitk_epilogue_kernel", "linear_smallm_kernel")),
           ("dgq_quantise_on_load", ("quant_act_",)),
           ("dgq_attention", ("attn3_", "attn_stats", "attn_pv", "fakequant_rows", "logquant", "max_f32")),
           ("dgq_groupnorm_statistics", ("gn_partial", "gn_finalize", "gn_from_partials")),
           ("dgq_fp32_conv", ("conv_f32w",)))
lines = [l for l in open(src) if not l.startswith("#")]
head = [l for l in open(src) if l.startswith("#")]
rows = list(csv.DictReader(lines))
agg = {name: [0.0, 0.0] for name, _ in CLASSES}
agg["torch_or_miopen"] = [0.0, 0.0]
top = {}
for r in rows:
    k, ms, calls = r["kernel"], float(r["ms_per_step"]), float(r["calls_per_step"])
    for name, pats in CLASSES:
        if any(p in k for p in pats):
            agg[name][0] += ms; agg[name][1] += calls
            break
    else:
        agg["torch_or_miopen"][0] += ms; agg["torch_or_miopen"][1] += calls
        top[k[:70]] = round(top.get(k[:70], 0.0) + ms, 4)
busy = sum(v[0] for v in agg.values())
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = "unknown"
json.dump({"source": "%s (rocprofv3 --kernel-trace of `bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --windows 1`)" % src,
           "profile_header": " ".join(h.strip("# \n") for h in head), "config": config, "dtype": dtype, "graph": True,
           "measured_at_commit": commit, "gpu_busy_ms_per_step": round(busy, 3),
           "classes": {k: {"ms_per_step": round(v[0], 3), "dispatches_per_step": round(v[1], 1), "share": round(v[0] / busy, 4)}
                       for k, v in agg.items()},
           "torch_or_miopen_top": dict(sorted(top.items(), key=lambda kv: -kv[1])[:8])}, open(out, "w"), indent=1)
print(open(out).read())
