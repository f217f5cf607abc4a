"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py.
usage: python tools/prof_summary.py <kernel_trace.csv> <timed_steps> <out.csv> [comment]
Profile `bench.py --no-roofline --no-cpu-baseline`: the window = the last `timed_steps` forwards of the trace."""
import collections
import csv
import sys

trace, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
comment = sys.argv[4] if len(sys.argv) > 4 else ""
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# step boundaries: the batched time-embedding projection (dgq_linear_smallm_batch) runs exactly once per forward; the window
# is the kernels between its occurrence `steps` forwards before the last one and the last one = `steps` whole periods
marker = "linear_smallm_kernel"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(idx) > steps:
    win = rows[idx[-steps - 1] + 1: idx[-1] + 1]
else:                                              # older builds: one gemm_wxa8_kernel per quantized layer (280 in SD1.4)
    idx = [i for i, r in enumerate(rows) if "gemm_wxa8_kernel" in r["Kernel_Name"]]
    per = 280 if len(idx) >= 280 * (steps + 1) else len(idx) // (steps + 1)
    win = rows[idx[-per * steps]:]
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    agg[r["Kernel_Name"]][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[r["Kernel_Name"]][1] += 1
tot = sum(v[0] for v in agg.values())
wall = (int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])) / steps / 1e6
with open(out, "w") as f:
    f.write("# %s\n" % comment)
    f.write("# window = %d timed steps; wall %.3f ms/step under the profiler, GPU busy %.3f ms/step, %d dispatches/step\n"
            % (steps, wall, tot / steps / 1e6, len(win) // steps))
    f.write("kernel,calls_per_step,ms_per_step,avg_us,pct_of_busy\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        f.write('"%s",%.1f,%.4f,%.2f,%.2f\n' % (k.replace('"', "'"), v[1] / steps, v[0] / steps / 1e6, v[0] / v[1] / 1e3,
                                                 100 * v[0] / tot))
print(open(out).read()[:3000])
