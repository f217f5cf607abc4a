"""Every dispatch of the LAST step in a bench.py kernel trace, in start order: start offset, duration, gap to the previous
dispatch's end, grid, workgroup, LDS, kernel name (template arguments kept).  tools/step_trace.py <kernel_trace.csv> [steps]
-> stdout (tab separated).  The in-step counterpart of the replayed per-launch tables (DGQ_BENCH_*_DUMP)."""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step ends with conv_out's kernel (conv_f32w_smalln_kernel); the last step is the dispatches after the second-to-last one
ends = [i for i, r in enumerate(rows) if "conv_f32w_smalln_kernel" in r["Kernel_Name"]]
if len(ends) < 2:
    sys.exit("no step boundary (conv_f32w_smalln_kernel) found")
win = rows[ends[-2] + 1:ends[-1] + 1]
t0 = int(win[0]["Start_Timestamp"])
prev_end = t0
print("#start_us\tdur_us\tgap_us\tgrid\twg\tlds\tkernel")
for r in win:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0] if "<" not in name else name[:name.rfind(">") + 1] if ">(" not in name else name[:name.index(">(") + 1]
    print("%.2f\t%.2f\t%.2f\t%sx%sx%s\t%s\t%s\t%s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Grid_Size_X"], r["Grid_Size_Y"],
                                                  r["Grid_Size_Z"], r["Workgroup_Size_X"], r.get("LDS_Block_Size", ""), name[:150]))
    prev_end = max(prev_end, e)
