"""Attention kernels of the last step in a bench.py kernel trace, grouped by (kernel, grid): count, avg us."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2] if len(sys.argv) > 2 else "attn"
idx = [i for i, r in enumerate(rows) if "gemm_wxa8_kernel" in r["Kernel_Name"]]
win = rows[idx[-280 * 2]:idx[-280]]
agg = collections.OrderedDict()
for r in win:
    if pat in r["Kernel_Name"]:
        k = (r["Kernel_Name"].split("(")[0][:60], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
        a = agg.setdefault(k, [0, 0])
        a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, a in agg.items():
    print("%-62s grid %6s x %4s wg %4s  n=%3d avg %8.1f us  tot %8.1f" % (*k, a[0], a[1] / a[0] / 1e3, a[1] / 1e3))
