"""Reads the kernel trace of `COLD_PROBE_TRACE=1 python tools/cold_probe.py` (rocprofv3 --kernel-trace --output-format csv): per library
kernel, the median duration of its dispatches by machine state — hot (replayed back to back), code-cold (46 other kernels ran in between),
data-cold (a 384 MB copy ran in between), both.  usage: cold_probe_trace.py <kernel_trace.csv>"""
import collections, csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
state, seen_since_marker = None, 0
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "logaddexp" in n: state = "hot"; seen_since_marker = 0; continue
    if "hardswish" in n: state = "code"; continue
    if "copyBuffer" in n or ("direct_copy" in n and int(r["Grid_Size_X"]) > 1 << 20): state = "data"; continue
    if "hardtanh" in n or "relu6" in n or "clamp" in n: state = "both"; continue
    if n.startswith("void at::native") or "rocclr" in n or state is None:
        continue
    key = n.split("(")[0].replace("void ", "")[:70]
    if state == "hot":
        seen_since_marker += 1
        if seen_since_marker <= 3:        # the first iteration of a hot run is not hot yet
            continue
    agg.setdefault(key, collections.defaultdict(list))[state].append(d)
print("%-72s %8s %10s %10s %10s   (median us, n)" % ("kernel", "hot", "code-cold", "data-cold", "both"))
for k, v in agg.items():
    cell = lambda s: ("%7.2f/%-3d" % (statistics.median(v[s]), len(v[s]))) if v[s] else "      -    "
    print("%-72s %s %s %s %s" % (k, cell("hot"), cell("code"), cell("data"), cell("both")))
