"""Per-kernel sums of a `rocprofv3 --pmc ... --kernel-trace --output-format csv` run: <dir>/*counter_collection.csv -> one row per
kernel name (dispatch count, each counter summed over the dispatches and per dispatch), widest kernels first.
usage: python tools/pmc_by_kernel.py <dir> [skip_first_n_dispatches] > table.txt"""
import csv, glob, sys, collections
d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
names = set()
with open(f) as fh:
    for row in csv.DictReader(fh):
        did = int(row["Dispatch_Id"])
        if did <= skip:
            continue
        k = row["Kernel_Name"][:90]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k].add(did)
        names.add(row["Counter_Name"])
names = sorted(names)
print("%-92s %6s " % ("kernel", "disp") + " ".join("%22s" % n for n in names))
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", agg[k].get(names[0], 0))):
    n = len(cnt[k])
    print("%-92s %6d " % (k, n) + " ".join("%22.0f" % (agg[k][c] / n) for c in names))
