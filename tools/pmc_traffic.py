"""HBM-side bytes of a kernel family from rocprofv3 --pmc passes (counter_collection CSVs).
usage: pmc_traffic.py <fetch_pass.csv> <write_pass.csv> <kernel substring[|substring...]> <out.json> [config dtype commit [prompts_per_gpu]]
A family of kernels is named by '|'-separated substrings; bytes are summed over all of them, launches count every dispatch but the
split-K combine (splitk_epilogue: the second kernel of ONE GEMM launch, as bench.py counts launches).
FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived metrics); on gfx950 FETCH_SIZE tallies the 128-byte requests of
wide (16 B/lane) reads at 64 B, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte stores."""
import collections, csv, json, sys
fetch_csv, write_csv, pat, out = sys.argv[1:5]
pats = pat.split("|")

def total(path, counter):
    tot, n = 0.0, set()
    for r in csv.DictReader(open(path)):
        if any(q in r["Kernel_Name"] for q in pats) and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            if "splitk_epilogue" not in r["Kernel_Name"] or pats == ["splitk_epilogue"]:
                n.add(r["Dispatch_Id"])
    return tot, len(n)

f_kib, nf = total(fetch_csv, "FETCH_SIZE")
w_kib, nw = total(write_csv, "WRITE_SIZE")
res = {"kernel": pat, "launches_fetch_pass": nf, "launches_write_pass": nw,
       "fetch_bytes_per_launch_raw": f_kib * 1024 / max(nf, 1), "fetch_bytes_per_launch_corrected_x2": 2 * f_kib * 1024 / max(nf, 1),
       "write_bytes_per_launch": w_kib * 1024 / max(nw, 1)}
res["traffic_bytes_per_launch"] = res["fetch_bytes_per_launch_corrected_x2"] + res["write_bytes_per_launch"]
# stamp: which kernel sources these counters describe (bench.py refuses a record whose digest differs from its own build's) and
# which configuration was profiled (optional arguments: config dtype commit)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
files = ("quant_act.hip", "quant_common.h") if "quant_act" in pat else bench.GEMM_SOURCES
res["digest_files"] = list(files)
res["csrc_digest"] = bench.csrc_digest(files)
if len(sys.argv) > 5:
    res["config"] = sys.argv[5]
    res["dtype"] = sys.argv[6] if len(sys.argv) > 6 else "fp32"
    res["measured_at_commit"] = sys.argv[7] if len(sys.argv) > 7 else None
    if len(sys.argv) > 8:
        res["prompts_per_gpu"] = int(sys.argv[8])
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
