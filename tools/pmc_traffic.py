"""HBM-side bytes of a kernel family from rocprofv3 --pmc passes (counter_collection CSVs).
usage: pmc_traffic.py <fetch_pass.csv> <write_pass.csv> <kernel substring> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived metrics); on gfx950 FETCH_SIZE tallies the 128-byte requests of
wide (16 B/lane) reads at 64 B, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte stores."""
import collections, csv, json, sys
fetch_csv, write_csv, pat, out = sys.argv[1:5]

def total(path, counter):
    tot, n = 0.0, set()
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n.add(r["Dispatch_Id"])
    return tot, len(n)

f_kib, nf = total(fetch_csv, "FETCH_SIZE")
w_kib, nw = total(write_csv, "WRITE_SIZE")
res = {"kernel": pat, "launches_fetch_pass": nf, "launches_write_pass": nw,
       "fetch_bytes_per_launch_raw": f_kib * 1024 / max(nf, 1), "fetch_bytes_per_launch_corrected_x2": 2 * f_kib * 1024 / max(nf, 1),
       "write_bytes_per_launch": w_kib * 1024 / max(nw, 1)}
res["traffic_bytes_per_launch"] = res["fetch_bytes_per_launch_corrected_x2"] + res["write_bytes_per_launch"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
