#!/bin/bash
# HBM-side bytes (separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide reads) of
# the scalar-δ 3x3 convolution pair — quantise pass + GEMM — with the unfolded (materialised) operand and with the implicit one, per
# kernel, on the first C5 shape (8 x 320 x 128 x 128 -> 320).   -> gpurun_out/pmc_conv_implicit.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  PLANS=default ONLY_SHAPE=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -- python3 $R/tools/bench_conv_implicit.py 4 > $O/$c.log 2>&1
done
python3 - $O > $R/gpurun_out/pmc_conv_implicit.txt <<'PY'
import csv, sys, glob, collections
O = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(O + "/" + c + "/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        n = r["Kernel_Name"]
        if not any(k in n for k in ("gemm_wxa8_kernel", "gemm_big_kernel", "quant_act", "conv_rowsum")): continue
        key = n.split("(")[0][:110]
        t = tot[key][c]; t[0] += float(r["Counter_Value"]) * 1024; t[1] += 1
print("# HBM-side MB per launch (FETCH_SIZE x 2 + WRITE_SIZE), 8 x 320 x 128 x 128 -> 320, 3x3, scalar δ (W4A6): M = 131072, K = 2880")
print("# algorithmic: input 168 MB fp32; unfolded codes 131072 x 2944 = 386 MB (written by the quantise pass, read by the GEMM); implicit codes 42 MB; output 168 MB fp32")
for k, v in sorted(tot.items()):
    f, w = v["FETCH_SIZE"], v["WRITE_SIZE"]
    print("%-112s launches %3d  fetch %8.1f MB  write %8.1f MB" % (k, f[1], 2 * f[0] / max(f[1], 1) / 1e6, w[0] / max(w[1], 1) / 1e6))
PY
cat $R/gpurun_out/pmc_conv_implicit.txt
