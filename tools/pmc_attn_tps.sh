#!/bin/bash
# Counters of the 4096 x 4096 (D = 40) attention kernels with one key tile per ring stage (DGQ_ATTN_TPS=1) and with two (default):
# busy / wait / MFMA-busy cycles per launch -> gpurun_out/pmc_attn_tps.txt (profiles/r05_attention_tps_pmc_counters.txt)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmct; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
: > $R/gpurun_out/pmc_attn_tps.txt
for tps in 1 2; do
  if [ $tps = 1 ]; then export DGQ_ATTN_TPS=1; else unset DGQ_ATTN_TPS; fi
  echo "== key tiles per ring stage: $tps" >> $R/gpurun_out/pmc_attn_tps.txt
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/s$tps -- python3 $R/tools/bench_attn.py 40,4096,4096 > $O/s$tps.log 2>&1
  C=$(find $O/s$tps -name "*counter_collection.csv" | head -1)
  python3 - "$C" >> $R/gpurun_out/pmc_attn_tps.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "attn3_stats" in k or "attn3_pv" in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    print("  " + k)
    for c, v in sorted(cs.items()):
        print("      %-28s mean per launch %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
  rm -rf $O/s$tps
done
cat $R/gpurun_out/pmc_attn_tps.txt
