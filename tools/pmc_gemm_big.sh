#!/bin/bash
# Counter picture of the 256-row ping-pong GEMM next to the 128x128 tile on the same shapes (two PMC passes each: MFMA busy /
# instruction counts, and wait / LDS counters) -> gpurun_out/pmc_gemm_big.txt.  Program directly after `--`; counters only with
# --kernel-trace.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcb; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_gemm_big.txt; : > $OUT
SETS=("SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
      "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS")
for cfg in "8192 8192 8192 perM" "8192 8192 8192 perK" "8192 10240 1280 perM" "8192 10240 1280 perK"; do
  for force in "128,128,1" "256,256,1"; do
    for si in 0 1; do
      tag=$(echo ${cfg}_${force}_$si | tr ' ,' '__')
      DGQ_GEMM_FORCE=$force rocprofv3 --pmc ${SETS[$si]} --kernel-trace --output-format csv -d $O/$tag -- python3 $R/tools/one_gemm.py $cfg 5 bf16 > $O/$tag.log 2>&1
      C=$(find $O/$tag -name "*counter_collection.csv" | head -1)
      K=$(find $O/$tag -name "*kernel_trace.csv" | head -1)
      python3 - "$C" "$K" "$cfg" "$force" >> $OUT <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
pick = lambda n: ("gemm_wxa8_kernel" in n) or ("gemm_big_kernel" in n)
for r in csv.DictReader(open(sys.argv[1])):
    if pick(r["Kernel_Name"]):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = [r for r in csv.DictReader(open(sys.argv[2])) if pick(r["Kernel_Name"])]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("M N K mode = %s | plan %s | %s | duration us mean %.1f min %.1f (under the profiler)" % (sys.argv[3], sys.argv[4], rows[0]["Kernel_Name"][:70], sum(d) / len(d), min(d)))
for c, v in sorted(agg.items()):
    print("  %-28s mean per launch %.6g" % (c, sum(v) / len(v)))
PY
      rm -rf $O/$tag
    done
  done
done
cat $OUT
