#!/bin/bash
# Instruction-mix counters of the GEMM on the XL GEGLU shape, per-K g16 against per-M (the flush / widening VALU next to
# the MFMAs): tools/pmc_valu.sh -> gpurun_out/pmc_valu.txt.  Program directly after `--`, counters only with --kernel-trace.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcv; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_[A-Z0-9_]*\|SQ_ACTIVE_INST_[A-Z0-9_]*\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES" | sort -u | tr '\n' ' ' > $O/available.txt
: > $R/gpurun_out/pmc_valu.txt
for cfg in "8192 10240 1280 perK 5 bf16" "8192 10240 1280 perM 5 bf16" "512 10240 1280 perK 5 bf16" "512 10240 1280 perM 5 bf16"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/$tag -- python3 $R/tools/one_gemm.py $cfg > $O/$tag.log 2>&1
  C=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  K=$(find $O/$tag -name "*kernel_trace.csv" | head -1)
  python3 - "$C" "$K" "$cfg" >> $R/gpurun_out/pmc_valu.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_wxa8_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = [r for r in csv.DictReader(open(sys.argv[2])) if "gemm_wxa8_kernel" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("M N K mode =", sys.argv[3], "| kernel", rows[0]["Kernel_Name"][:60], "| duration us mean %.1f" % (sum(d) / len(d)))
for c, v in sorted(agg.items()):
    print("  %-24s mean per launch %.6g" % (c, sum(v) / len(v)))
PY
  rm -rf $O/$tag
done
cat $O/available.txt | head -c 1500; echo; cat $R/gpurun_out/pmc_valu.txt
