#!/bin/bash
# Counters of the attention kernels on the 4096x4096 D=40 self-attention (Q1K3 + int8 paths): where do the statistics and
# P̂·V passes spend their cycles?  tools/pmc_attn.sh -> gpurun_out/pmc_attn.txt.  Program directly after `--`; counters in
# their own runs with --kernel-trace only.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmca; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
: > $R/gpurun_out/pmc_attn.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/s$i -- python3 $R/tools/bench_attn.py 40,4096,4096 > $O/s$i.log 2>&1
  C=$(find $O/s$i -name "*counter_collection.csv" | head -1)
  python3 - "$C" >> $R/gpurun_out/pmc_attn.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "attn3_" in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("    %-28s mean per launch %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
  rm -rf $O/s$i
done
cat $R/gpurun_out/pmc_attn.txt
