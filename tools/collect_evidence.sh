set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02m
mkdir -p $O
cd $R
HEADLINE=1 python tools/bench_gemm.py 20 > $O/gemm_headline_shapes.txt 2>&1
python tools/bench_attn.py > $O/attention_shapes.txt 2>&1
cd /tmp; export TMPDIR=/tmp
for cfg in "8192 10240 1280 perM 5 bf16" "8192 10240 1280 perK 5 bf16" "8192 8192 8192 perM 3 bf16" "512 10240 1280 perK 5 bf16"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_$tag -- python3 $R/tools/one_gemm.py $cfg > $O/pmc_$tag.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmcb_$tag -- python3 $R/tools/one_gemm.py $cfg > $O/pmcb_$tag.log 2>&1
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --windows 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --windows 1 > $O/pmc_write.log 2>&1
cd $R
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W gemm_wxa8_kernel $O/gemm_hbm_traffic.json
python tools/pmc_traffic.py $F $W splitk_epilogue $O/splitk_hbm_traffic.json
python tools/pmc_traffic.py $F $W quant_act $O/quant_act_hbm_traffic.json
# keep only small summaries of the per-shape PMC runs
for d in $O/pmc_8* $O/pmc_5* $O/pmcb_*; do
  [ -d "$d" ] || continue
  C=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$C" "$d.summary.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "gemm_wxa8_kernel" in r["Kernel_Name"]:
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    for k, cs in agg.items():
        f.write(k + "\n")
        for c, v in cs.items():
            f.write("  %-32s launches %d  mean %.6g\n" % (c, len(v), sum(v) / len(v)))
PY
  K=$(find $d -name "*kernel_trace.csv" | head -1)
  python - "$K" "$d.summary.txt" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_wxa8_kernel" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
open(sys.argv[2], "a").write("  kernel duration us: launches %d  mean %.2f  min %.2f\n" % (len(d), sum(d) / len(d), min(d)))
PY
  rm -rf $d
done
rm -rf $O/pmc_fetch $O/pmc_write
ls -la $O
cat $O/gemm_headline_shapes.txt
cat $O/*.summary.txt
cat $O/gemm_hbm_traffic.json
