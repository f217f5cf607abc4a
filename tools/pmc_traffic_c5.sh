#!/bin/bash
# HBM-side traffic of the GEMM family over a C5 step (8 prompts): the same two --pmc passes as tools/final_evidence.sh, eager launches.
#   tools/pmc_traffic_c5.sh <commit>  -> gpurun_out/c5t/gemm_hbm_traffic.json
COMMIT=${1:-unknown}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c5t; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --config c5 --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --windows 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --config c5 --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --windows 1 > $O/pmc_write.log 2>&1
cd $R
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W gemm_ $O/gemm_hbm_traffic.json c5 fp32 $COMMIT 8 | tail -16
python tools/pmc_traffic.py $F $W quant_act $O/quant_act_hbm_traffic.json c5 fp32 $COMMIT 8 | tail -4
python tools/pmc_traffic.py $F $W attn3_ $O/attention_hbm_traffic.json c5 fp32 $COMMIT 8 | tail -4
rm -rf $O/pmc_fetch $O/pmc_write
