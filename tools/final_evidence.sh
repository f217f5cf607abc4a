#!/bin/bash
# Round evidence in one gpurun call: bench line, attention shapes, HBM-side traffic of the step's GEMM / quantise kernels
# (separate --pmc FETCH_SIZE / WRITE_SIZE passes, eager launches so that every dispatch is a counter sample).
#   tools/final_evidence.sh <tag> <commit>   -> gpurun_out/<tag>/*   (the traffic records are stamped with the digest of the kernel sources)
TAG=${1:-r05z}; COMMIT=${2:-unknown}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 600 $O/bench_line.json
python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids > $O/attention_shapes.txt; cat $O/attention_shapes.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --windows 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --windows 1 > $O/pmc_write.log 2>&1
cd $R
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W "gemm_wxa8_kernel|gemm_big_kernel|gemm_panel_kernel|gemm_convq_kernel|splitk_epilogue" $O/gemm_hbm_traffic.json c2 fp32 $COMMIT | tail -3
python tools/pmc_traffic.py $F $W splitk_epilogue $O/splitk_hbm_traffic.json | tail -2
python tools/pmc_traffic.py $F $W quant_act $O/quant_act_hbm_traffic.json c2 fp32 $COMMIT | tail -2
rm -rf $O/pmc_fetch $O/pmc_write
ls $O
