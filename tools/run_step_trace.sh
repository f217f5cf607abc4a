set -e
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
DGQ_BENCH_GEMM_DUMP=gpurun_out/c2_gemm.tsv DGQ_BENCH_QUANT_DUMP=gpurun_out/c2_quant.tsv DGQ_BENCH_ATTN_DUMP=gpurun_out/c2_attn.tsv python bench.py --no-cpu-baseline > gpurun_out/bench_head.json 2> gpurun_out/bench_head.err
tail -c 600 gpurun_out/bench_head.json | head -c 300; echo
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --windows 1 > $R/gpurun_out/prof_bench.log 2>&1
cd $R
K=$(find /tmp/trace -name "*kernel_trace.csv" | head -1)
python tools/step_trace.py $K > gpurun_out/step_trace.tsv
python tools/prof_summary.py $K 5 gpurun_out/head_per_step.csv head > /dev/null
wc -l gpurun_out/step_trace.tsv
