#!/bin/bash
# rocprofv3 kernel trace of bench.py (5 timed steps) -> per-step kernel table (tools/prof_summary.py).  Run through gpurun:
#   [BENCH_ARGS="--config c4"] tools/profile_step.sh <tag>    writes gpurun_out/<tag>_bench_kernel_stats_per_step.csv (+ the whole-process stats CSV)
set -e
TAG=${1:-prof}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --windows 1 ${BENCH_ARGS:-} > $O/bench.log 2>&1
cd $R
K=$(find $O/trace -name "*kernel_trace.csv" | head -1)
S=$(find $O/trace -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py $K 5 gpurun_out/${TAG}_bench_kernel_stats_per_step.csv "$TAG" > /dev/null
python tools/step_trace.py $K > gpurun_out/${TAG}_step_dispatch_trace.tsv     # every dispatch of the last step in order (start, duration, gap, grid)
cp $S gpurun_out/${TAG}_bench_rocprofv3_kernel_stats_whole_process.csv
rm -rf $O/trace
head -40 gpurun_out/${TAG}_bench_kernel_stats_per_step.csv | cut -c1-150
