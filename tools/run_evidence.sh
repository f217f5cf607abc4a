set -e
cd $GRAFT_REPO_ROOT
bash tools/final_evidence.sh r05s6 ec6fd20 > gpurun_out/evidence.log 2>&1 || { tail -20 gpurun_out/evidence.log; exit 1; }
tail -5 gpurun_out/evidence.log
bash tools/profile_step.sh r05s6 > gpurun_out/profile_step.log 2>&1 || { tail -20 gpurun_out/profile_step.log; exit 1; }
python tools/kernel_classes.py gpurun_out/r05s6_bench_kernel_stats_per_step.csv gpurun_out/r05s6_step_kernel_classes.json | tail -3
