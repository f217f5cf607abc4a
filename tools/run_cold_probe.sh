set -e
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
COLD_PROBE_TRACE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/cp -- python3 $R/tools/cold_probe.py > $R/gpurun_out/cold_probe_run.log 2>&1
cd $R
K=$(find /tmp/cp -name "*kernel_trace.csv" | head -1)
python tools/cold_probe_trace.py $K > gpurun_out/cold_probe_trace.txt
cat gpurun_out/cold_probe_trace.txt
