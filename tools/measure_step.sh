R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02n; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['kernel_ms_per_step'], r['launches_per_step'], r['layers_covered'])"
python tools/tile_sweep.py sd > $O/tile_sweep_sd.txt 2>&1; tail -1 $O/tile_sweep_sd.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --windows 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --windows 1 > $O/pmc_write.log 2>&1
cd $R
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W gemm_wxa8_kernel $O/gemm_hbm_traffic.json | tail -4
python tools/pmc_traffic.py $F $W quant_act $O/quant_act_hbm_traffic.json | tail -3
rm -rf $O/pmc_fetch $O/pmc_write
