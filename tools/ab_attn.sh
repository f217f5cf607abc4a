cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/attnxl_prof -o attnxl --output-format csv -- python3 $R/tools/bench_attn_xl.py c5 > $R/gpurun_out/attnxl_prof.log 2>&1
cd $R
F=$(find gpurun_out/attnxl_prof -name "*kernel_stats.csv" | head -1); head -8 $F | cut -c1-200
for v in pvst3 pvst4; do echo $v; DGQ_HIP_LIB=dgq_amd/csrc/variants/libdgq_$v.so python tools/bench_attn_xl.py c5 2>&1 | grep -v amdgpu.ids; done
echo st2; python tools/bench_attn_xl.py c5 2>&1 | grep -v amdgpu.ids
