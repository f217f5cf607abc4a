#!/bin/bash
# copies one evidence run (tools/run_evidence_r06.sh <tag> <commit>) from gpurun_out/ into profiles/ under the round's names
TAG=${1:-r06b}; O=gpurun_out/$TAG
cp $O/bench_line.json profiles/r06_bench_line.json
for c in c3 c4 c5 bf16 fp32_beside_bf16 c2_prompts4; do cp $O/bench_line_$c.json profiles/r06_bench_line_$c.json; done
cp $O/gemm_hbm_traffic.json profiles/r06_gemm_hbm_traffic.json
cp $O/quant_act_hbm_traffic.json profiles/r06_quant_act_hbm_traffic.json
cp $O/splitk_hbm_traffic.json profiles/r06_splitk_hbm_traffic.json
cp $O/attention_shapes.txt profiles/r06_attention_shapes.txt
grep -v amdgpu.ids $O/gemm_headline_shapes.txt > profiles/r06_gemm_headline_shapes.txt
cp gpurun_out/${TAG}_bench_kernel_stats_per_step.csv profiles/r06_bench_kernel_stats_per_step.csv
cp gpurun_out/${TAG}_bench_rocprofv3_kernel_stats_whole_process.csv profiles/r06_bench_rocprofv3_kernel_stats_whole_process.csv
cp gpurun_out/${TAG}_step_dispatch_trace.tsv profiles/r06_step_dispatch_trace.tsv
cp gpurun_out/${TAG}_step_kernel_classes.json profiles/r06_step_kernel_classes.json
ls -la profiles | grep r06_
