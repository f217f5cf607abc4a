#!/bin/bash
# A/B of kernel builds on one box: tools/ab_gemm.sh "<bench_gemm env>" name1 name2 ...  (libraries from tools/build_variants.sh);
# each variant is first checked against the exact-integer cases (CHECK=0 skips that)
envs="$1"; shift
for v in "$@"; do
  lib=$PWD/dgq_amd/csrc/variants/libdgq_$v.so
  if [ "${CHECK:-1}" = "1" ]; then
    ok=$(DGQ_HIP_LIB=$lib timeout -k 10 120 python tests/dev/debug_big_gemm.py 2>&1 | grep -c " 0 / ")
    echo "=== $v   (exact cases passed: $ok of 18)"
  else
    echo "=== $v"
  fi
  env $envs DGQ_HIP_LIB=$lib timeout -k 10 240 python tools/bench_gemm.py 20 2>&1 | grep -v "amdgpu.ids\|^shape"
done
