#!/bin/bash
# tools/run_stamps.sh <variant> : s_memtime stamps of a BIG_STAMP build for both scale modes (NPH_M / NPH_K as built)
lib=$PWD/dgq_amd/csrc/variants/libdgq_$1.so
NPH=${NPH_M:-2} DGQ_HIP_LIB=$lib python tools/stamp_big.py perM 2>&1 | grep -v amdgpu
NPH=${NPH_K:-1} DGQ_HIP_LIB=$lib python tools/stamp_big.py perK 2>&1 | grep -v amdgpu
