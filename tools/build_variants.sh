#!/bin/bash
# Builds one libdgq_hip.so per experiment setting of ONE translation unit (default gemm_wxa8_big.hip) into dgq_amd/csrc/variants/ for
# A/B runs inside one gpurun call (DGQ_HIP_LIB=dgq_amd/csrc/variants/libdgq_<name>.so python tools/bench_gemm.py ...).
#   usage: tools/build_variants.sh name1:"-DBIG_GROUP_M=4" name2:"-DBIG_ABL=1" ...        (UNIT=gemm_wxa8.hip to vary another file)
set -e
cd "$(dirname "$0")/../dgq_amd/csrc"
make -s
UNIT=${UNIT:-gemm_wxa8_big.hip}
OBJ=${UNIT%.hip}.o
mkdir -p variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off -Wall -Werror=inline-asm"
OTHERS=$(ls *.o | grep -v "^$OBJ$")
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c $UNIT -o variants/$name.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS variants/$name.o -Wl,--version-script=exports.map -o variants/libdgq_$name.so &&
    echo "built variants/libdgq_$name.so ($defs)" ) &
  while [ $(jobs -r | wc -l) -ge 3 ]; do sleep 1; done
done
wait
