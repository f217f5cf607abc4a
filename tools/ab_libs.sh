#!/bin/bash
# A/B of whole-library builds inside one gpurun call: tools/ab_libs.sh "<command>" lib1.so lib2.so ...   ("-" = the in-tree library)
CMD=$1; shift
for lib in "$@"; do
  echo "=== $lib"
  if [ "$lib" = "-" ]; then bash -c "$CMD" 2>&1 | grep -v amdgpu.ids; else DGQ_HIP_LIB=$lib bash -c "$CMD" 2>&1 | grep -v amdgpu.ids; fi
done
