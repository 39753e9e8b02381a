#!/bin/bash
# A/B of prebuilt library variants (scripts/_bin/libpiso_hip_d*.so) loaded through PISO_HIP_LIB; the product library is untouched
set -u
R=$GRAFT_REPO_ROOT
for d in 6 5 4; do
  lib=$R/scripts/_bin/libpiso_hip_d$d.so
  [ -f "$lib" ] || { echo "missing $lib"; continue; }
  echo "depth $d"; PISO_HIP_LIB=$lib python $R/scripts/bench_cg.py 2048 1024 2>&1 | grep grid
done
