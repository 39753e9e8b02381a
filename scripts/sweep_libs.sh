#!/bin/bash
# A/B of prebuilt library variants: scripts/sweep_libs.sh <size> <lib1> <lib2> ...   (libraries under scripts/_bin, loaded
# through PISO_HIP_LIB; the product library is untouched)
set -u
R=$GRAFT_REPO_ROOT
size=$1; shift
echo "baseline"; python $R/scripts/bench_cg.py $size 2>&1 | grep grid
for l in "$@"; do
  [ -f "$R/scripts/_bin/$l" ] || { echo "missing $l"; continue; }
  echo "$l"; PISO_HIP_LIB=$R/scripts/_bin/$l python $R/scripts/bench_cg.py $size 2>&1 | grep grid
done
