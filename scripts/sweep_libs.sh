#!/bin/bash
# A/B of prebuilt library variants: scripts/sweep_libs.sh <size> <lib1> <lib2> ...   (libraries under scripts/_bin)
R=$GRAFT_REPO_ROOT
size=$1; shift
cp $R/differentiable-piso_amd/diffpiso/libpiso_hip.so /tmp/lib_orig.so
echo "baseline"; python $R/scripts/bench_cg.py $size 2>&1 | grep grid
for l in "$@"; do
  cp $R/scripts/_bin/$l $R/differentiable-piso_amd/diffpiso/libpiso_hip.so
  echo "$l"; python $R/scripts/bench_cg.py $size 2>&1 | grep grid
done
cp /tmp/lib_orig.so $R/differentiable-piso_amd/diffpiso/libpiso_hip.so
