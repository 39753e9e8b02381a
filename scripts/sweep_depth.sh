#!/bin/bash
# A/B of prebuilt library variants (scripts/_bin/libpiso_hip_d*.so): swaps the library in place, runs the CG micro-benchmark
R=$GRAFT_REPO_ROOT
cp $R/differentiable-piso_amd/diffpiso/libpiso_hip.so /tmp/lib_orig.so
for d in 6 5 4; do
  cp $R/scripts/_bin/libpiso_hip_d$d.so $R/differentiable-piso_amd/diffpiso/libpiso_hip.so
  echo "depth $d"; python $R/scripts/bench_cg.py 2048 1024 2>&1 | grep grid
done
cp /tmp/lib_orig.so $R/differentiable-piso_amd/diffpiso/libpiso_hip.so
