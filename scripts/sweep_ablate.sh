#!/bin/bash
# timing-only ablations of the persistent CG kernel: variant libraries built from scripts/ablate/ into scripts/_bin
# (results are wrong by design).  The PRODUCT library is never touched: variants are loaded through PISO_HIP_LIB.
set -u
R=$GRAFT_REPO_ROOT
echo "baseline"; python $R/scripts/bench_cg.py 2048 2>&1 | grep grid
for a in 1 2 3 4; do
  lib=$R/scripts/_bin/libpiso_hip_ab$a.so
  [ -f "$lib" ] || { echo "missing $lib"; continue; }
  echo "ablate $a (1 no publish stores, 2 no halo loads, 3 trivial stencil, 4 no coefficient loads)"
  PISO_HIP_LIB=$lib python $R/scripts/bench_cg.py 2048 2>&1 | grep grid
done
