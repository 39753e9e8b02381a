#!/bin/bash
# timing-only ablations of the persistent CG kernel (PISO_HIPCC_FLAGS=-DPISO_ABLATE=n builds in scripts/_bin; results are wrong by design)
R=$GRAFT_REPO_ROOT
cp $R/differentiable-piso_amd/diffpiso/libpiso_hip.so /tmp/lib_orig.so
echo "baseline"; python $R/scripts/bench_cg.py 2048 2>&1 | grep grid
for a in 1 2 3 4; do
  cp $R/scripts/_bin/libpiso_hip_ab$a.so $R/differentiable-piso_amd/diffpiso/libpiso_hip.so
  echo "ablate $a (1 no publish stores, 2 no halo loads, 3 trivial stencil, 4 no coefficient loads)"; python $R/scripts/bench_cg.py 2048 2>&1 | grep grid
done
cp /tmp/lib_orig.so $R/differentiable-piso_amd/diffpiso/libpiso_hip.so
