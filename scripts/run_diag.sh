#!/bin/bash
# per-phase clocks of the persistent CG kernel: needs scripts/_bin/libpiso_hip_diag.so (PISO_HIPCC_FLAGS=-DPISO_PERSIST_DIAG build)
R=$GRAFT_REPO_ROOT
cp $R/differentiable-piso_amd/diffpiso/libpiso_hip.so /tmp/lib_orig.so
cp $R/scripts/_bin/libpiso_hip_diag.so $R/differentiable-piso_amd/diffpiso/libpiso_hip.so
PISO_CG_PERSIST_TIMING=1 python $R/scripts/bench_cg.py "$@" 2>&1 | grep "grid\|cg_persist"
cp /tmp/lib_orig.so $R/differentiable-piso_amd/diffpiso/libpiso_hip.so
