#!/bin/bash
# per-phase clocks of the persistent CG kernel: needs scripts/_bin/libpiso_hip_diag.so (PISO_HIPCC_FLAGS=-DPISO_PERSIST_DIAG build),
# loaded through PISO_HIP_LIB (the product library is untouched)
set -u
R=$GRAFT_REPO_ROOT
PISO_HIP_LIB=$R/scripts/_bin/libpiso_hip_diag.so PISO_CG_PERSIST_TIMING=1 python $R/scripts/bench_cg.py "$@" 2>&1 | grep "grid\|cg_persist"
