#!/bin/bash
# A/B on ONE box: scripts/ab.sh <size> <reps> lib1 lib2 ...  ("product" = the in-tree library); diag libraries print per-phase clocks
R=$GRAFT_REPO_ROOT
size=$1; reps=$2; shift; shift
for r in $(seq $reps); do
  for l in "$@"; do
    if [ "$l" = product ]; then
      echo -n "$l: "; python $R/scripts/bench_cg.py $size 2>&1 | grep grid
    else
      echo -n "$l: "; PISO_HIP_LIB=$R/scripts/_bin/$l PISO_CG_PERSIST_TIMING=1 python $R/scripts/bench_cg.py $size 2>&1 | grep "grid\|cg_persist"
    fi
  done
done
