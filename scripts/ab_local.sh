#!/bin/bash
# A/B of library variants on the XCD-local grids (256^2, 512 x 256): scripts/ab_local.sh <reps> lib1 lib2 ... ("product" = in-tree)
R=$GRAFT_REPO_ROOT
reps=$1; shift
for r in $(seq $reps); do
  for l in "$@"; do
    if [ "$l" = product ]; then
      python $R/scripts/bench_cg_rect.py 2>&1 | grep "half -1" | grep " 256 x  256\| 512 x  256" | sed "s/^/$l: /"
    else
      PISO_HIP_LIB=$R/scripts/_bin/lib$l.so python $R/scripts/bench_cg_rect.py 2>&1 | grep "half -1" | grep " 256 x  256\| 512 x  256" | sed "s/^/$l: /"
    fi
  done
done
