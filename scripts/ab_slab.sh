#!/bin/bash
# A/B of the slab persistent kernel on ONE box: scripts/ab_slab.sh <n> <reps> lib1 lib2 ...  ("product" = the in-tree library)
R=$GRAFT_REPO_ROOT
n=$1; reps=$2; shift; shift
for r in $(seq $reps); do
  for l in "$@"; do
    if [ "$l" = product ]; then
      python $R/scripts/bench_slab1.py $n 2>&1 | grep "grid\|fallbacks" | sed "s/^/$l: /"
    else
      PISO_HIP_LIB=$R/scripts/_bin/lib$l.so python $R/scripts/bench_slab1.py $n 2>&1 | grep "grid\|fallbacks" | sed "s/^/$l: /"
    fi
  done
done
