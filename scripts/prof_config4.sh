#!/bin/bash
# kernel stats of one config-4 training iteration (1024 x 256, CNN closure in the loop): which CG instances run, how long per launch.
# Usage (via gpurun): bash scripts/prof_config4.sh   -> gpurun_out/prof/config4_kernel_stats.csv
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cat > /tmp/c4.py <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/differentiable-piso_amd")
import torch, bench
r = bench.config4_training_iteration(torch.device("cuda"))
print({k: v for k, v in r.items() if k != "what"})
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o c4 -- python3 /tmp/c4.py > $OUT/config4_run.log 2>&1
for f in $(find /tmp/prof_c4 -name "*kernel_stats.csv"); do cp $f $OUT/config4_kernel_stats.csv; done
tail -1 $OUT/config4_run.log
head -8 $OUT/config4_kernel_stats.csv | cut -c1-200
