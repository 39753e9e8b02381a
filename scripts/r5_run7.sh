#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 200 python scripts/debug_sharded_case.py case:xper_ywall:32:128:1 2 > gpurun_out/r5_d2.log 2>&1
grep -n "NONFINITE\|SLAB_WORKER" gpurun_out/r5_d2.log | cut -c1-600
