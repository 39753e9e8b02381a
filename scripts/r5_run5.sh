#!/bin/bash
# the whole GPU suite (what the driver runs at round end) with per-test timeouts, then the default bench line
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 2700 python -m pytest tests/ -q -m gpu --timeout 700 --maxfail 6 -p no:cacheprovider > gpurun_out/r5_full.log 2>&1
echo "rc $?" >> gpurun_out/r5_full.log
tail -n 15 gpurun_out/r5_full.log | cut -c1-300



