#!/bin/bash
# full GPU suite, then the profiling recipe on the same sources
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/ -q -m gpu --timeout 500 --maxfail 6 -p no:cacheprovider > gpurun_out/r5_full2.log 2>&1
echo "rc $?" >> gpurun_out/r5_full2.log
tail -n 8 gpurun_out/r5_full2.log | cut -c1-300
timeout 1500 bash scripts/profile_bench.sh r05_b
echo "profile rc $?"
cat gpurun_out/prof/r05_b_bench2048.json | cut -c1-1500
cd $R
timeout 300 bash scripts/prof_small.sh 256 > gpurun_out/prof_small.log 2>&1; echo "prof_small rc $?"
timeout 400 bash scripts/prof_config4.sh > gpurun_out/prof_config4.log 2>&1; echo "prof_config4 rc $?"
ls gpurun_out/prof | head -50
