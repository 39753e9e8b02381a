#!/bin/bash
# kernel tests of the new instances, then the profiling recipe on the same sources
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_multiproc.py -q -m gpu --timeout 500 -p no:cacheprovider > gpurun_out/r5_k.log 2>&1; grep -a "passed\|failed\|^FAILED" gpurun_out/r5_k.log | tail -5 | cut -c1-300
timeout 1500 bash scripts/profile_bench.sh $1
echo "profile rc $?"
cd $R
timeout 300 bash scripts/prof_small.sh 256 > gpurun_out/prof_small.log 2>&1; echo "prof_small rc $?"
timeout 400 bash scripts/prof_config4.sh > gpurun_out/prof_config4.log 2>&1; echo "prof_config4 rc $?"
