#!/bin/bash
# sharded step on local storage: per-test timeouts and names (a hang must say which test and must not eat the GPU budget)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1300 python -m pytest tests/test_gpu_multiproc.py -x -v -m gpu --timeout 420 -p no:cacheprovider > gpurun_out/r5_q2.log 2>&1
echo "rc $?" >> gpurun_out/r5_q2.log
timeout 900 python -m pytest tests/test_gpu_sharded_fields.py -x -v -m gpu -s --timeout 420 -k "not eight and not 8" > gpurun_out/r5_q1.log 2>&1
echo "rc $?" >> gpurun_out/r5_q1.log
grep -n "PASSED\|FAILED\|Timeout\|rc \|ratio to\|sharded vs one" gpurun_out/r5_q2.log gpurun_out/r5_q1.log | cut -c1-260 | tail -n 60
