#!/bin/bash
# round 5, second GPU call: the sharded step on LOCAL storage (fields, memory) and everything that touches the row map
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1700 python -m pytest tests/test_gpu_sharded_fields.py -x -q -m gpu -s > gpurun_out/r5_s1.log 2>&1
echo "sharded fields rc $?" >> gpurun_out/r5_s1.log
timeout 1200 python -m pytest tests/test_gpu_multiproc.py -x -q -m gpu > gpurun_out/r5_s2.log 2>&1
echo "multiproc rc $?" >> gpurun_out/r5_s2.log
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_fused.py tests/test_gpu_slab.py tests/test_csr_container_golden.py tests/test_diffusion_golden.py tests/test_gpu_api_surface.py -x -q -m gpu > gpurun_out/r5_s3.log 2>&1
echo "unsharded rc $?" >> gpurun_out/r5_s3.log
tail -n 25 gpurun_out/r5_s1.log; tail -n 8 gpurun_out/r5_s2.log; tail -n 8 gpurun_out/r5_s3.log
