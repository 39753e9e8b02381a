#!/bin/bash
# round 5, first GPU call: kernel-level correctness of the edited persistent kernels, then A/B against the round-4 library on this box
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pressure_phiflow.py tests/test_gpu_slab.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r5_t1.log 2>&1
echo "kernel tests rc $?" >> gpurun_out/r5_t1.log
timeout 900 python -m pytest tests/test_gpu_multiproc.py -x -q -m gpu -k "slab_cg_over_processes or slab_bicgstab_over_processes" > gpurun_out/r5_t2.log 2>&1
echo "multiproc tests rc $?" >> gpurun_out/r5_t2.log
for rep in 1 2; do
  PISO_HIP_LIB=$R/scripts/_bin/libr4base.so timeout 600 python scripts/r5_ab.py small big slab >> gpurun_out/r5_ab1.log 2>&1
  timeout 600 python scripts/r5_ab.py small big slab >> gpurun_out/r5_ab1.log 2>&1
done
timeout 600 python scripts/r5_ab.py sweep >> gpurun_out/r5_ab1.log 2>&1
tail -5 gpurun_out/r5_t1.log gpurun_out/r5_t2.log; cat gpurun_out/r5_ab1.log
