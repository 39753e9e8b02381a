#!/bin/bash
# the rest of the multi-process tests, the eight-rank sharded tests, the new unsharded tests, band heights of the BiCGStab on small grids
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_gpu_multiproc.py -x -v -m gpu --timeout 600 -p no:cacheprovider -k "not slab_cg_over_processes and not test_config5_4096_eight_slabs" > gpurun_out/r5_q3.log 2>&1
echo "rc $?" >> gpurun_out/r5_q3.log
timeout 1500 python -m pytest tests/test_gpu_sharded_fields.py -x -v -m gpu -s --timeout 700 -k "eight or 8" > gpurun_out/r5_q4.log 2>&1
echo "rc $?" >> gpurun_out/r5_q4.log
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_configs.py tests/test_gpu_api_surface.py tests/test_spectral_golden.py tests/test_gpu_closure.py -x -q -m gpu -s --timeout 300 > gpurun_out/r5_q5.log 2>&1
echo "rc $?" >> gpurun_out/r5_q5.log
for n in 256 512 1024; do timeout 200 python scripts/bicg_bands.py $n >> gpurun_out/r5_bands.log 2>&1; done
grep -n "PASSED\|FAILED\|Timeout\|rc \|passed\|failed\|per step\|launches per" gpurun_out/r5_q3.log gpurun_out/r5_q4.log gpurun_out/r5_q5.log | cut -c1-240 | tail -n 70
cat gpurun_out/r5_bands.log | grep band_rows
