#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api_surface.py tests/test_gpu_slab.py tests/test_gpu_fused.py -q -m gpu --timeout 300 -p no:cacheprovider > gpurun_out/r5_bicg_tests.log 2>&1; grep -a "passed\|failed\|^FAILED\|Error" gpurun_out/r5_bicg_tests.log | tail -8 | cut -c1-300
for n in 256 1024 2048; do for f in 0 1; do
  PISO_BICG_FUSE_P=$f timeout 200 python scripts/bench_bicg.py $n 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('n $n fuse $f: us/it', round(d['us_per_iteration'],1), 'frac', round(d['frac'],3), d['solve_to_1e-6'])"
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o bicg -- python3 $GRAFT_REPO_ROOT/scripts/bench_bicg.py 2048 > /dev/null 2>&1
for f in $(find /tmp/pb -name "*kernel_stats.csv"); do head -10 $f | cut -c1-160; done
