#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 300 python -m pytest "tests/test_gpu_sharded_fields.py::test_sharded_small_cases_without_periodic_x_two_ranks_vs_one_gpu" tests/test_gpu_conv.py -x -q -m gpu -s --timeout 200 > gpurun_out/r5_d1.log 2>&1
echo "rc $?" >> gpurun_out/r5_d1.log
grep -n "non_finite\|rel-L2\|passed\|failed\|rows missing\|Error" gpurun_out/r5_d1.log | cut -c1-400 | head -40
echo "--- bicg fold"
for f in -1 1; do PISO_BICG_FOLD=$f timeout 120 python scripts/bench_bicg.py 2048 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold $f', d['us_per_iteration'], d['frac'], d['solve_to_1e-6'])"; done
