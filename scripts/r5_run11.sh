#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for n in 256 1024 2048; do
  timeout 200 python scripts/bench_bicg.py $n 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('n $n: us/it', round(d['us_per_iteration'],1), 'frac', round(d['frac'],3), d['solve_to_1e-6'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o bicg -- python3 $GRAFT_REPO_ROOT/scripts/bench_bicg.py 2048 > /dev/null 2>&1
for f in $(find /tmp/pb -name "*kernel_stats.csv"); do head -12 $f | cut -c1-160; done
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -q -m gpu --timeout 500 --maxfail 8 -p no:cacheprovider > gpurun_out/r5_full3.log 2>&1
echo "rc $?" >> gpurun_out/r5_full3.log
grep -a "passed\|failed\|^FAILED\|^ERROR" gpurun_out/r5_full3.log | tail -12 | cut -c1-300
