#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python scripts/r5_ab.py small 2>/dev/null | grep "grid"
timeout 1500 python -m pytest tests/ -q -m gpu --timeout 500 --maxfail 8 -p no:cacheprovider > gpurun_out/r5_full4.log 2>&1
echo "rc $?" >> gpurun_out/r5_full4.log
grep -a "passed\|failed\|^FAILED\|^ERROR" gpurun_out/r5_full4.log | tail -12 | cut -c1-300
