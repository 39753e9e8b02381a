#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for x in 0 256 1280 4352 20736 33024; do
export PISO_EXP=$x
rm -rf /tmp/pb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o bicg -- python3 $GRAFT_REPO_ROOT/scripts/bench_bicg.py 2048 > /dev/null 2>&1
echo "stagger $x"; for f in $(find /tmp/pb -name "*kernel_stats.csv"); do grep "bi_sweep\|bi_spmv\|bi_update" $f | cut -c1-30,60-140; done
done
