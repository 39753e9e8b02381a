#!/bin/bash
# kernel-trace + stats of forward steps at a small grid (default 256^2, BASELINE config 2): which kernels, how long, how many.
# Usage (via gpurun): bash scripts/prof_small.sh [n]   -> gpurun_out/prof/small_<n>_kernel_stats.csv
R=$GRAFT_REPO_ROOT
N=${1:-256}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_small -o small -- python3 $R/scripts/time_small.py $N > $OUT/small_${N}_run.log 2>&1
for f in $(find /tmp/prof_small -name "*kernel_stats.csv"); do cp $f $OUT/small_${N}_kernel_stats.csv; done
for f in $(find /tmp/prof_small -name "*kernel_trace.csv"); do python3 - "$f" > $OUT/small_${N}_bicg_gaps.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# busy time and gaps inside runs of bi_* kernels
busy = gap = n = 0
prev_end = None
for r in rows:
    name = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "bi_" in name:
        busy += e - s; n += 1
        if prev_end is not None and s - prev_end < 200000:
            gap += max(0, s - prev_end)
        prev_end = e
    else:
        prev_end = None
print("bi_* kernels: %d launches, busy %.1f us, gaps between consecutive ones %.1f us (avg kernel %.2f us, avg gap %.2f us)" % (n, busy / 1e3, gap / 1e3, busy / 1e3 / max(n, 1), gap / 1e3 / max(n, 1)))
PY
done
tail -3 $OUT/small_${N}_run.log; cat $OUT/small_${N}_bicg_gaps.txt
