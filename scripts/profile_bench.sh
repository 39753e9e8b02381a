#!/bin/bash
# Run on the GPU box via gpurun: rocprofv3 kernel-trace stats of the default bench + PMC passes (HBM bytes) on a shortened run.
# Raw traces stay in /tmp on the box; only summaries are copied to gpurun_out/ (64 MiB cap).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/stats_run.log 2>&1
find /tmp/prof_stats -type f | head -20 > $OUT/stats_files.txt
for f in $(find /tmp/prof_stats -name "*stats*.csv"); do cp $f $OUT/; done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --max-iterations 150 > $OUT/fetch_run.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --max-iterations 150 > $OUT/write_run.log 2>&1
find /tmp/prof_fetch /tmp/prof_write -type f | head -20 >> $OUT/stats_files.txt
python3 $R/scripts/summarize_pmc.py /tmp/prof_fetch /tmp/prof_write > $OUT/pmc_summary.txt 2>&1
