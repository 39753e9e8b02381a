#!/bin/bash
# What the driver runs at round end, plus the profiling recipe, in ONE gpurun call: the whole GPU suite (per-test timeouts), then
# profile_sq.sh (SQ counters of the CG kernel), profile_bench.sh <tag> (kernel stats, PMC passes, traffic.json, the bench line),
# prof_small.sh 256 and prof_config4.sh.
#   gpurun --timeout 3000 -- 'bash scripts/round_end.sh r05_x'      then copy gpurun_out/prof/<tag>_* into profiles/
set -u
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the root of the snapshot)}
TAG=${1:-rXX}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/ -q -m gpu --timeout 500 --maxfail 8 -p no:cacheprovider > gpurun_out/full_suite.log 2>&1
echo "rc $?" >> gpurun_out/full_suite.log
grep -a "passed\|failed\|^FAILED\|^ERROR" gpurun_out/full_suite.log | tail -12 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
# SQ counters of the CG kernel first: the bench line of profile_bench.sh's last step reads profiles/sq_counters.json (same-sources rule)
timeout 600 bash scripts/profile_sq.sh $TAG 2048 > gpurun_out/profile_sq.log 2>&1; echo "profile_sq rc $?"
[ -s gpurun_out/prof/sq_counters.json ] && cp gpurun_out/prof/sq_counters.json profiles/sq_counters.json
timeout 1500 bash scripts/profile_bench.sh $TAG
echo "profile rc $?"
cd $R
timeout 300 bash scripts/prof_small.sh 256 > gpurun_out/prof_small.log 2>&1; echo "prof_small rc $?"
timeout 400 bash scripts/prof_config4.sh > gpurun_out/prof_config4.log 2>&1; echo "prof_config4 rc $?"
