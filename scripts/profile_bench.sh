#!/bin/bash
# Run on the GPU box via gpurun: the profiling recipe behind profiles/.
#   1. rocprofv3 --kernel-trace --stats of the default bench (1 step)                     -> <tag>_bench2048_kernel_stats.csv
#   2. PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) on a run shortened with --max-iterations 150
#   3. the same two PMC passes on scripts/_bin/pmc_calib (known byte counts per launch: calibration of the counters)
#   4. rocprofv3 --kernel-trace --stats of a fixed-work BiCGStab run (scripts/bench_bicg.py)  -> <tag>_bicgstab2048_kernel_stats.csv
#   5. scripts/make_traffic_json.py condenses 2 + 3 into traffic.json (with the sha of the kernel sources it was measured on)
# Raw traces stay in /tmp on the box; only summaries are copied to gpurun_out/prof/ (64 MiB cap).  Usage: profile_bench.sh <tag>
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the root of the snapshot)}
TAG=${1:-rXX}
OUT=$R/gpurun_out/prof
mkdir -p $OUT $R/scripts/_bin
[ -x $R/scripts/_bin/pmc_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/scripts/pmc_calib.hip -o $R/scripts/_bin/pmc_calib > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/${TAG}_stats_run.log 2>&1
for f in $(find /tmp/prof_stats -name "*kernel_stats.csv"); do cp $f $OUT/${TAG}_bench2048_kernel_stats.csv; done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --max-iterations 150 > $OUT/${TAG}_fetch_run.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --max-iterations 150 > $OUT/${TAG}_write_run.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/calib_fetch -o calib -- $R/scripts/_bin/pmc_calib > $OUT/${TAG}_calib_fetch_run.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/calib_write -o calib -- $R/scripts/_bin/pmc_calib > $OUT/${TAG}_calib_write_run.log 2>&1
python3 $R/scripts/summarize_pmc.py /tmp/prof_fetch /tmp/prof_write /tmp/calib_fetch /tmp/calib_write > $OUT/${TAG}_bench2048_pmc_fetch_write_summary.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bicg -o bicg -- python3 $R/scripts/bench_bicg.py 2048 > $OUT/${TAG}_bicg_run.log 2>&1
for f in $(find /tmp/prof_bicg -name "*kernel_stats.csv"); do cp $f $OUT/${TAG}_bicgstab2048_kernel_stats.csv; done
# 4b. PMC passes of the same fixed-work BiCGStab run (bytes per solve over all bi_* kernels: bench.py -> bicgstab.traffic)
export PISO_BICG_PROFILE=1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/bicg_fetch -o bicg -- python3 $R/scripts/bench_bicg.py 2048 > $OUT/${TAG}_bicg_fetch_run.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/bicg_write -o bicg -- python3 $R/scripts/bench_bicg.py 2048 > $OUT/${TAG}_bicg_write_run.log 2>&1
unset PISO_BICG_PROFILE
python3 $R/scripts/summarize_pmc.py /tmp/bicg_fetch /tmp/bicg_write > $OUT/${TAG}_bicgstab2048_pmc_fetch_write_summary.txt 2>&1
python3 $R/scripts/make_traffic_json.py $OUT/${TAG}_bench2048_pmc_fetch_write_summary.txt $TAG $OUT/${TAG}_bicgstab2048_pmc_fetch_write_summary.txt $OUT/${TAG}_bicg_fetch_run.log > $OUT/traffic.json 2> $OUT/${TAG}_traffic.log
# 6. the bench line of the SAME sources with the traffic record just taken (bench.py reads profiles/traffic.json and uses it only if the
#    sha of the kernel sources matches): fail loudly if it does not
cp $OUT/traffic.json $R/profiles/traffic.json
python3 $R/scripts/check_traffic_sha.py || { echo "traffic.json does not match the kernel sources" >&2; exit 1; }
python3 $R/bench.py > $OUT/${TAG}_bench2048.json 2> $OUT/${TAG}_bench_run.log
python3 $R/scripts/check_traffic_sha.py $OUT/${TAG}_bench2048.json $TAG
