#!/bin/bash
# Run on the GPU box via gpurun: kernel trace of RANK 1 of a two-rank sharded step (two processes sharing the box's one GPU; rank 1 is
# started directly under rocprofv3, rank 0 plainly) next to the trace of ONE process that steps the same 1024 x 2048 box alone.
# scripts/compare_sharded_trace.py condenses both into <tag>_sharded_rank1_vs_one_gpu.txt: per kernel, launches and workgroups per
# launch - a kernel that still worked on the whole box in the sharded run would show the one-GPU grid size.   Usage: profile_sharded.sh <tag>
R=$GRAFT_REPO_ROOT
TAG=${1:-rXX}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --grid 1024 --no-cpu-baseline --no-extras --max-iterations 200"
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_one -o one -- python3 $R/bench.py --gpus 1 --grid-ny 2048 $ARGS > $OUT/${TAG}_sharded_one_gpu_run.log 2>&1
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29731 WORLD_SIZE=2 PISO_BENCH_SHARE_GPU=1 PISO_BENCH_SLAB_CHECK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
RANK=0 LOCAL_RANK=0 python3 $R/bench.py --gpus 2 --decomp slab-weak $ARGS > $OUT/${TAG}_sharded_rank0_run.log 2>&1 &
P0=$!
export RANK=1 LOCAL_RANK=1
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_rank1 -o rank1 -- python3 $R/bench.py --gpus 2 --decomp slab-weak $ARGS > $OUT/${TAG}_sharded_rank1_run.log 2>&1
wait $P0
python3 $R/scripts/compare_sharded_trace.py /tmp/prof_one /tmp/prof_rank1 > $OUT/${TAG}_sharded_rank1_vs_one_gpu.txt 2>&1
tail -3 $OUT/${TAG}_sharded_rank0_run.log | cut -c1-600
