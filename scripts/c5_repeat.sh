cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
PISO_BENCH_SHARE_GPU=1 PISO_BENCH_SLAB_CHECK=0 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $((29600+i)) bench.py --gpus 8 --decomp slab --steps 1 --warmup 0 --grid 4096 --no-cpu-baseline --no-extras --max-iterations 100 > gpurun_out/c5_$i.out 2> gpurun_out/c5_$i.err
echo "run $i rc $?"; grep -E "Error|error|gave up|status" gpurun_out/c5_$i.err | head -5; python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/c5_$i.out") if l.startswith("{")][-1]); print(d["ms_per_step"], d["config"]["loss"])
except Exception as e: print("no line", e)
PY
done
