"""Fixed-work run of the ILU(0)-BiCGStab on the benchmark's matrices (tol 0: every launch of both components does work, no
early returns) -- for rocprofv3 --kernel-trace --stats (scripts/profile_bench.sh) and for a quick rate:
    python3 scripts/bench_bicg.py 2048"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
P = bench.build_problem(n, torch.device("cuda"), 1e-6, 10000, 1000)
# PISO_BICG_PROFILE=1 (the PMC passes of scripts/profile_bench.sh): the four fixed-work solves only, so that the counters' sums divide by four
print(json.dumps(bench.bicgstab_fixed_work(P, n, iters=10, reps=3, real=os.environ.get("PISO_BICG_PROFILE", "0") != "1")))
