"""bench.py's other_configs leg alone (configs 1 - 4 + config 5's grid), e.g. under PISO_HIP_LIB=<another build> for an A/B on one box."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import torch
import bench
r = bench.other_configs(torch.device("cuda"))
r.pop("config5_4096x4096_one_gpu_cg_algorithmic_GBs", None)
c4 = r.get("config4_1024x256_cnn_closure_16_step_unroll")
if isinstance(c4, dict):
    c4.pop("what", None)
print(os.path.basename(os.environ.get("PISO_HIP_LIB", "product")), json.dumps(r))
