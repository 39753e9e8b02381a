import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import numpy as np, torch
from oracle import piso_ref as R
from tests.cases import make_case, oracle_setup, product_setup
from tests.test_gpu_step import run_product_step, SOLVER
f32 = np.float32
name = "spatial_ml"
c = make_case(name, 16, 12, seed=4)
kw = dict(SOLVER, lin_double=True, lin_tol=1e-10)
s = oracle_setup(c, **kw); P = product_setup(c, **kw)
rng = np.random.default_rng(2)
forcing = (0.1 * rng.standard_normal(c["vel"].shape)).astype(f32)
valid = np.zeros(c["vel"].shape, f32); valid[0, :, :s.nx, 0] = 1; valid[0, :s.ny, :, 1] = 1
gv = (rng.standard_normal(c["vel"].shape) * valid).astype(f32)
act = s.active[0, 1:-1, 1:-1, 0]
gp = rng.standard_normal(c["p"].shape) * act
gp = (gp - gp.sum() / act.sum() * act).astype(f32)
vo, po, tape = R.piso_step(s, c["vel"], c["p"], c["dt"], c["dirichlet_values"], forcing)
go = R.piso_step_backward(s, tape, gv, gp)
vel_t, p_t, f_t, v3, pn, warn = run_product_step(c, P, forcing, requires_grad=True)
loss = (v3.staggered_tensor() * torch.tensor(gv, device="cuda")).sum() + (pn.data[0, :, :, 0] * torch.tensor(gp, device="cuda")).sum()
loss.backward()
np.savez(os.path.join(ROOT, "gpurun_out", "diag_step.npz"), d_vel=vel_t.grad.cpu().numpy(), d_p=p_t.grad[0,:,:,0].cpu().numpy(),
         d_f=f_t.grad.cpu().numpy(), o_vel=go["d_vel"], o_p=go["d_p"], o_f=go["d_forcing"], gp=gp, gv=gv,
         dmask=c["dirichlet_mask"], v3=v3.staggered_tensor().detach().cpu().numpy(), vo=vo, pn=pn.data[0,:,:,0].detach().cpu().numpy(), po=po)
print("saved")
