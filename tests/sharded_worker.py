"""One rank of a slab-decomposed (sharded) PISO run that hands back FIELDS (started by tests/test_gpu_sharded_fields.py):

    python tests/sharded_worker.py RANK WORLD PORT CASE OUTDIR

Every rank builds the same case (seeded builders), attaches the peer-mailbox communicator to both linear solvers and a
`StepSharding` to the simulation parameters, runs the unrolled steps forward and the reverse sweep of L = 1/2 |u_K|^2 (the sum of
the ranks' parts), and writes ITS rows of u_K, p_K, dL/du_0 and dL/dp_0 to OUTDIR/rank<r>.npz.  `run_case` is also what the test
calls in its own process for the one-GPU run of the same case.

CASE:  fixture:bench1024_tight_step.npz      the benchmark's workload at 1024^2 with converged solves (committed oracle fixture)
       fixture:cfg3_tml_512x256.npz          BASELINE config 3, 4 steps (x periodic, walls in y)
       case:NAME:NY:NX:STEPS                 a set-up of tests/cases.py at that size (spatial_ml: inflow / outflow in x, open y, a per-face viscosity field;
                                             cavity: solid lid row, no-slip mask), converged solves
       box:NX:NY:STEPS:TOL:MAXIT:SHIFT[:PERSIST]   decaying turbulence on an NX x NY periodic box (bench.py's builder); SHIFT 0: un-shifted CG;
                                             PERSIST 0: two-kernel CG iteration"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
HERE = os.path.dirname(os.path.abspath(__file__))


def build_case(case, device):
    """-> dict(sim, lin, ps, domain, vel_t, p_t, dt, steps, nx, ny, p_tol_adjoint)"""
    import numpy as np
    import torch
    import diffpiso as dp
    kind, _, rest = case.partition(":")
    if rest.endswith(":persist0"):                          # (more slab kernels than the ONE shared GPU holds side by side: two-kernel CG iteration)
        import diffpiso._native as N
        N.set_option("cg_persist", 0)
        rest = rest[:-len(":persist0")]
    if kind == "fixture" and rest.startswith("bench"):
        import bench
        d = np.load(os.path.join(HERE, "golden", rest))
        meta = json.loads(str(d["meta"]))
        n, sv = meta["grid"], meta["solver"]
        P = bench.build_problem(n, device, sv["p_tol"], sv["p_max_it"], sv["p_reset"])
        P["lin"].accuracy, P["lin"].max_iterations = sv["lin_tol"], sv["lin_max_it"]
        return dict(sim=P["sim"], lin=P["lin"], ps=P["ps"], domain=P["domain"], vel_t=P["vel_t"], p_t=P["p_t"], dt=P["dt"], steps=1,
                    nx=n, ny=n, p_tol_adjoint=sv.get("p_tol_adjoint"))
    if kind == "fixture":
        from tests.cases import product_setup
        from tests import cases as G                    # (the full-size case builders: tests/cases.py, not the fixture generator)
        d = np.load(os.path.join(HERE, "golden", rest))
        meta = json.loads(str(d["meta"]))
        c = G.tml_case()
        P = product_setup(c, device=str(device), **meta["solver"])
        return dict(sim=P["sim"], lin=P["lin"], ps=P["ps"], domain=P["domain"], vel_t=P["vel_tensor"], p_t=P["pressure"].data, dt=c["dt"],
                    steps=meta["steps"], nx=c["nx"], ny=c["ny"], p_tol_adjoint=None, dx_yx=c["dx_yx"])
    if kind == "case":
        # case:NAME:NY:NX:STEPS - a synthetic set-up of tests/cases.py (cavity, spatial_ml, xper_ywall, periodic) with converged solves
        from tests.cases import make_case, product_setup
        name, ny, nx, steps = rest.split(":")[:4]
        c = make_case(name, int(ny), int(nx), seed=3, variable_viscosity=(name == "spatial_ml"))
        # (un-shifted pressure CG: the shifted operator's iterates are not reproducible between summation orders)
        # (xper_ywall's pressure matrix is singular: the mean of the float32 divergence - round-off, ~2e-10 per cell - is a floor under every
        # cell's residual, so 1e-10 is never met, the loop runs to p_max_it and the reference's unguarded beta = -(r.z) / (p.z),
        # pressure_solve_eigen.cu.cc:351-352, ends 0 / 0 in whichever summation order gets there first; 1e-8 is met in ~100 iterations)
        P = product_setup(c, device=str(device), lin_tol=1e-9, lin_max_it=300, p_tol=1e-8 if name == "xper_ywall" else 1e-10, p_max_it=6000,
                          p_reset=1000, lin_double=True, rank_deficient=False)
        return dict(sim=P["sim"], lin=P["lin"], ps=P["ps"], domain=P["domain"], vel_t=P["vel_tensor"], p_t=P["pressure"].data, dt=c["dt"],
                    steps=int(steps), nx=c["nx"], ny=c["ny"], p_tol_adjoint=None, dx_yx=c["dx_yx"])
    if kind == "box":
        import bench
        nx, ny, steps, tol, maxit, shift = rest.split(":")[:6]
        if len(rest.split(":")) > 6:                        # PERSIST 0: two-kernel CG (slabs whose persistent kernels cannot all be resident on ONE shared GPU)
            import diffpiso._native as N
            N.set_option("cg_persist", int(rest.split(":")[6]))
        P = bench.build_problem(int(nx), device, float(tol), int(maxit), 1000, ny=int(ny))
        P["lin"].accuracy = 1e-6                               # (TOL / MAXIT are the pressure solver's; the advection solves converge: ~3 iterations)
        if int(shift) == 0:
            P["ps"].laplace_rank_deficient = False
        return dict(sim=P["sim"], lin=P["lin"], ps=P["ps"], domain=P["domain"], vel_t=P["vel_t"], p_t=P["p_t"], dt=P["dt"], steps=int(steps),
                    nx=int(nx), ny=int(ny), p_tol_adjoint=None)
    raise ValueError(case)


def run_case(B, sharding=None):
    """Forward `steps` unrolled steps + reverse sweep -> (u_K, p_K, dL/du_0, dL/dp_0, loss, warn): staggered tensors / cell arrays of the
    whole grid on one GPU; with a `sharding` the rank's STORED rows (flat u-first face vectors, [1, rows, nx, 1] cell arrays) - the
    rank's part of L = 1/2 |u_K|^2 is the sum over the faces it owns."""
    import torch
    import diffpiso as dp
    ext = dp.Material.extrapolation_mode(B["domain"].boundaries)
    p_ext = dp.pressure_extrapolation(B["domain"].boundaries)
    if sharding is None:
        vel_t = B["vel_t"].clone().requires_grad_(True)
        p_t = B["p_t"].clone().requires_grad_(True)
        velocity = dp.StaggeredGrid(vel_t, B["domain"].box, extrapolation=ext)
        pressure = dp.CenteredGrid(p_t, B["domain"].box, p_ext)
    else:
        vel_t = sharding.scatter_staggered(B["vel_t"]).requires_grad_(True)
        p_t = sharding.scatter_cells(B["p_t"]).requires_grad_(True)
        velocity = sharding.staggered_grid(vel_t, B["domain"].box, ext)
        pressure = sharding.centered_grid(p_t, B["domain"].box, p_ext)
    va, pa, vn, pn, warn = dp.unroll_piso_steps(velocity, pressure, B["dt"], B["sim"], step_count=B["steps"])
    u = vn.staggered_tensor()
    loss = 0.5 * (u ** 2).sum() if sharding is None else 0.5 * sharding.owned_sum_of_squares(u)
    if B.get("p_tol_adjoint"):
        B["ps"].accuracy = B["p_tol_adjoint"]
    loss.backward()
    return u.detach(), pn.data.detach(), vel_t.grad, p_t.grad, float(loss.detach()), float(sum(float(w.detach().sum()) for w in warn))


def owned_rows_npz(path, sh, u, p, du, dp_):
    """The rows a rank owns, in the shapes the tests gather them in: face rows [rows, nx + 1] (v: the staggered tensor's pad column is
    zero), cell rows [rows, nx]."""
    import numpy as np
    import torch

    def faces(flat):
        uo, vo = sh.owned_faces(flat)
        vpad = torch.cat([vo, torch.zeros((vo.shape[0], 1), dtype=vo.dtype, device=vo.device)], dim=1)
        return uo.cpu().numpy(), vpad.cpu().numpy()
    u_u, u_v = faces(u)
    du_u, du_v = faces(du)
    np.savez(path, j0=sh.j0, j1=sh.j1, last=1 if sh.last else 0, u_v=u_v, u_u=u_u, p=sh.owned_cells(p).cpu().numpy(),
             du_v=du_v, du_u=du_u, dp=sh.owned_cells(dp_).cpu().numpy())


def main():
    rank, world, port = (int(v) for v in sys.argv[1:4])
    case, outdir = sys.argv[4], sys.argv[5]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    out = {"rank": rank, "world": world, "ok": False}
    if world == 1:
        # the ONE-GPU run of the case in a process of its own: the memory figure the sharded ranks' 1 / ranks is measured against
        try:
            B = build_case(case, device)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats(device)
            u, p, du, dp_, loss, warn = run_case(B, None)
            torch.cuda.synchronize()
            out.update(ok=True, max_memory_allocated=int(torch.cuda.max_memory_allocated(device)), loss=loss, warn=warn,
                       cg_iterations=[int(B["ps"].last_iterations or 0), int(B["ps"].last_adjoint_iterations or 0)],
                       bicgstab_iterations=[int(v) for v in (B["lin"].last_iterations or ())])
            ny = B["ny"]
            np.savez(os.path.join(outdir, "rank0.npz"), j0=0, j1=ny, last=1, u_v=u[0, :, :, 0].cpu().numpy(), u_u=u[0, :ny, :, 1].cpu().numpy(),
                     p=p[0, :, :, 0].cpu().numpy(), du_v=du[0, :, :, 0].cpu().numpy(), du_u=du[0, :ny, :, 1].cpu().numpy(),
                     dp=dp_[0, :, :, 0].cpu().numpy())
        except Exception as e:
            import traceback
            out["error"] = "%r\n%s" % (e, traceback.format_exc()[-1500:])
        print("SLAB_WORKER " + json.dumps(out), flush=True)
        return
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = None
    try:
        from diffpiso.distributed import SlabCommunicator
        from diffpiso.sharding import StepSharding
        # the case is built on the HOST (every rank the same seeded arrays of the whole grid): only the rank's rows go to its GPU
        B = build_case(case, torch.device("cpu") if os.environ.get("PISO_SHARDED_HOST_SETUP", "1") == "1" else device)
        nx, ny = B["nx"], B["ny"]
        comm = SlabCommunicator(rank=rank, world=world, device=device, transport="peer", row_capacity=26 * nx + 64)
        B["ps"].slab_comm = comm
        B["lin"].slab_comm = comm
        sh = B["sim"].sharding = StepSharding(comm, nx, ny)
        torch.cuda.reset_peak_memory_stats(device)
        u, p, du, dp_, loss, warn = run_case(B, sh)
        sh.check()
        owned_rows_npz(os.path.join(outdir, "rank%d.npz" % rank), sh, u, p, du, dp_)
        st = comm.stats()
        out["non_finite"] = [int((~torch.isfinite(t)).sum()) for t in (u, p, du, dp_)]          # (u_K, p_K, dL/du_0, dL/dp_0: stored rows)
        out.update(ok=True, loss=loss, warn=warn, stats=st, halo_exchanges=sh.exchanges,
                   cg_iterations=[int(B["ps"].last_iterations or 0), int(B["ps"].last_adjoint_iterations or 0)],
                   bicgstab_iterations=[int(v) for v in (B["lin"].last_iterations or ())],
                   max_memory_allocated=int(torch.cuda.max_memory_allocated(device)))
    except Exception as e:        # the parent reads the reason
        import traceback
        out["error"] = "%r\n%s" % (e, traceback.format_exc()[-1500:])
    finally:
        print("SLAB_WORKER " + json.dumps(out), flush=True)
        try:
            if comm is not None:
                comm.close()
        except Exception:
            pass
        try:
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
