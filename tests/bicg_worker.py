"""One rank of a multi-process slab-BiCGStab run (started by tests/test_gpu_multiproc.py):

    python tests/bicg_worker.py RANK WORLD PORT CASE NX NY

Every rank assembles the same advection-diffusion matrices (oracle assembly of the test case, same seed), solves A x = b and
A^T x = b in float64 and float32 on ONE GPU and cut into WORLD y-slabs (peer transport: dot products all-reduced inside the
scalar kernels, edge rows of the SpMV inputs through the mailboxes) and compares.  All ranks share cuda:0.  One JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))


def main():
    rank, world, port = (int(v) for v in sys.argv[1:4])
    name, nx, ny = sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from oracle import piso_ref as R
    from tests.cases import make_case, oracle_setup
    from diffpiso.distributed import SlabCommunicator
    from diffpiso.solvers import multi_bicgstab_ilu_native
    out = {"rank": rank, "world": world, "runs": []}
    comm = None
    try:
        c = make_case(name, ny, nx, seed=7, variable_viscosity=(name == "spatial_ml"))
        s = oracle_setup(c)
        beta = float(np.prod(c["dx_yx"])) / c["dt"]
        val, rp, col, _, _ = R.advection_matrix(s, c["vel"], beta)
        rhs = np.random.default_rng(11).standard_normal(s.n_u + s.n_v).astype(np.float32)
        x0 = R.flatten_staggered(c["vel"], True)
        comm = SlabCommunicator(rank=rank, world=world, transport="peer", row_capacity=3 * nx + 8)
        dev = lambda a, dt=None: torch.tensor(np.ascontiguousarray(a), device="cuda", dtype=dt)  # noqa: E731
        for tdt, tol in ((torch.float64, 1e-9), (torch.float32, 1e-5)):
            for transpose in (False, True):
                args = (dev(-val, tdt), dev(rp), dev(col), dev(rhs, tdt), dev(x0, tdt), nx, ny, tol, 200, transpose, 8)
                w1 = torch.zeros(1, dtype=torch.uint8, device="cuda")
                w2 = torch.zeros(1, dtype=torch.uint8, device="cuda")
                x1, it1 = multi_bicgstab_ilu_native(*args, w1)
                x2, it2 = multi_bicgstab_ilu_native(*args, w2, slab_comm=comm)
                out["runs"].append({"dtype": str(tdt), "transpose": transpose, "its_single": list(it1), "its_slab": list(it2),
                                    "rel_diff": float((x1 - x2).norm() / x1.norm()), "warn": [int(w1.item()), int(w2.item())]})
        # NaN in the right-hand side of ONE rank's rows: every rank must raise the warning
        bad = rhs.copy()
        bad[3] = np.nan
        w = torch.zeros(1, dtype=torch.uint8, device="cuda")
        multi_bicgstab_ilu_native(dev(-val, torch.float32), dev(rp), dev(col), dev(bad), dev(x0), nx, ny, 1e-5, 20, False, 8, w, slab_comm=comm)
        out["nan_warn"] = int(w.item())
        out["ok"] = True
    except Exception as e:  # noqa: BLE001
        import traceback
        out["ok"] = False
        out["error"] = repr(e) + " | " + traceback.format_exc()[-1500:]
    finally:
        try:
            if comm is not None:
                comm.close()
        except Exception as e:  # noqa: BLE001
            out["close_error"] = repr(e)
    print("SLAB_WORKER " + json.dumps(out), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if out.get("ok") else 1)


if __name__ == "__main__":
    main()
