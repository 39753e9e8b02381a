"""One rank of a multi-process slab-CG run (started by tests/test_gpu_multiproc.py and usable by hand):

    python tests/slab_worker.py RANK WORLD PORT NX NY WALLS [SHARE_GPU [REGION_ROWS]]

Every rank builds the same NX x NY pressure system (same seed), solves its slab through the PEER transport (mailboxes mapped
across processes with hipIpc handles; torch.distributed/gloo only carries the handles) and compares its rows with the
single-GPU two-kernel solve of the whole system computed in the same process.  SHARE_GPU=1 (default): all ranks use cuda:0 --
the multi-process code path on a one-GPU box; 0: rank r uses cuda:r.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))


def main():
    rank, world, port, nx, ny, walls = (int(v) for v in sys.argv[1:7])
    share = int(sys.argv[7]) if len(sys.argv) > 7 else 1
    region_rows = int(sys.argv[8]) if len(sys.argv) > 8 else 16
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0 if share else rank)
    import diffpiso._native as N
    from tests.cases import pressure_system as case
    from diffpiso.distributed import SlabCommunicator, cg_solve_slab, slab_rows
    from diffpiso.solvers import cg_solve_native
    out = {"rank": rank, "world": world}
    comm = None
    try:
        L, b = case(nx, ny, walls=bool(walls))
        per = not walls
        j0, j1 = slab_rows(rank, world, ny)
        own = slice(j0 * nx, j1 * nx)
        comm = SlabCommunicator(rank=rank, world=world, transport="peer", row_capacity=nx)
        N.set_option("cg_persist_r", region_rows)     # 16-row regions: a slab of ny/world rows takes (nx/128)(ny/world/16)/8 CUs; 4: the
                                                      # instance BASELINE config 5's 4096 x 512 slabs run (two regions per wave)
        N.set_option("cg_segment", 60)                # several segments per solve
        runs = {}
        for label, persist in (("persistent", -1), ("two_kernel", 0)):
            diffs = []
            for nit in (1, 2, 7, 45, 150):
                N.set_option("cg_persist", 0)         # reference: single-GPU two-kernel iteration (no co-residency needed)
                xa, _ = cg_solve_native(nx, ny, per, per, L, b, 1e-30, nit, False, 1000)
                N.set_option("cg_persist", persist)
                xb, itb = cg_solve_slab(comm, nx, ny, per, per, L, b, 1e-30, nit, False, 1000, gather=False)
                assert itb == nit, (itb, nit)
                diffs.append(float((xa[own] - xb).abs().max() / xa.abs().max()))
            N.set_option("cg_persist", 0)
            tol = 1e-7 if nx * ny <= 1024 * 1024 else 1e-6
            xa, ita = cg_solve_native(nx, ny, per, per, L, b, tol, 20000, False, 1000)
            N.set_option("cg_persist", persist)
            xb, itb = cg_solve_slab(comm, nx, ny, per, per, L, b, tol, 20000, False, 1000, gather=False)
            runs[label] = {"fixed_run_diffs": diffs, "converged_its": [ita, itb],
                           "converged_diff": float((xa[own] - xb).abs().max() / xa.abs().max())}
        out.update(runs)
        out["stats"] = comm.stats()
        out["peer_map"] = comm.peer_map                # how the mailboxes were mapped: "ipc" (hipIpc handles) or "fd" (shared file descriptors)
        out["hop_us_matrix"] = comm.hop_matrix(iters=500)
        # timing of the persistent slab iteration (all ranks together)
        N.set_option("cg_persist", -1)
        N.set_option("cg_segment", -1)
        dist.barrier()
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        cg_solve_slab(comm, nx, ny, per, per, L, b, 1e-30, 1500, False, 1 << 30, gather=False)
        torch.cuda.synchronize()
        out["us_per_iteration_persistent_slab"] = 1e6 * (time.perf_counter() - t0) / 1500
        out["ok"] = True
    except Exception as e:  # noqa: BLE001  (reported to the parent, which fails the test)
        out["ok"] = False
        out["error"] = repr(e)
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
