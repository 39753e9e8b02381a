"""Slab-decomposed pressure CG across the GPUs of one node -- host side.

One process per GPU (torch.distributed, backend "nccl" = RCCL).  The library owns its own RCCL communicator: rank 0 creates
the 128-byte unique id, it is broadcast with torch.distributed, every rank calls piso_comm_create.  In this round the rest of
the PISO step is replicated on every rank (it is ~1 % of the step at 2048^2); the CG -- 98 % of the time -- is decomposed:
each rank solves the rows [rank * ny/world, (rank + 1) * ny/world) and all ranks receive the full pressure (all-gather).
"""
import ctypes as C

import torch

from . import _native as N


def slab_rows(rank, world, ny):
    """Rows [begin, end) of the cell grid owned by `rank` (contiguous y-slabs; ny must be divisible by world)."""
    if ny % world != 0:
        raise ValueError("slab decomposition needs ny (%d) divisible by the number of ranks (%d)" % (ny, world))
    nyl = ny // world
    return rank * nyl, (rank + 1) * nyl


def exchange_unique_id(rank, world, device, make_id=None):
    """Rank 0 creates the 128-byte RCCL unique id, everybody receives it through torch.distributed (any backend)."""
    import torch.distributed as dist
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        if make_id is None:
            buf = (C.c_ubyte * 128)()
            N.check(N.lib.piso_comm_unique_id(buf), "piso_comm_unique_id")
            uid = torch.tensor(list(buf), dtype=torch.uint8)
        else:
            uid = make_id()
    if world > 1:
        uid_dev = uid.to(device)
        dist.broadcast(uid_dev, src=0)
        uid = uid_dev.cpu()
    return uid


def max_over_ranks(value, device):
    """The bench contract's timing rule: every rank reports the slowest rank's time."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class SlabCommunicator(object):
    def __init__(self, rank=None, world=None, device=None):
        import torch.distributed as dist
        if rank is None:
            rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        self.rank, self.world = rank, world
        device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        uid = exchange_unique_id(rank, world, device)
        raw = (C.c_ubyte * 128)(*[int(v) for v in uid.tolist()])
        handle = C.c_void_p()
        N.check(N.lib.piso_comm_create(raw, rank, world, C.byref(handle)), "piso_comm_create")
        self.handle = handle

    def close(self):
        if self.handle:
            N.lib.piso_comm_destroy(self.handle)
            self.handle = None


def cg_solve_slab(comm, nx, ny, per_x, per_y, L, div, accuracy, max_iterations, rank_deficient, residual_reset):
    """Distributed counterpart of solvers.cg_solve_native: L [ny*nx*5] and div [ny*nx] are the FULL (replicated) arrays;
    this rank solves its slab and every rank returns the full solution."""
    assert L.dtype == torch.float64 and ny % comm.world == 0, "slab CG: fp64, ny divisible by the number of ranks"
    j0, j1 = slab_rows(comm.rank, comm.world, ny)
    nyl = j1 - j0
    off = j0 * nx
    L_loc = L[off * 5:(off + nyl * nx) * 5].contiguous()
    d_loc = div.reshape(-1).to(torch.float64)[off:off + nyl * nx].contiguous()
    x_loc = torch.empty_like(d_loc)
    x_all = torch.empty(ny * nx, dtype=torch.float64, device=d_loc.device)
    ws = N.workspace(N.lib.piso_cg_slab_workspace_bytes(nx, nyl, 1), d_loc.device, "cg_slab")
    it = C.c_int(0)
    st = N.lib.piso_cg_solve_slab_f64(comm.handle, nx, nyl, int(per_x), int(per_y), N.ptr(L_loc), N.ptr(d_loc), N.ptr(x_loc),
                                      N.ptr(x_all), C.c_float(accuracy), int(max_iterations), int(bool(rank_deficient)),
                                      int(residual_reset), C.byref(it), N.ptr(ws), C.c_size_t(ws.numel()), N.stream_ptr())
    N.check(st, "piso_cg_solve_slab_f64")
    return x_all, it.value


def cg_solve_slab_emulated(slabs, nx, ny, per_x, per_y, L, div, accuracy, max_iterations, rank_deficient, residual_reset):
    """`slabs` virtual ranks on one device (loopback communication): test harness for the multi-rank logic."""
    d = div.reshape(-1).to(torch.float64).contiguous()
    x = torch.empty_like(d)
    ws = N.workspace(N.lib.piso_cg_slab_workspace_bytes(nx, ny // slabs, slabs), d.device, "cg_slab_emu")
    it = C.c_int(0)
    st = N.lib.piso_cg_solve_slab_emulated_f64(int(slabs), nx, ny, int(per_x), int(per_y), N.ptr(L), N.ptr(d), N.ptr(x),
                                               C.c_float(accuracy), int(max_iterations), int(bool(rank_deficient)),
                                               int(residual_reset), C.byref(it), N.ptr(ws), C.c_size_t(ws.numel()),
                                               N.stream_ptr())
    N.check(st, "piso_cg_solve_slab_emulated_f64")
    return x, it.value
