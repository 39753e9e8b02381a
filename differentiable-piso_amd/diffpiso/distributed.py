"""Slab-decomposed pressure CG across the GPUs of one node -- host side.

One process per GPU (torch.distributed).  The grid is cut into contiguous y-slabs; each rank solves the rows
[rank * ny/world, (rank + 1) * ny/world).  Two transports carry what crosses the slab edges (include/piso_hip.h):

* "peer" (default): every rank owns a peer-mapped mailbox inside the library (hipIpc handle, xGMI peer access); reductions
  and halo rows are written by kernels straight into the consumer's mailbox, and the NORMAL CG iterations run inside the
  persistent kernel (state on chip, one in-kernel exchange per iteration).  torch.distributed only carries the 64-byte
  handles once (any backend: "nccl" on a multi-GPU node, "gloo" when several processes share one GPU in the tests).
* "rccl": the library's own RCCL communicator (unique id broadcast through torch.distributed), two-kernel iteration with
  ncclAllReduce / ncclSend / ncclRecv between the kernels.
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


def _comm_device(device):
    """torch.distributed moves tensors of the backend's kind: device tensors for nccl (= RCCL), host tensors for gloo."""
    import torch.distributed as dist
    return device if dist.get_backend() == "nccl" else torch.device("cpu")


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
        uid_dev = uid.to(_comm_device(device))
        dist.broadcast(uid_dev, src=0)
        uid = uid_dev.cpu()
    return uid


def all_gather_bytes(payload, rank, world, device):
    """Every rank contributes len(payload) bytes and receives all of them in rank order (mailbox handles)."""
    mine = torch.tensor(list(payload), dtype=torch.uint8)
    if world == 1:
        return bytes(payload)
    import torch.distributed as dist
    dev = _comm_device(device)
    parts = [torch.zeros(len(payload), dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(parts, mine.to(dev))
    return b"".join(bytes(t.cpu().tolist()) for t in parts)


def exchange_fds(my_fd, rank, world, device, timeout_s=60.0):
    """Every rank hands a duplicate of its file descriptor to every other rank: a Unix-domain socket per rank (abstract namespace, name
    agreed through torch.distributed), descriptors as SCM_RIGHTS ancillary data.  Returns ([fd received from rank r, -1 for myself], None)
    or ([...], error text).  The received descriptors belong to the caller (close them after the import)."""
    import os
    import socket
    import threading
    got = [-1] * world
    if world == 1:
        return got, None
    name = ("\0piso_fd_%d_%s" % (rank, os.urandom(8).hex())).encode()
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    err = []
    try:
        srv.bind(name)
        srv.listen(world)
        srv.settimeout(timeout_s)

        def serve():
            try:
                for _ in range(world - 1):
                    conn, _addr = srv.accept()
                    with conn:
                        socket.send_fds(conn, [b"fd"], [my_fd])
            except Exception as e:       # noqa: BLE001  (reported below, on every rank)
                err.append("serving: %r" % (e,))
        th = threading.Thread(target=serve, daemon=True)
        th.start()
        names = all_gather_bytes(name.ljust(64, b"\0"), rank, world, device)     # (after listen(): a connect cannot come too early)
        for r in range(world):
            if r == rank:
                continue
            try:
                with socket.socket(socket.AF_UNIX, socket.SOCK_STREAM) as c:
                    c.settimeout(timeout_s)
                    c.connect(b"\0" + names[64 * r + 1:64 * (r + 1)].rstrip(b"\0"))
                    _msg, fds, _flags, _addr = socket.recv_fds(c, 16, 1)
                    got[r] = fds[0]
            except Exception as e:       # noqa: BLE001
                err.append("receiving from rank %d: %r" % (r, e))
        th.join(timeout_s)
    finally:
        srv.close()
    return got, ("; ".join(err) if err else None)


def max_over_ranks(value, device):
    """The bench contract's timing rule: every rank reports the slowest rank's time."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class SlabCommunicator(object):
    """transport "peer": mailboxes mapped into every rank (row_capacity = longest grid row it has to carry);
    transport "rccl": the library's RCCL communicator."""

    def __init__(self, rank=None, world=None, device=None, transport="peer", row_capacity=8192):
        import torch.distributed as dist
        if rank is None:
            rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        self.rank, self.world, self.transport = rank, world, transport
        self.sharded = False                # set by sharding.StepSharding: solver results stay on the rank's rows (no all-gather)
        device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.handle = None
        handle = C.c_void_p()
        if transport == "peer":
            # Two ways to map the mailboxes into every rank, tried in this order (PISO_PEER_MAP = "ipc" / "fd" picks one):
            #   "ipc"  hipIpcGetMemHandle / hipIpcOpenMemHandle (64-byte handles through torch.distributed)
            #   "fd"   exportable virtual-memory allocations, their POSIX file descriptors handed over Unix sockets (SCM_RIGHTS) - for
            #          nodes whose driver refuses hipIpc handles across ranks
            # Setting the transport up is a collective: a rank where a step fails still takes part in the exchanges, and EVERY rank
            # moves on to the next mechanism (or raises) together - nobody is left waiting for a peer that gave up.
            import os
            want = os.environ.get("PISO_PEER_MAP", "auto")
            self.peer_map, reasons = None, []
            for how in (("ipc", "fd") if want == "auto" else (want,)):
                failure = self._peer_setup(how, rank, world, device, int(row_capacity))
                if failure is None:
                    self.peer_map = how
                    break
                reasons.append("%s: %s" % (how, failure))
            if self.peer_map is None:
                raise N.PisoNativeError("the peer transport could not be set up (" + "; ".join(reasons) + ")")
        elif transport == "rccl":
            uid = exchange_unique_id(rank, world, device)
            raw = (C.c_ubyte * 128)(*[int(v) for v in uid.tolist()])
            N.check(N.lib.piso_comm_create(raw, rank, world, C.byref(handle)), "piso_comm_create")
            self.handle = handle
        else:
            raise ValueError("transport must be 'peer' or 'rccl'")

    def _peer_setup(self, how, rank, world, device, row_capacity):
        """One attempt at mapping the mailboxes (collective).  Returns None on success (self.handle set), else what failed - the same
        verdict on every rank."""
        import os
        handle = C.c_void_p()
        failure, created = None, False
        mine64, my_fd = (C.c_ubyte * 64)(), C.c_int(-1)
        with torch.cuda.device(device):
            if os.environ.get("PISO_TEST_REFUSE_PEER", "0") in ("1", how) and rank == world - 1:      # test knob: ONE rank's environment says no
                failure = "refused (PISO_TEST_REFUSE_PEER)"
            else:
                if how == "ipc":
                    status = N.lib.piso_comm_peer_create(rank, world, row_capacity, C.byref(handle), mine64)
                else:
                    status = N.lib.piso_comm_peer_create_fd(rank, world, row_capacity, C.byref(handle), C.byref(my_fd))
                if status != 0:
                    failure = "create failed with status %d: %s" % (status, N.lib.piso_last_error_string().decode())
                else:
                    created = True
            try:
                everybody = all_gather_bytes(bytes(mine64) + bytes([1 if failure else 0]), rank, world, device)
                failed = [r for r in range(world) if everybody[65 * r + 64]]
                if not failed:
                    if how == "ipc":
                        raw = (C.c_ubyte * (64 * world)).from_buffer_copy(b"".join(everybody[65 * r:65 * r + 64] for r in range(world)))
                        status = N.lib.piso_comm_peer_connect(handle, raw)
                    else:
                        fds, err = exchange_fds(my_fd.value, rank, world, device)
                        try:
                            if err is None:
                                status = N.lib.piso_comm_peer_connect_fd(handle, (C.c_int * world)(*fds))
                        finally:
                            for f in fds:
                                if f >= 0:
                                    os.close(f)
                        if err is not None:
                            status, failure = -1, "passing the file descriptors failed: " + err
                    if status != 0 and failure is None:
                        failure = "connect failed with status %d: %s" % (status, N.lib.piso_last_error_string().decode())
                    # (also the barrier: nobody writes into a mailbox that is not mapped everywhere yet)
                    if max_over_ranks(1.0 if failure else 0.0, device) > 0:
                        failed = [-1]
            finally:
                if my_fd.value >= 0:
                    os.close(my_fd.value)
        if failed:
            if created:                             # (no barrier here: the ranks whose mailbox never existed would not join it)
                N.lib.piso_comm_destroy(handle)
            return failure or "failed on rank(s) %s" % (failed,)
        self.handle = handle
        return None

    def pingpong_us(self, a, b, iters=2000):
        """Round-trip microseconds of one tagged word between ranks a and b through the mailboxes (collective: every rank calls it;
        the value is returned on rank a, None elsewhere).  a == b: a rank's own mailbox."""
        us = C.c_float(0.0)
        N.check(N.lib.piso_comm_pingpong(self.handle, int(a), int(b), int(iters), C.byref(us), N.stream_ptr()), "piso_comm_pingpong")
        return float(us.value) if self.rank == a else None

    def hop_matrix(self, iters=2000):
        """hop[a][b] = one-way microseconds (half a round trip) between every pair of ranks, measured pair by pair with everybody else
        idle; the diagonal is a rank's own mailbox.  Collective; every rank returns the full matrix."""
        import torch.distributed as dist
        m = torch.zeros((self.world, self.world), dtype=torch.float64)
        for a in range(self.world):
            for b in range(a, self.world):
                if self.world > 1:
                    torch.cuda.synchronize()
                    dist.barrier()
                us = self.pingpong_us(a, b, iters)
                if us is not None:
                    m[a, b] = m[b, a] = 0.5 * us
        if self.world > 1:
            dev = _comm_device(self.device)
            md = m.to(dev)
            dist.all_reduce(md)                      # (every entry was written by exactly one rank)
            m = md.cpu()
        return [[float(v) for v in row] for row in m]

    def stats(self):
        out = (C.c_longlong * 6)()
        N.check(N.lib.piso_comm_stats(self.handle, out), "piso_comm_stats")
        return {"transport": {1: "rccl", 2: "peer"}[int(out[0])], "persistent_iterations": int(out[1]),
                "persistent_fallbacks": int(out[2]), "persistent_launches": int(out[3]),
                "solves_verified": int(out[4]), "verification_failures": int(out[5])}

    def close(self):
        if self.handle:
            if self.world > 1 and self.transport == "peer":
                import torch.distributed as dist
                torch.cuda.synchronize()
                dist.barrier()                      # nobody unmaps a mailbox a peer's kernel may still write to
            N.lib.piso_comm_destroy(self.handle)
            self.handle = None


def cg_solve_slab(comm, nx, ny, per_x, per_y, L, div, accuracy, max_iterations, rank_deficient, residual_reset, gather=True):
    """Distributed counterpart of solvers.cg_solve_native: L [ny*nx*5] and div [ny*nx] are the FULL (replicated) arrays; this
    rank solves its slab.  gather=True: every rank returns the full solution; gather=False: only its own rows."""
    assert L.dtype == torch.float64 and ny % comm.world == 0, "slab CG: fp64, ny divisible by the number of ranks"
    j0, j1 = slab_rows(comm.rank, comm.world, ny)
    nyl = j1 - j0
    off = j0 * nx
    L_loc = L[off * 5:(off + nyl * nx) * 5].contiguous()
    d_loc = div.reshape(-1).to(torch.float64)[off:off + nyl * nx].contiguous()
    return cg_solve_slab_local(comm, nx, nyl, per_x, per_y, L_loc, d_loc, accuracy, max_iterations, rank_deficient, residual_reset, gather)


def cg_solve_slab_local(comm, nx, nyl, per_x, per_y, L_loc, d_loc, accuracy, max_iterations, rank_deficient, residual_reset,
                        gather=False):
    """The same with this rank's slab only: L_loc [nyl*nx*5], d_loc [nyl*nx] (nothing is replicated)."""
    x_loc = torch.empty_like(d_loc)
    rccl_gather = gather and comm.transport == "rccl"
    x_all = torch.empty(nyl * comm.world * nx, dtype=torch.float64, device=d_loc.device) if gather else None
    ws = N.workspace(N.lib.piso_cg_slab_workspace_bytes(nx, nyl, 1), d_loc.device, "cg_slab")
    it = C.c_int(0)
    st = N.lib.piso_cg_solve_slab_f64(comm.handle, nx, nyl, int(per_x), int(per_y), N.ptr(L_loc), N.ptr(d_loc), N.ptr(x_loc),
                                      N.ptr(x_all) if rccl_gather else None, C.c_float(accuracy), int(max_iterations),
                                      int(bool(rank_deficient)), int(residual_reset), C.byref(it), N.ptr(ws), C.c_size_t(ws.numel()),
                                      N.stream_ptr())
    N.check(st, "piso_cg_solve_slab_f64")
    if not gather:
        return x_loc, it.value
    if not rccl_gather:
        if comm.world == 1:
            x_all.copy_(x_loc)
        else:
            import torch.distributed as dist
            dev = _comm_device(d_loc.device)
            parts = [torch.empty(nyl * nx, dtype=torch.float64, device=dev) for _ in range(comm.world)]
            dist.all_gather(parts, x_loc.to(dev))
            x_all = torch.cat(parts).to(d_loc.device)
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


def face_rows_of_rank(rank, world, nx, ny):
    """Element ranges of the flat u-first face vector [u (ny rows of nx + 1), v (ny + 1 rows of nx)] a rank owns: the face rows
    of its cell rows; the duplicate row v[ny] lives on the last rank."""
    j0, j1 = slab_rows(rank, world, ny)
    n_u = (nx + 1) * ny
    return (j0 * (nx + 1), j1 * (nx + 1)), (n_u + j0 * nx, n_u + (j1 + (1 if rank == world - 1 else 0)) * nx)


def gather_face_rows(comm, x, nx, ny):
    """Every rank holds valid values on its own face rows of `x`: fill in everybody else's (all-gather through torch.distributed)."""
    import torch.distributed as dist
    dev = _comm_device(x.device)
    (u0, u1), (v0, v1) = face_rows_of_rank(comm.rank, comm.world, nx, ny)
    per_u = (nx + 1) * (ny // comm.world)
    per_v = nx * (ny // comm.world + 1)                      # (padded to the last rank's size)
    mine = torch.zeros(per_u + per_v, dtype=x.dtype, device=dev)
    mine[:u1 - u0] = x[u0:u1].to(dev)
    mine[per_u:per_u + (v1 - v0)] = x[v0:v1].to(dev)
    parts = [torch.empty_like(mine) for _ in range(comm.world)]
    dist.all_gather(parts, mine)
    out = x.clone()
    for r, part in enumerate(parts):
        (a0, a1), (b0, b1) = face_rows_of_rank(r, comm.world, nx, ny)
        out[a0:a1] = part[:a1 - a0].to(x.device)
        out[b0:b1] = part[per_u:per_u + (b1 - b0)].to(x.device)
    return out


def multi_bicgstab_ilu_slab(comm, values, row_ptr, col_indices, rhs, x0, nx, ny, tol, max_it, transpose, band_rows, warn, gather=True, negate=False):
    """piso_multi_bicgstab_ilu_slab_{f32,f64}: all arrays are the FULL ones on every rank; the rank solves its slab."""
    dt = values.dtype
    assert dt in (torch.float32, torch.float64)      # (both transports: mailbox kernels, or RCCL send / recv + all-reduce)
    values, rhs, x0 = values.contiguous(), rhs.to(dt).contiguous(), x0.to(dt).contiguous()
    row_ptr, col_indices = row_ptr.contiguous(), col_indices.contiguous()
    x = torch.zeros_like(rhs)
    ws = N.workspace(N.lib.piso_bicgstab_workspace_bytes(nx, ny, 8 if dt == torch.float64 else 4), rhs.device, "bicgstab")
    its = (C.c_int * 2)()
    fn = N.lib.piso_multi_bicgstab_ilu_slab_f64 if dt == torch.float64 else N.lib.piso_multi_bicgstab_ilu_slab_f32
    st = fn(comm.handle, N.ptr(values), N.ptr(row_ptr), N.ptr(col_indices), N.ptr(rhs), N.ptr(x0), N.ptr(x), nx, ny, C.c_float(tol),
            int(max_it), (1 if transpose else 0) | (2 if negate else 0), int(band_rows), N.ptr(warn), its, N.ptr(ws), C.c_size_t(ws.numel()), N.stream_ptr())
    N.check(st, "piso_multi_bicgstab_ilu_slab")
    if gather and comm.world > 1:
        x = gather_face_rows(comm, x, nx, ny)
    return x, (its[0], its[1])


def multi_bicgstab_ilu_slab_local(comm, values, row_ptr, col_indices, rhs, x0, nx, ny, tol, max_it, transpose, band_rows, warn, negate=False):
    """piso_multi_bicgstab_ilu_slab_local_{f32,f64}: the slab-decomposed STEP's solve - every array (the workspace included) holds the
    rank's stored rows (sharding.StepSharding, reached through the communicator); nx, ny are the whole grid's.  x is written on the
    owned rows, the halo rows of the result stay zero."""
    sh = comm.step_sharding
    dt = values.dtype
    assert dt in (torch.float32, torch.float64) and sh is not None
    values, rhs, x0 = values.contiguous(), rhs.to(dt).contiguous(), x0.to(dt).contiguous()
    row_ptr, col_indices = row_ptr.contiguous(), col_indices.contiguous()
    assert rhs.numel() == sh.n_faces and x0.numel() == sh.n_faces
    x = torch.zeros_like(rhs)
    elem = 8 if dt == torch.float64 else 4
    ws = N.workspace(N.lib.piso_bicgstab_slab_workspace_bytes(nx, ny, elem, sh.slab_ptr), rhs.device, "bicgstab_slab")
    its = (C.c_int * 2)()
    per_x, per_y = sh.periodic_xy
    fn = N.lib.piso_multi_bicgstab_ilu_slab_local_f64 if dt == torch.float64 else N.lib.piso_multi_bicgstab_ilu_slab_local_f32
    st = fn(comm.handle, N.ptr(values), N.ptr(row_ptr), N.ptr(col_indices), N.ptr(rhs), N.ptr(x0), N.ptr(x), nx, ny, int(per_x), int(per_y),
            C.c_float(tol), int(max_it), (1 if transpose else 0) | (2 if negate else 0), int(band_rows), N.ptr(warn), its, N.ptr(ws), C.c_size_t(ws.numel()),
            N.stream_ptr(), sh.slab_ptr)
    N.check(st, "piso_multi_bicgstab_ilu_slab_local")
    return x, (its[0], its[1])
