"""World-size-2 gloo test (CPU) of the host side of the N > 1 path: unique-id distribution, slab partition, max-over-ranks
timing.  The GPU side (RCCL all-reduce + halo exchange inside the library) is covered on the GPU box by tests/test_gpu_slab.py."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diffpiso.distributed import all_gather_bytes, exchange_unique_id, max_over_ranks, slab_rows
        handles = all_gather_bytes(bytes([17 * (rank + 1)] * 64), rank, world, torch.device("cpu"))   # mailbox handles (peer transport)
        assert handles == bytes([17] * 64) + bytes([34] * 64), handles
        # the slab BiCGStab returns valid values on the owned face rows only: the gather that completes the vector
        from diffpiso.distributed import face_rows_of_rank, gather_face_rows

        class _C(object):
            pass
        c = _C()
        c.rank, c.world = rank, world
        nx, ny = 6, 8
        n_u, n_v = (nx + 1) * ny, nx * (ny + 1)
        x = torch.zeros(n_u + n_v)
        (u0, u1), (v0, v1) = face_rows_of_rank(rank, world, nx, ny)
        x[u0:u1] = 10.0 * (rank + 1) + torch.arange(u1 - u0) / 1000.0
        x[v0:v1] = 100.0 * (rank + 1) + torch.arange(v1 - v0) / 1000.0
        full = gather_face_rows(c, x, nx, ny)
        for r in range(world):
            (a0, a1), (b0, b1) = face_rows_of_rank(r, world, nx, ny)
            assert torch.equal(full[a0:a1], 10.0 * (r + 1) + torch.arange(a1 - a0) / 1000.0)
            assert torch.equal(full[b0:b1], 100.0 * (r + 1) + torch.arange(b1 - b0) / 1000.0)
        # the ranges of all ranks tile both components exactly once (the duplicate row v[ny] on the last rank)
        cover = torch.zeros(n_u + n_v)
        for r in range(world):
            (a0, a1), (b0, b1) = face_rows_of_rank(r, world, nx, ny)
            cover[a0:a1] += 1
            cover[b0:b1] += 1
        assert bool((cover == 1).all())
        uid = exchange_unique_id(rank, world, torch.device("cpu"),
                                 make_id=lambda: torch.arange(128, dtype=torch.uint8) * 3 + 1)
        rows = slab_rows(rank, world, 64)
        slow = max_over_ranks(1.0 + rank, torch.device("cpu"))
        # emulate what the slab solver exchanges per iteration with plain tensors: partial sums and one halo row
        part = torch.tensor([float(rank + 1), 2.0 * (rank + 1), 0.5], dtype=torch.float64)
        dist.all_reduce(part)
        top = torch.full((8,), float(rank))                     # my top row -> upper neighbour's lower halo (ring)
        lower_halo = torch.empty(8)
        hi, lo = (rank + 1) % world, (rank - 1) % world
        ops = [dist.P2POp(dist.isend, top, hi), dist.P2POp(dist.irecv, lower_halo, lo)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        q.put((rank, uid.tolist(), rows, slow, part.tolist(), lower_halo.tolist()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_host_logic():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, uid0, rows0, slow0, part0, halo0), (r1, uid1, rows1, slow1, part1, halo1) = res
    assert uid0 == uid1 == [(3 * i + 1) % 256 for i in range(128)]
    assert rows0 == (0, 32) and rows1 == (32, 64)
    assert slow0 == slow1 == 2.0
    assert part0 == part1 == [3.0, 6.0, 1.0]
    assert halo0 == [1.0] * 8 and halo1 == [0.0] * 8


def test_slab_rows_rejects_uneven_split():
    sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
    from diffpiso.distributed import slab_rows
    with pytest.raises(ValueError):
        slab_rows(0, 3, 64)


def test_step_sharding_halo_messages_are_consistent():
    """Host logic of the sharded step (diffpiso/sharding.py): for every pair of ring neighbours the segments a rank SENDS are the
    segments its neighbour expects to RECEIVE (same offsets in the globally indexed vector, same lengths), the duplicate face row
    v[ny] crosses the periodic seam with the last slab's rows, and the row window is what the kernels are told."""
    import ctypes as C
    import diffpiso._native as N
    from diffpiso.sharding import HALO, StepSharding

    class FakeComm(object):
        def __init__(self, rank, world):
            self.rank, self.world, self.handle, self.sharded = rank, world, None, False

    def msgs(arr):
        out = []
        for q in range(4):
            cnt = arr[7 * q]
            out.append([(arr[7 * q + 1 + k], arr[7 * q + 4 + k]) for k in range(cnt)])
        return out        # to_upper, to_lower, from_lower, from_upper

    nx, ny = 12, 48
    n_u = (nx + 1) * ny
    try:
        for world in (2, 4, 8):
            sh = [StepSharding(FakeComm(r, world), nx, ny) for r in range(world)]
            for r in range(world):
                j0, j1, last = C.c_int(), C.c_int(), C.c_int()
                # the row window is an attribute of every CALL (N.use_window names the sharding a kernel wrapper works for), not of
                # the process: building a sharding does not touch it, naming one sets exactly its rows, naming none clears it
                assert N.lib.piso_get_row_window(C.byref(j0), C.byref(j1), C.byref(last)) == 0
                N.use_window(sh[r])
                assert N.lib.piso_get_row_window(C.byref(j0), C.byref(j1), C.byref(last)) == 1
                assert (j0.value, j1.value, last.value) == (r * ny // world, (r + 1) * ny // world, int(r == world - 1))
                N.use_window(None)
                assert N.lib.piso_get_row_window(C.byref(j0), C.byref(j1), C.byref(last)) == 0
                up, lo = (r + 1) % world, (r - 1) % world
                for kind in ("msgs_faces", "msgs_faces_vfirst", "msgs_cells"):
                    mine, theirs_up, theirs_lo = msgs(getattr(sh[r], kind)), msgs(getattr(sh[up], kind)), msgs(getattr(sh[lo], kind))
                    assert mine[0] == theirs_up[2], (world, r, kind)         # what I send up is what my upper neighbour receives from below
                    assert mine[1] == theirs_lo[3], (world, r, kind)         # what I send down is what my lower neighbour receives from above
            # the last slab sends its HALO last v rows AND the duplicate row v[ny] upwards (across the seam to rank 0)
            top = msgs(sh[world - 1].msgs_faces)[0]
            assert top[1] == (n_u + (ny - HALO) * nx, (HALO + 1) * nx)
            assert msgs(sh[0].msgs_faces)[2][1] == top[1]
    finally:
        N.use_window(None)
