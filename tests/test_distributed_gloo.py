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
        # the second way to map the mailboxes hands FILE DESCRIPTORS from rank to rank (Unix sockets, SCM_RIGHTS): every rank ends up
        # with a descriptor of the other rank's file - here an anonymous temporary file that names its owner
        import tempfile
        from diffpiso.distributed import exchange_fds
        with tempfile.TemporaryFile() as f:
            f.write(b"mailbox of rank %d" % rank)
            f.flush()
            fds, err = exchange_fds(f.fileno(), rank, world, torch.device("cpu"))
            assert err is None, err
            assert fds[rank] == -1
            for r, fd in enumerate(fds):
                if r != rank:
                    assert fd >= 0 and fd != f.fileno()
                    assert os.pread(fd, 64, 0) == b"mailbox of rank %d" % r
                    os.close(fd)
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
    """Host logic of the sharded step (diffpiso/sharding.py, LOCAL storage): for every pair of ring neighbours the segments a rank SENDS
    are the segments its neighbour expects to RECEIVE - the same lengths, and the same rows of the whole grid once each side's offsets
    into its STORED arrays are mapped back; the duplicate face row v[ny] crosses the periodic seam with the last slab's rows; scatter
    and owned_* are inverse to each other; the library computes the same sizes."""
    import numpy as np
    import torch
    from diffpiso.sharding import HALO, StepSharding

    class FakeComm(object):
        def __init__(self, rank, world):
            self.rank, self.world, self.handle, self.sharded, self.device = rank, world, None, False, torch.device("cpu")

    def msgs(arr):
        out = []
        for q in range(4):
            cnt = arr[7 * q]
            out.append([(arr[7 * q + 1 + k], arr[7 * q + 4 + k]) for k in range(cnt)])
        return out        # to_upper, to_lower, from_lower, from_upper

    nx, ny = 12, 48

    def global_rows(sh, kind, segs):
        """(component, whole-grid row) of every row a list of (offset, length) segments of a stored array covers"""
        rows = []
        for off, ln in segs:
            if kind == "msgs_cells":
                assert off % nx == 0 and ln % nx == 0
                rows += [("c", (sh.cb + off // nx + k) % ny) for k in range(ln // nx)]
                continue
            first_u = kind == "msgs_faces"
            n_first = sh.n_u if first_u else sh.n_v
            comp = ("u" if first_u else "v") if off < n_first else ("v" if first_u else "u")
            o = off if off < n_first else off - n_first
            w = nx + 1 if comp == "u" else nx
            assert o % w == 0 and ln % w == 0
            base, period = (sh.cb, ny) if comp == "u" else (sh.vb, ny + 1)
            rows += [(comp, (base + o // w + k) % period) for k in range(ln // w)]
        return rows

    for world in (2, 4, 8):
        sh = [StepSharding(FakeComm(r, world), nx, ny) for r in range(world)]
        for r in range(world):
            up, lo = (r + 1) % world, (r - 1) % world
            for kind in ("msgs_faces", "msgs_faces_vfirst", "msgs_cells"):
                mine, theirs_up, theirs_lo = msgs(getattr(sh[r], kind)), msgs(getattr(sh[up], kind)), msgs(getattr(sh[lo], kind))
                assert [n for _, n in mine[0]] == [n for _, n in theirs_up[2]] and [n for _, n in mine[1]] == [n for _, n in theirs_lo[3]]
                # what I send up is what my upper neighbour receives from below, row for row of the whole grid; the same downwards
                assert global_rows(sh[r], kind, mine[0]) == global_rows(sh[up], kind, theirs_up[2]), (world, r, kind)
                assert global_rows(sh[r], kind, mine[1]) == global_rows(sh[lo], kind, theirs_lo[3]), (world, r, kind)
                # what a rank sends are rows it owns; what it receives are rows it stores and does not own
                own_u, own_v = range(sh[r].j0, sh[r].j1), range(sh[r].j0, sh[r].j1 + (1 if sh[r].last else 0))
                for comp, j in global_rows(sh[r], kind, mine[0]) + global_rows(sh[r], kind, mine[1]):
                    assert j in (own_v if comp == "v" else own_u)
                for comp, j in global_rows(sh[r], kind, mine[2]) + global_rows(sh[r], kind, mine[3]):
                    assert j not in (own_v if comp == "v" else own_u)
        # the last slab sends its HALO last v rows AND the duplicate row v[ny] upwards (across the seam to rank 0)
        top = global_rows(sh[world - 1], "msgs_faces", msgs(sh[world - 1].msgs_faces)[0])
        assert [j for c, j in top if c == "v"] == [ny - 2, ny - 1, ny] and HALO == 2
        # scatter / owned round trip: every face and cell of the whole grid is owned by exactly one rank, with its own value
        st = np.arange((ny + 1) * (nx + 1) * 2, dtype=np.float32).reshape(1, ny + 1, nx + 1, 2)
        cells = np.arange(ny * nx, dtype=np.float32).reshape(1, ny, nx, 1)
        got_u, got_v, got_c = [], [], []
        for r in range(world):
            flat = sh[r].scatter_staggered(st)
            assert flat.numel() == sh[r].n_faces
            u, v = sh[r].owned_faces(flat)
            got_u.append(u.numpy()); got_v.append(v.numpy())
            got_c.append(sh[r].owned_cells(sh[r].scatter_cells(cells)).numpy())
            # the stored rows next to the owned ones are the ring neighbours' (first slab below: v[ny - 2], v[ny - 1], v[ny])
            vs = flat[sh[r].n_u:].view(sh[r].vr, nx).numpy()
            np.testing.assert_array_equal(vs[0], st[0, (sh[r].j0 - 3) % (ny + 1), :nx, 0])
            np.testing.assert_array_equal(vs[-1], st[0, (sh[r].j1 + 2) % (ny + 1), :nx, 0])
        np.testing.assert_array_equal(np.concatenate(got_u), st[0, :ny, :, 1])
        np.testing.assert_array_equal(np.concatenate(got_v), st[0, :, :nx, 0])
        np.testing.assert_array_equal(np.concatenate(got_c), cells[0, :, :, 0])
        for r in range(world):
            z = sh[r].sizes(True, True)
            assert z["mask_rows"] == sh[r].mr and z["nnz_u"] == 5 * sh[r].n_u and z["nnz_v"] == 5 * sh[r].n_v      # (periodic: 5 entries per row)
            z = sh[r].sizes(False, False)
            assert 0 < z["nnz_u"] <= 5 * sh[r].n_u and 0 < z["nnz_v"] <= 5 * sh[r].n_v


def test_step_sharding_caches_follow_the_content_not_only_the_identity():
    """The sharded step's caches (StepSharding.cached_scatter_* / sim_tensors) are keyed like grids.device_constant: identity AND a
    content stamp.  An array updated in place (a dirichlet_update_fn returning the same numpy array, a tensor edited between steps) is
    cut again; a new sim object never finds another object's masks even if id() is reused; the CSR pattern is kept per periodicity."""
    import numpy as np
    import torch
    from diffpiso.sharding import StepSharding

    class FakeComm(object):
        def __init__(self, rank, world):
            self.rank, self.world, self.handle, self.sharded, self.device = rank, world, None, False, torch.device("cpu")

    nx, ny = 12, 16
    sh = StepSharding(FakeComm(1, 2), nx, ny)
    dv = np.zeros((1, ny + 1, nx + 1, 2), np.float32)
    a = sh.cached_scatter_staggered(dv)
    assert sh.cached_scatter_staggered(dv) is a                                   # same object, same content: the cached rows
    dv[0, 10, :, 1] = 3.0                                                         # updated IN PLACE
    b = sh.cached_scatter_staggered(dv)
    assert b is not a and float(b.abs().max()) == 3.0
    np.testing.assert_array_equal(b.numpy(), sh.scatter_staggered(dv).numpy())
    t = torch.zeros((nx + 1) * ny + nx * (ny + 1))
    f0 = sh.cached_scatter_faces(t)
    assert sh.cached_scatter_faces(t) is f0
    t[5] = 1.0                                                                    # tensor._version moves
    assert sh.cached_scatter_faces(t) is not f0

    class Sim(object):
        def __init__(self, fill):
            self.active_mask = np.full((1, ny + 2, nx + 2, 1), fill, np.float32)
            self.accessible_mask = np.ones((1, ny + 2, nx + 2, 1), np.float32)
            self.dirichlet_mask = np.zeros((1, ny + 1, nx + 1, 2), bool)
            self.no_slip_mask = None

    s1 = Sim(1.0)
    c1 = sh.sim_tensors(s1, torch.device("cpu"))
    assert sh.sim_tensors(s1, torch.device("cpu")) is c1
    s1.active_mask[0, 9, :, 0] = 0.0                                              # in place
    c1b = sh.sim_tensors(s1, torch.device("cpu"))
    assert c1b is not c1 and float(c1b["active"].min()) == 0.0
    # an entry only answers for the object it was made from (a recycled id() must not return another simulation's masks)
    key = (id(s1), str(torch.device("cpu")))
    s2 = Sim(0.5)
    sh._sim_cache[(id(s2), str(torch.device("cpu")))] = sh._sim_cache[key]       # what a recycled id would look like
    assert float(sh.sim_tensors(s2, torch.device("cpu"))["active"].max()) == 0.5
    # CSR pattern: one entry per periodicity, never another periodicity's
    assert sh.pattern_for(True, True) is None
    rp = torch.arange(sh.n_u + sh.n_v + 2, dtype=torch.int32)
    sh.set_pattern(torch.zeros(4, dtype=torch.int32), rp, 7, per_xy=(True, True), nnz=(7, 9))
    assert sh.pattern_for(True, False) is None and sh.pattern_for(True, True) is not None and sh.nnz == (7, 9)


def test_flat_faces_shortcut_only_while_the_components_are_views_of_the_flat_vector():
    import torch
    import diffpiso as dp
    from diffpiso import fused

    class Geom(object):
        sh, nx, ny = None, 5, 4
        n_u = 4 * 6

    flat = torch.arange(4 * 6 + 5 * 5, dtype=torch.float32)
    grid = fused.faces_to_grid(flat, Geom, None, "boundary")
    assert fused.flat_faces(grid) is flat
    grid.data[1].data = grid.data[1].data * 2.0                                   # user code rebinds a component between two steps
    got = fused.flat_faces(grid)
    assert got is not flat
    assert torch.equal(got[:Geom.n_u], 2.0 * flat[:Geom.n_u]) and torch.equal(got[Geom.n_u:], flat[Geom.n_u:])
