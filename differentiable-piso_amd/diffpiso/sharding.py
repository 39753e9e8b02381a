"""The slab-decomposed STEP (new design, the reference is single-GPU) -- host side.

`distributed.py` cuts the two linear solvers into y-slabs; this module cuts everything else of `piso_step`: matrix assembly, the
stencil glue (forward and reverse mode), the Laplacian and the CSR product work on the face / cell rows of the rank's slab only
(`piso_set_row_window`, csrc/*.hip), and nothing is all-gathered.  Arrays stay GLOBALLY indexed on every rank (288 GB of HBM: a
2048 x 16384 face vector is 268 MB); a rank's copy is valid on its own rows plus, after `halo_*`, on two rows either side of
them in the ring -- which is what the gather kernels (pressure gradient, divergence, CSR product and their reference adjoints,
including the divergence adjoint's dc[n-2] quirk and the v[ny] / v[1] seam of the advection matrix) read.

Every kernel of the step is "one thread per OUTPUT element, gathering its inputs": in reverse mode the incoming cotangents are
halo-filled before the adjoint kernel gathers from them, so a halo exchange never needs an adjoint of its own.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N

HALO = 2      # rows either side: the widest stencil of the step (padded v of the last slab reads v[1]; the periodic divergence adjoint dc[n-2])


def _msg(segments):
    """One message: up to three (offset, length) element segments -> 7 ints {count, off[3], len[3]}."""
    segments = [(int(o), int(n)) for o, n in segments if n > 0]
    assert len(segments) <= 3
    off = [o for o, _ in segments] + [0] * (3 - len(segments))
    ln = [n for _, n in segments] + [0] * (3 - len(segments))
    return [len(segments)] + off + ln


class StepSharding(object):
    """Row window + halo messages of one rank for an nx x ny grid cut into `comm.world` y-slabs."""

    def __init__(self, comm, nx, ny):
        world, rank = comm.world, comm.rank
        if ny % world != 0 or ny // world < 2 * HALO:
            raise ValueError("slab-decomposed step: ny (%d) must be divisible by the ranks (%d) with at least %d rows per slab"
                             % (ny, world, 2 * HALO))
        self.comm, self.nx, self.ny, self.world, self.rank = comm, int(nx), int(ny), world, rank
        nyl = ny // world
        self.j0, self.j1, self.last = rank * nyl, (rank + 1) * nyl, rank == world - 1
        self.nyl = nyl
        w = HALO
        lower, upper = (rank - 1) % world, (rank + 1) % world
        jl1 = lower * nyl + nyl                      # end row of the ring-lower slab
        ju0 = upper * nyl                            # first row of the ring-upper slab
        lower_last = 1 if lower == world - 1 else 0
        mine_last = 1 if self.last else 0
        # (row interval) per message for u rows, v rows and cell rows
        self._rows = dict(
            to_upper=dict(u=(self.j1 - w, self.j1), v=(self.j1 - w, self.j1 + mine_last), c=(self.j1 - w, self.j1)),
            to_lower=dict(u=(self.j0, self.j0 + w), v=(self.j0, self.j0 + w), c=(self.j0, self.j0 + w)),
            from_lower=dict(u=(jl1 - w, jl1), v=(jl1 - w, jl1 + lower_last), c=(jl1 - w, jl1)),
            from_upper=dict(u=(ju0, ju0 + w), v=(ju0, ju0 + w), c=(ju0, ju0 + w)))
        nxu, n_u, n_v = nx + 1, (nx + 1) * ny, nx * (ny + 1)
        order = ("to_upper", "to_lower", "from_lower", "from_upper")

        def pack(fn):
            flat = []
            for key in order:
                flat += _msg(fn(self._rows[key]))
            return (C.c_int * 28)(*flat)
        self.msgs_faces = pack(lambda r: [(r["u"][0] * nxu, (r["u"][1] - r["u"][0]) * nxu),
                                          (n_u + r["v"][0] * nx, (r["v"][1] - r["v"][0]) * nx)])
        self.msgs_faces_vfirst = pack(lambda r: [(r["v"][0] * nx, (r["v"][1] - r["v"][0]) * nx),
                                                 (n_v + r["u"][0] * nxu, (r["u"][1] - r["u"][0]) * nxu)])
        self.msgs_cells = pack(lambda r: [(r["c"][0] * nx, (r["c"][1] - r["c"][0]) * nx)])
        self.msgs_csr = None                         # needs the row pointers: set by `set_pattern`
        self.pattern = None                          # (col_indices, row_pointers) of the whole grid: geometry only, built once
        self.exchanges = 0
        comm.sharded = True
        # (the row window itself is named by every kernel wrapper at call time: _native.use_window)

    # ------------------------------------------------------------------------------------------------ pattern of the two matrices
    def set_pattern(self, col_indices, row_pointers, nnz_u):
        """col / rowptr of the WHOLE grid (one un-windowed assembly at set-up: they depend on the geometry only) and, from the
        row pointers, the segments of the value array that hold a block of face rows."""
        nx, ny = self.nx, self.ny
        nxu, n_u = nx + 1, (nx + 1) * ny
        rp = row_pointers.cpu().numpy().astype(np.int64)
        rp_u, rp_v = rp[:n_u + 1], rp[n_u + 1:]

        def segs(r):
            (ua, ub), (va, vb) = r["u"], r["v"]
            return [(rp_u[ua * nxu], rp_u[ub * nxu] - rp_u[ua * nxu]),
                    (nnz_u + rp_v[va * nx], rp_v[vb * nx] - rp_v[va * nx])]
        flat = []
        for key in ("to_upper", "to_lower", "from_lower", "from_upper"):
            flat += _msg(segs(self._rows[key]))
        self.msgs_csr = (C.c_int * 28)(*flat)
        self.pattern = (col_indices, row_pointers)

    # ------------------------------------------------------------------------------------------------ halo exchanges (in place)
    def _exchange(self, t, msgs):
        if self.world == 1:
            return t
        if not t.is_contiguous():
            raise N.PisoNativeError("halo exchange of a non-contiguous tensor")
        code = {torch.float32: 0, torch.float64: 1, torch.int32: 2}[t.dtype]
        N.check(N.lib.piso_comm_exchange(self.comm.handle, N.ptr(t), code, msgs, N.stream_ptr()), "piso_comm_exchange")
        self.exchanges += 1
        return t

    def halo_faces(self, t):
        """Flat u-first face vector: two rows below / above the slab (incl. the duplicate row v[ny] across the seam)."""
        return self._exchange(t, self.msgs_faces)

    def halo_faces_vfirst(self, t):
        return self._exchange(t, self.msgs_faces_vfirst)

    def halo_cells(self, t):
        return self._exchange(t, self.msgs_cells)

    def halo_csr_values(self, t):
        return self._exchange(t, self.msgs_csr)

    # ------------------------------------------------------------------------------------------------ helpers for callers
    def owned_mask_staggered(self, device):
        """[1, ny+1, nx+1, 2] float mask of the faces this rank owns (channel 0 = v rows [j0, j1 + last), channel 1 = u rows)."""
        m = torch.zeros((1, self.ny + 1, self.nx + 1, 2), dtype=torch.float32, device=device)
        m[0, self.j0:self.j1 + (1 if self.last else 0), :self.nx, 0] = 1
        m[0, self.j0:self.j1, :, 1] = 1
        return m

    def owned_mask_cells(self, device):
        m = torch.zeros((1, self.ny, self.nx, 1), dtype=torch.float32, device=device)
        m[0, self.j0:self.j1] = 1
        return m

    def check(self):
        """Raise if a wait on a peer gave up since the last check (agreed over the ranks; synchronises the stream)."""
        N.check(N.lib.piso_comm_check(self.comm.handle, N.stream_ptr()), "piso_comm_check")

    def close(self):
        N.use_window(None)
        self.comm.sharded = False
