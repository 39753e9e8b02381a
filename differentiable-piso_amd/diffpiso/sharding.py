"""The slab-decomposed STEP (new design, the reference is single-GPU) -- host side.  Round 5: LOCAL storage.

`distributed.py` cuts the two linear solvers into y-slabs; this module cuts everything else of `piso_step`: matrix assembly, the
stencil glue (forward and reverse mode), the Laplacian and the CSR product work on the face / cell rows of the rank's slab
(the `*_slab` entry points of libpiso_hip.so, include/piso_hip.h: piso_slab_t) and nothing is all-gathered.  A rank STORES its own
rows plus the rows either side that its gathers read, and nothing else - every tensor of a sharded step has 1 / ranks of the grid's
size (plus the halo rows), the library's workspaces included:

  cells, u faces   the ring rows [j0 - 2, j1 + 2)   (mod ny)                                     nyl + 4 rows
  v faces          the ring rows [j0 - 3, j1 + 3)   of the ring v[0] .. v[ny - 1], v[ny]        nyl + 6 rows
                   (the duplicate row v[ny] is a row of its own between v[ny - 1] and v[0]: the padded v of the last slab reads
                   v[0] and v[1], the first slab's lower halo is v[ny - 2], v[ny - 1], v[ny])
  flat face vectors = the stored u rows followed by the stored v rows ("v-first": v, then u)

The index arithmetic inside the kernels stays the whole grid's (boundary rules, seams, the reference's adjoint quirks are decided on
global (i, j)); only "where does row j live" goes through the rank's row map.  Every kernel of the step is "one thread per OUTPUT
element, gathering its inputs": in reverse mode the incoming cotangents are halo-filled before the adjoint kernel gathers from them,
so a halo exchange never needs an adjoint of its own.

Fields of a sharded simulation are `SlabStaggered` / `SlabCentered` (this module): what `piso_step` / `unroll_piso_steps` take and
return instead of StaggeredGrid / CenteredGrid when `simulation_physics.sharding` is set.  `StepSharding.scatter_*` cut a whole-grid
array (host or device) into the rank's stored rows, `owned_*` pick the rows a rank owns out of a stored array.
"""
import ctypes as C
import zlib

import numpy as np
import torch

from . import _native as N
from .grids import AABox, as_tensor

HALO = 2      # u / cell rows either side: the widest stencil of the step (the periodic divergence adjoint's dc[n-2]; v: 3, see above)


def _msg(segments):
    """One message: up to three (offset, length) element segments -> 7 ints {count, off[3], len[3]}."""
    segments = [(int(o), int(n)) for o, n in segments if n > 0]
    assert len(segments) <= 3
    off = [o for o, _ in segments] + [0] * (3 - len(segments))
    ln = [n for _, n in segments] + [0] * (3 - len(segments))
    return [len(segments)] + off + ln


class StepSharding(object):
    """Row map + halo messages of one rank for an nx x ny grid cut into `comm.world` y-slabs (local storage)."""

    def __init__(self, comm, nx, ny):
        world, rank = comm.world, comm.rank
        nyl = ny // max(world, 1)
        if world < 2 or ny % world != 0 or nyl < 2 * HALO or nyl + 6 > ny:
            raise ValueError("slab-decomposed step: at least two ranks, ny (%d) divisible by the ranks (%d), at least %d rows per slab"
                             % (ny, world, 2 * HALO))
        self.comm, self.nx, self.ny, self.world, self.rank = comm, int(nx), int(ny), world, rank
        self.j0, self.j1, self.last = rank * nyl, (rank + 1) * nyl, rank == world - 1
        self.nyl = nyl
        self.slab = N.Slab(int(ny), int(self.j0), int(self.j1), 1 if self.last else 0)
        self.slab_ptr = C.pointer(self.slab)
        # stored rows (csrc/piso_common.h: RowMap)
        self.cb, self.cr = (self.j0 - 2) % ny, nyl + 4
        self.vb, self.vr = (self.j0 - 3) % (ny + 1), nyl + 6
        self.mb, self.mr = self.j0, nyl + 3
        self.n_u, self.n_v = self.cr * (nx + 1), self.vr * nx
        self.n_faces, self.n_cells = self.n_u + self.n_v, self.cr * nx
        self.n_pad = (nyl + 2) * (nx + 3) + (nyl + 3) * (nx + 2)
        w = HALO
        lower, upper = (rank - 1) % world, (rank + 1) % world
        jl1 = lower * nyl + nyl                      # end row of the ring-lower slab
        ju0 = upper * nyl                            # first row of the ring-upper slab
        lower_last = 1 if lower == world - 1 else 0
        mine_last = 1 if self.last else 0
        # (row interval) per message for u rows, v rows and cell rows - whole-grid row numbers; the rows of an interval follow each
        # other around the ring, so they are neighbours in the stored arrays as well
        self._rows = dict(
            to_upper=dict(u=(self.j1 - w, self.j1), v=(self.j1 - w, self.j1 + mine_last), c=(self.j1 - w, self.j1)),
            to_lower=dict(u=(self.j0, self.j0 + w), v=(self.j0, self.j0 + w), c=(self.j0, self.j0 + w)),
            from_lower=dict(u=(jl1 - w, jl1), v=(jl1 - w, jl1 + lower_last), c=(jl1 - w, jl1)),
            from_upper=dict(u=(ju0, ju0 + w), v=(ju0, ju0 + w), c=(ju0, ju0 + w)))
        nxu = nx + 1
        self._order = ("to_upper", "to_lower", "from_lower", "from_upper")

        def pack(fn):
            flat = []
            for key in self._order:
                flat += _msg(fn(self._rows[key]))
            return (C.c_int * 28)(*flat)
        self.msgs_faces = pack(lambda r: [(self.urow(r["u"][0]) * nxu, (r["u"][1] - r["u"][0]) * nxu),
                                          (self.n_u + self.vrow(r["v"][0]) * nx, (r["v"][1] - r["v"][0]) * nx)])
        self.msgs_faces_vfirst = pack(lambda r: [(self.vrow(r["v"][0]) * nx, (r["v"][1] - r["v"][0]) * nx),
                                                 (self.n_v + self.urow(r["u"][0]) * nxu, (r["u"][1] - r["u"][0]) * nxu)])
        self.msgs_cells = pack(lambda r: [(self.urow(r["c"][0]) * nx, (r["c"][1] - r["c"][0]) * nx)])
        self.msgs_csr = None                         # needs the row pointers: set by `set_pattern`
        self.pattern = None                          # (col_indices, row_pointers) of the STORED rows: geometry only, built once
        self.nnz = None                              # stored CSR entries (u matrix, v matrix)
        self.exchanges = 0
        self._sim_cache = {}
        self._scatter_cache = {}
        self._patterns = {}                          # (per_x, per_y) -> (pattern, msgs_csr, nnz)
        self.periodic_xy = (False, False)            # (x, y) periodicity of the velocity: set by the step (the CSR numbering needs it)
        comm.sharded = True
        comm.step_sharding = self                    # the solvers reach the row map through their communicator

    # ------------------------------------------------------------------------------------------------ the row map
    def urow(self, j):
        """Stored row of whole-grid u / cell row j."""
        return (int(j) - self.cb) % self.ny

    def vrow(self, j):
        return (int(j) - self.vb) % (self.ny + 1)

    def sizes(self, per_x, per_y):
        out = (C.c_int * 8)()
        N.check(N.lib.piso_slab_sizes(self.slab_ptr, self.nx, self.ny, int(per_x), int(per_y), out), "piso_slab_sizes")
        assert (out[0], out[1], out[2], out[3]) == (self.cr, self.vr, self.n_u, self.n_v) and out[7] == self.n_pad
        return dict(nnz_u=int(out[4]), nnz_v=int(out[5]), mask_rows=int(out[6]))

    # ------------------------------------------------------------------------------------------------ pattern of the two matrices
    def pattern_for(self, per_x, per_y):
        """The cached (col, rowptr, msgs_csr, (nnz_u, nnz_v)) of this periodicity, or None.  The stored nnz and the segments of the value
        array depend on (per_x, per_y): one entry per periodicity the sharding has been stepped with - never a pattern of another one."""
        hit = self._patterns.get((bool(per_x), bool(per_y)))
        if hit is not None:
            self.pattern, self.msgs_csr, self.nnz = hit[0], hit[1], hit[2]
        return hit

    def set_pattern(self, col_indices, row_pointers, nnz_u, per_xy=None, nnz=None):
        """col / rowptr of the STORED rows (one pattern-only assembly per periodicity: they depend on the geometry only) and, from the
        row pointers, the segments of the stored value array that hold a block of face rows."""
        nx = self.nx
        nxu = nx + 1
        rp = row_pointers.cpu().numpy().astype(np.int64)
        rp_u, rp_v = rp[:self.n_u + 1], rp[self.n_u + 1:]

        def segs(r):
            (ua, ub), (va, vb) = r["u"], r["v"]
            la, lb = self.urow(ua) * nxu, self.urow(ua) * nxu + (ub - ua) * nxu
            ma, mb = self.vrow(va) * nx, self.vrow(va) * nx + (vb - va) * nx
            return [(rp_u[la], rp_u[lb] - rp_u[la]), (nnz_u + rp_v[ma], rp_v[mb] - rp_v[ma])]
        flat = []
        for key in self._order:
            flat += _msg(segs(self._rows[key]))
        self.msgs_csr = (C.c_int * 28)(*flat)
        self.pattern = (col_indices, row_pointers)
        if nnz is not None:
            self.nnz = tuple(nnz)
        if per_xy is not None:
            self._patterns[(bool(per_xy[0]), bool(per_xy[1]))] = (self.pattern, self.msgs_csr, self.nnz)

    # ------------------------------------------------------------------------------------------------ halo exchanges (in place)
    def _exchange(self, t, msgs):
        if not t.is_contiguous():
            raise N.PisoNativeError("halo exchange of a non-contiguous tensor")
        code = {torch.float32: 0, torch.float64: 1, torch.int32: 2}[t.dtype]
        N.check(N.lib.piso_comm_exchange(self.comm.handle, N.ptr(t), code, msgs, N.stream_ptr()), "piso_comm_exchange")
        self.exchanges += 1
        return t

    def halo_faces(self, t):
        """Flat u-first face vector (stored rows): two rows below / above the slab (incl. the duplicate row v[ny] across the seam)."""
        assert t.numel() == self.n_faces
        return self._exchange(t, self.msgs_faces)

    def halo_faces_vfirst(self, t):
        assert t.numel() == self.n_faces
        return self._exchange(t, self.msgs_faces_vfirst)

    def halo_cells(self, t):
        assert t.numel() == self.n_cells
        return self._exchange(t, self.msgs_cells)

    def halo_csr_values(self, t):
        return self._exchange(t, self.msgs_csr)

    # ------------------------------------------------------------------------------------------------ whole grid <-> stored rows
    def _row_index(self, base, rows, period, device):
        return ((torch.arange(rows, device=device) + base) % period)

    def scatter_staggered(self, tensor, dtype=torch.float32, device=None):
        """Whole-grid staggered tensor [1, ny + 1, nx + 1, 2] (numpy / host / device) -> this rank's flat u-first face vector."""
        t = torch.as_tensor(tensor) if not isinstance(tensor, torch.Tensor) else tensor
        dev = device if device is not None else self.comm.device
        ny, nx = self.ny, self.nx
        u = t[0, :ny, :, 1].index_select(0, self._row_index(self.cb, self.cr, ny, t.device))
        v = t[0, :, :nx, 0].index_select(0, self._row_index(self.vb, self.vr, ny + 1, t.device))
        return torch.cat([u.reshape(-1), v.reshape(-1)]).to(device=dev, dtype=dtype).contiguous()

    def scatter_faces(self, flat, dtype=None, device=None):
        """Whole-grid flat u-first face vector -> stored rows."""
        t = torch.as_tensor(flat) if not isinstance(flat, torch.Tensor) else flat
        dev = device if device is not None else self.comm.device
        ny, nx = self.ny, self.nx
        n_u = (nx + 1) * ny
        u = t[:n_u].reshape(ny, nx + 1).index_select(0, self._row_index(self.cb, self.cr, ny, t.device))
        v = t[n_u:].reshape(ny + 1, nx).index_select(0, self._row_index(self.vb, self.vr, ny + 1, t.device))
        out = torch.cat([u.reshape(-1), v.reshape(-1)])
        return out.to(device=dev, dtype=dtype if dtype is not None else out.dtype).contiguous()

    def scatter_cells(self, tensor, dtype=torch.float32, device=None):
        """Whole-grid cell array [1, ny, nx, 1] (or [ny, nx]) -> [1, stored rows, nx, 1]."""
        t = torch.as_tensor(tensor) if not isinstance(tensor, torch.Tensor) else tensor
        dev = device if device is not None else self.comm.device
        t = t.reshape(self.ny, self.nx)
        c = t.index_select(0, self._row_index(self.cb, self.cr, self.ny, t.device))
        return c.to(device=dev, dtype=dtype).reshape(1, self.cr, self.nx, 1).contiguous()

    def scatter_mask(self, tensor, dtype=torch.float32, device=None):
        """Whole-grid padded cell mask [1, ny + 2, nx + 2, 1] (any shape with (ny + 2)(nx + 2) elements) -> the stored mask rows, flat."""
        t = torch.as_tensor(tensor) if not isinstance(tensor, torch.Tensor) else tensor
        dev = device if device is not None else self.comm.device
        t = t.reshape(self.ny + 2, self.nx + 2)
        hi = min(self.mb + self.mr, self.ny + 2)
        out = torch.zeros((self.mr, self.nx + 2), dtype=t.dtype, device=t.device)
        out[:hi - self.mb] = t[self.mb:hi]
        return out.to(device=dev, dtype=dtype).reshape(-1).contiguous()

    def owned_faces(self, flat):
        """(u rows [nyl, nx + 1], v rows [nyl + last, nx]) this rank owns, as views of a stored flat face vector."""
        nx, nyl = self.nx, self.nyl
        u = flat[:self.n_u].view(self.cr, nx + 1)[2:2 + nyl]
        v = flat[self.n_u:].view(self.vr, nx)[3:3 + nyl + (1 if self.last else 0)]
        return u, v

    def owned_cells(self, cells):
        return cells.reshape(self.cr, self.nx)[2:2 + self.nyl]

    def owned_sum_of_squares(self, flat):
        u, v = self.owned_faces(flat)
        return (u ** 2).sum() + (v ** 2).sum()

    # ------------------------------------------------------------------------------------------------ the simulation's constants, stored rows only
    def sim_tensors(self, sim, device):
        """active / accessible masks, Dirichlet mask, no-slip mask of `sim` cut to this rank's stored rows.  Cached per sim OBJECT (the
        entry holds the object: an id() reused after a collection cannot hit) and per content stamp of its four arrays: replaced arrays and
        in-place edits of arrays up to 1 MB (tensors: of any size) cut them again; larger numpy masks are constants of the sim object,
        as they are on the one-GPU path (SimulationParameters._cached keys its device copies by the array's identity)."""
        key = (id(sim), str(device))
        stamp = tuple(_stamp(a, big="identity") for a in (sim.active_mask, sim.accessible_mask, sim.dirichlet_mask, sim.no_slip_mask))
        hit = self._sim_cache.get(key)
        if hit is not None and hit[0] is sim and hit[1] == stamp:
            return hit[2]
        if len(self._sim_cache) > 8:
            self._sim_cache.clear()
        c = dict(active=self.scatter_mask(_host(sim.active_mask), torch.float32, device),
                 accessible=self.scatter_mask(_host(sim.accessible_mask), torch.float32, device),
                 dmask=self.scatter_staggered(_host(sim.dirichlet_mask).astype(np.float32), torch.float32, device).ne(0).to(torch.uint8).contiguous(),
                 no_slip=None)
        if sim.no_slip_mask is not None:
            c["no_slip"] = self.scatter_mask(_host(sim.no_slip_mask).astype(np.float32), torch.float32, device).ne(0).to(torch.uint8).contiguous()
        self._sim_cache[key] = (sim, stamp, c)
        return c

    def _cached(self, kind, obj, cut):
        """`cut(obj)` once per object AND content: identity + checksum for numpy arrays, identity + tensor._version for tensors (an
        array updated in place - a dirichlet_update_fn that returns the same array, a viscosity field edited between steps - is cut
        again; the one-GPU path re-uploads in exactly these cases, grids.device_constant).  Numpy arrays above 1 MB are not cached."""
        key = (kind, id(obj))
        stamp = _stamp(obj)
        if stamp is None:                              # a numpy array above 1 MB: cut every time (the one-GPU path uploads it every time)
            return cut(obj)
        hit = self._scatter_cache.get(key)
        if hit is None or hit[0] is not obj or hit[1] != stamp:
            if len(self._scatter_cache) > 32:
                self._scatter_cache.clear()
            hit = (obj, stamp, cut(obj))
            self._scatter_cache[key] = hit
        return hit[2]

    def cached_scatter_staggered(self, tensor):
        """scatter_staggered of a constant of the simulation (Dirichlet values)."""
        return self._cached("st", tensor, lambda t: self.scatter_staggered(_host(t).astype(np.float32)))

    def cached_scatter_faces(self, flat):
        return self._cached("fl", flat, lambda f: self.scatter_faces(torch.as_tensor(_host(f).reshape(-1)), dtype=torch.float32))

    # ------------------------------------------------------------------------------------------------ fields
    def staggered_grid(self, flat, box, extrapolation):
        return SlabStaggered(flat, self, box, extrapolation)

    def centered_grid(self, data, box, extrapolation):
        return SlabCentered(data, self, box, extrapolation)

    def check(self):
        """Raise if a wait on a peer gave up since the last check (agreed over the ranks; synchronises the stream)."""
        N.check(N.lib.piso_comm_check(self.comm.handle, N.stream_ptr()), "piso_comm_check")

    def close(self):
        self.comm.sharded = False
        self.comm.step_sharding = None


def _stamp(x, big=None):
    """What changes when the CONTENT of a cached input changes: tensor._version for tensors (bumped by every in-place op), an adler32
    checksum for numpy arrays up to 1 MB (the bound of grids.device_constant; ~1 ms per MB).  Above that: None ("do not cache"), or with
    big="identity" the array's address and shape (constants of a simulation object)."""
    if x is None:
        return ("none",)
    if isinstance(x, torch.Tensor):
        return ("v", x._version, tuple(x.shape), x.data_ptr())
    a = np.asarray(x)
    if a.nbytes > (1 << 20):
        return ("i", a.__array_interface__["data"][0], a.shape, str(a.dtype)) if big == "identity" else None
    return ("c", zlib.adler32(a if a.flags.c_contiguous else np.ascontiguousarray(a)), a.shape, str(a.dtype))


def _host(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


class SlabStaggered(object):
    """A staggered field of a slab-decomposed simulation: the rank's stored face rows as ONE flat u-first vector.  Quacks like the
    StaggeredGrid that `piso_step` / `unroll_piso_steps` take and return; `resolution`, `dx` and `box` are the WHOLE grid's."""

    def __init__(self, flat, sharding, box=None, extrapolation=None):
        if flat.numel() != sharding.n_faces:
            raise ValueError("SlabStaggered: %d elements, the rank stores %d" % (flat.numel(), sharding.n_faces))
        self.flat, self.sharding, self.extrapolation = flat.reshape(-1), sharding, extrapolation
        self.box = AABox.to_box(box, resolution_hint=[sharding.ny, sharding.nx]) if box is not None else None

    @property
    def resolution(self):
        return np.array([self.sharding.ny, self.sharding.nx])

    @property
    def dx(self):
        return self.box.size / self.resolution

    @property
    def device(self):
        return self.flat.device

    def staggered_tensor(self):
        return self.flat

    def rewrap(self, flat):
        return SlabStaggered(flat, self.sharding, self.box, self.extrapolation)


class SlabCentered(object):
    """A cell field of a slab-decomposed simulation: [1, stored rows, nx, 1]."""

    def __init__(self, data, sharding, box=None, extrapolation=None):
        if data.numel() != sharding.n_cells:
            raise ValueError("SlabCentered: %d elements, the rank stores %d" % (data.numel(), sharding.n_cells))
        self.data, self.sharding, self.extrapolation = data.reshape(1, sharding.cr, sharding.nx, 1), sharding, extrapolation
        self.box = AABox.to_box(box, resolution_hint=[sharding.ny, sharding.nx]) if box is not None else None

    @property
    def resolution(self):
        return np.array([self.sharding.ny, self.sharding.nx])

    @property
    def dx(self):
        return self.box.size / self.resolution

    def rewrap(self, data):
        return SlabCentered(data, self.sharding, self.box, self.extrapolation)

    def __add__(self, other):
        return self.rewrap(self.data + (other.data if isinstance(other, SlabCentered) else other))
