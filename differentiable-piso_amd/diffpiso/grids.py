"""Minimal torch re-creation of the PhiFlow 1.4 surface the PISO path touches.

The reference builds its fields with its vendored PhiFlow (StaggeredGrid, CenteredGrid, Domain, box, OPEN / CLOSED /
PERIODIC ...).  Only the parts `piso_step` and the driver scripts use are provided here, with the same names, argument
meaning and memory layout:

  StaggeredGrid.staggered_tensor() -> [1, Ny+1, Nx+1, 2], channel 0 = v, channel 1 = u, unused last column / row zero
                                       (PhiFlow/phi/physics/field/staggered_grid.py:33-46, :208-210)
  CenteredGrid.data                -> [1, Ny, Nx, C]
  Domain(resolution=[Ny, Nx], boundaries, box)   (PhiFlow/phi/physics/domain.py:13-209)
  Material extrapolation modes                   (PhiFlow/phi/physics/material.py:70-108)

Everything is a thin wrapper around torch tensors (device memory); no general field algebra is implemented.
"""
import numpy as np
import torch


def default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


_constant_uploads = {}


def device_constant(x, dtype=None, device=None):
    """as_tensor for inputs a caller hands over again every step (the reference's scripts pass `sim.dirichlet_values`, a numpy
    array, to every piso_step: lid_driven_cavity_2d.py:57-61).  A pageable host-to-device copy is stream-ordered - it waits for
    every kernel queued before it - so a small numpy array is uploaded ONCE and found again by identity + checksum (an array
    modified in place is uploaded again); tensors and large arrays go through as_tensor."""
    if isinstance(x, np.ndarray) and 0 < x.nbytes <= (1 << 20) and device is not None:
        import zlib
        key = (id(x), x.shape, str(x.dtype), str(dtype), str(device))
        crc = zlib.adler32(x if x.flags.c_contiguous else np.ascontiguousarray(x))
        hit = _constant_uploads.get(key)
        if hit is not None and hit[0] == crc:
            return hit[1]
        if len(_constant_uploads) > 32:
            _constant_uploads.clear()
        t = as_tensor(x, dtype=dtype, device=device)
        _constant_uploads[key] = (crc, t)
        return t
    return as_tensor(x, dtype=dtype, device=device)


def as_tensor(x, dtype=None, device=None):
    """numpy / python / torch -> torch tensor on the working device (no copy if already there)."""
    if isinstance(x, torch.Tensor):
        t = x
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        if device is not None and t.device != torch.device(device):
            t = t.to(device)
        return t
    a = np.asarray(x)
    if a.dtype == np.float64 and dtype is None:
        dtype = torch.float32 if False else None
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device if device is not None else default_device())


# ---------------------------------------------------------------------------------------------------- materials
class Material(object):
    """PhiFlow/phi/physics/material.py:8-108 (only the attributes the extrapolation modes need)."""

    def __init__(self, name, solid=False, friction=0.0, periodic=False):
        self.name, self.solid, self.friction, self.periodic = name, solid, friction, periodic

    def __repr__(self):
        return self.name

    @property
    def _extrapolation_mode(self):          # material.py:70-83
        if self.periodic:
            return "periodic"
        return "boundary" if self.solid else "constant"

    @property
    def _accessible_extrapolation_mode(self):   # material.py:85-92
        if self.periodic:
            return "periodic"
        return "constant" if self.solid else "boundary"

    @staticmethod
    def _map(fn, boundaries):
        if isinstance(boundaries, (tuple, list)):
            return tuple(Material._map(fn, b) for b in boundaries)
        return fn(boundaries)

    @staticmethod
    def extrapolation_mode(boundaries):
        return _collapse(Material._map(lambda m: m._extrapolation_mode, boundaries))

    @staticmethod
    def accessible_extrapolation_mode(boundaries):
        return _collapse(Material._map(lambda m: m._accessible_extrapolation_mode, boundaries))


def _collapse(v):
    """PhiFlow's struct `collapse`: a tuple whose entries are all equal becomes that entry."""
    if isinstance(v, (tuple, list)):
        v = tuple(_collapse(e) for e in v)
        if all(e == v[0] for e in v):
            return v[0]
    return v


OPEN = Material("open", solid=False)
CLOSED = NO_STICK = SLIPPERY = Material("slippery", solid=True, friction=0)
NO_SLIP = STICKY = Material("sticky", solid=True, friction=1)
PERIODIC = Material("periodic", solid=False, periodic=True)


def axis_extrapolation(extrapolation, rank=2):
    """Normalise an extrapolation spec to a per-axis tuple; each entry is a string or a (lower, upper) pair."""
    if isinstance(extrapolation, str) or extrapolation is None:
        return tuple([extrapolation or "boundary"] * rank)
    assert len(extrapolation) == rank, extrapolation
    return tuple(e if isinstance(e, str) else tuple(e) for e in extrapolation)


def is_periodic(ext_axis):
    return ext_axis == "periodic"


# ---------------------------------------------------------------------------------------------------- boxes
class AABox(object):
    """PhiFlow/phi/geom/box.py (lower / upper corners, y first)."""

    def __init__(self, lower, upper):
        self.lower = np.atleast_1d(np.asarray(lower, dtype=np.float64))
        self.upper = np.atleast_1d(np.asarray(upper, dtype=np.float64))

    @property
    def size(self):
        return self.upper - self.lower

    @property
    def half_size(self):
        return self.size * 0.5

    @property
    def rank(self):
        return len(self.size)

    def __eq__(self, other):
        return isinstance(other, AABox) and np.allclose(self.lower, other.lower) and np.allclose(self.upper, other.upper)

    def __repr__(self):
        return "AABox(%s, %s)" % (self.lower, self.upper)

    @staticmethod
    def to_box(value, resolution_hint=None):
        if value is None:
            return AABox(np.zeros(len(resolution_hint)), np.asarray(resolution_hint, np.float64))
        if isinstance(value, AABox):
            return value
        return AABox(np.zeros_like(np.asarray(value, np.float64)), value)


class _BoxType(object):
    """`box[0:1, 0:2]` (PhiFlow/phi/geom/box.py BoxType.__getitem__)."""

    def __getitem__(self, item):
        if not isinstance(item, (tuple, list)):
            item = [item]
        lower = [0 if it.start is None else it.start for it in item]
        upper = [it.stop for it in item]
        return AABox(lower, upper)


box = _BoxType()


# ---------------------------------------------------------------------------------------------------- layout helpers
def unstack_staggered_tensor(tensor):
    """staggered_grid.py:33-39 -> [v [1,Ny+1,Nx,1], u [1,Ny,Nx+1,1]]."""
    return [tensor[:, :, :-1, 0:1], tensor[:, :-1, :, 1:2]]


def stack_staggered_components(tensors):
    """staggered_grid.py:42-46."""
    v, u = tensors
    v = torch.nn.functional.pad(v, (0, 0, 0, 1))          # pad x (dim 2) at the end
    u = torch.nn.functional.pad(u, (0, 0, 0, 0, 0, 1))    # pad y (dim 1) at the end
    return torch.cat([v, u], dim=-1)


# ---------------------------------------------------------------------------------------------------- resampling
def _axis_interpolate(t, dim, coords, mode):
    """Linear interpolation of `t` along `dim` at fractional index positions `coords` (1-D tensor).  Outside the samples:
    'boundary' -> edge value (phi.math.resample with boundary='replicate'), 'periodic' -> wrap, 'constant' -> zeros."""
    n = t.shape[dim]
    lo_f = torch.floor(coords)
    w = (coords - lo_f).to(t.dtype)
    lo = lo_f.long()
    hi = lo + 1
    if mode == "periodic":
        lo_i, hi_i = lo % n, hi % n
        m_lo = m_hi = None
    else:
        lo_i, hi_i = lo.clamp(0, n - 1), hi.clamp(0, n - 1)
        m_lo = m_hi = None
        if mode == "constant":
            m_lo = ((lo >= 0) & (lo < n)).to(t.dtype)
            m_hi = ((hi >= 0) & (hi < n)).to(t.dtype)
    shape = [1] * t.dim()
    shape[dim] = -1
    a, b = t.index_select(dim, lo_i), t.index_select(dim, hi_i)
    if m_lo is not None:
        a, b = a * m_lo.view(shape), b * m_hi.view(shape)
    return a * (1 - w).view(shape) + b * w.view(shape)


def _sample_grid(data, lower, dx, points_yx, extrapolation):
    """Values of a cell-centred array `data` [1,H,W,C] (cell (0,0) starts at `lower`, spacing `dx`) at the tensor-product
    points (ys, xs) -- what phi's Field.at does for axis-aligned grids: linear interpolation in index space, sample
    centres at (i + 1/2) dx."""
    ext = axis_extrapolation(extrapolation, 2)
    out = data
    for axis in (0, 1):
        mode = ext[axis] if isinstance(ext[axis], str) else ext[axis][0]
        coords = (torch.as_tensor(points_yx[axis], dtype=torch.float64, device=data.device) - float(lower[axis])) / float(dx[axis]) - 0.5
        out = _axis_interpolate(out, axis + 1, coords, mode)
    return out


def _centre_points(box, resolution):
    return [box.lower[a] + (np.arange(int(resolution[a])) + 0.5) * (box.size[a] / resolution[a]) for a in (0, 1)]


def _face_points(box, resolution, comp):
    """Sample points of staggered component `comp` (0: v on y-faces, 1: u on x-faces)."""
    pts = _centre_points(box, resolution)
    d = box.size[comp] / resolution[comp]
    pts[comp] = box.lower[comp] + np.arange(int(resolution[comp]) + 1) * d
    return pts


# ---------------------------------------------------------------------------------------------------- fields
class CenteredGrid(object):
    """PhiFlow/phi/physics/field/grid.py:25-194 (data holder + `padded`)."""

    def __init__(self, data, box=None, extrapolation=None, name=None, **kwargs):
        self.data = as_tensor(data)
        if self.data.dtype == torch.float64:
            self.data = self.data.to(torch.float32)
        self.box = AABox.to_box(box, resolution_hint=self.resolution)
        self.extrapolation = extrapolation if extrapolation is not None else "boundary"
        self.name = name

    @property
    def resolution(self):
        return np.array(self.data.shape[1:-1])

    @property
    def rank(self):
        return self.data.dim() - 2

    @property
    def dx(self):
        return self.box.size / self.resolution

    @property
    def component_count(self):
        return self.data.shape[-1]

    def copied_with(self, **kw):
        return CenteredGrid(kw.get("data", self.data), kw.get("box", self.box), kw.get("extrapolation", self.extrapolation))

    def padded(self, widths):
        """grid.py:188-194 with the pad modes of :257-281: 'constant' -> zeros, 'boundary' -> replicate, 'periodic' -> wrap."""
        from .stencils import pad_axis_sides
        if isinstance(widths, int):
            widths = [[widths, widths]] * self.rank
        ext = axis_extrapolation(self.extrapolation, self.rank)
        d = self.data
        for axis, (lo, hi) in enumerate(widths):
            e = ext[axis]
            lo_mode, hi_mode = (e, e) if isinstance(e, str) else e
            d = pad_axis_sides(d, axis + 1, lo, hi, lo_mode, hi_mode)
        w_lo = np.array([w[0] for w in widths])
        w_hi = np.array([w[1] for w in widths])
        return CenteredGrid(d, AABox(self.box.lower - w_lo * self.dx, self.box.upper + w_hi * self.dx), self.extrapolation)

    def _op(self, other, fn):
        o = other.data if isinstance(other, CenteredGrid) else other
        return CenteredGrid(fn(self.data, o), self.box, self.extrapolation)

    def __add__(self, other):
        return self._op(other, torch.add)

    __radd__ = __add__

    def __sub__(self, other):
        return self._op(other, torch.sub)

    def __mul__(self, other):
        return self._op(other, torch.mul)

    __rmul__ = __mul__

    def __truediv__(self, other):
        return self._op(other, torch.div)

    @staticmethod
    def sample(value, domain, batch_size=None, name=None):
        return domain.centered_grid(value)

    def at(self, other):
        """Field.at for axis-aligned grids (phi/physics/field/grid.py:125-140): this field sampled at the points of `other`
        -- a CenteredGrid (cell centres) or a StaggeredGrid (face centres; a scalar field goes to both components, a
        2-channel field channel-wise).  Used by the scripts to bring high-resolution frames and cell-centred viscosity to the
        simulation grid."""
        if isinstance(other, StaggeredGrid):
            comps = []
            for c in (0, 1):
                d = self.data if self.data.shape[-1] == 1 else self.data[..., c:c + 1]
                comps.append(_sample_grid(d, self.box.lower, self.dx, _face_points(other.box, other.resolution, c), self.extrapolation))
            return StaggeredGrid(stack_staggered_components(comps), other.box, extrapolation=other.extrapolation)
        pts = _centre_points(other.box, other.resolution)
        return CenteredGrid(_sample_grid(self.data, self.box.lower, self.dx, pts, self.extrapolation), other.box, other.extrapolation)


class StaggeredGrid(object):
    """PhiFlow/phi/physics/field/staggered_grid.py:56-228: `.data` is the tuple (v-component, u-component) of CenteredGrids."""

    def __init__(self, data, box=None, name=None, extrapolation=None, **kwargs):
        if isinstance(data, StaggeredGrid):
            data = data.staggered_tensor()
        if isinstance(data, (list, tuple)):
            comps = [c.data if isinstance(c, CenteredGrid) else as_tensor(c) for c in data]
        else:
            t = as_tensor(data)
            if t.dtype == torch.float64:
                t = t.to(torch.float32)
            comps = unstack_staggered_tensor(t)
        self.extrapolation = extrapolation if extrapolation is not None else "boundary"
        ny = comps[1].shape[1]
        nx = comps[0].shape[2]
        self._resolution = np.array([ny, nx])
        self.box = AABox.to_box(box, resolution_hint=self._resolution)
        self.data = tuple(CenteredGrid(c, None, self.extrapolation) for c in comps)
        self.name = name

    @staticmethod
    def sample(value, domain, batch_size=None, name=None):
        return domain.staggered_grid(value)

    @property
    def resolution(self):
        return self._resolution

    @property
    def rank(self):
        return 2

    @property
    def dx(self):
        return self.box.size / self.resolution

    def staggered_tensor(self):
        return stack_staggered_components([c.data for c in self.data])

    def copied_with(self, **kw):
        return StaggeredGrid(kw.get("data", self.staggered_tensor()), kw.get("box", self.box),
                             extrapolation=kw.get("extrapolation", self.extrapolation))

    def unstack(self):
        return self.data

    def padded(self, widths):
        """staggered_grid.py:222-228: every component padded by `widths` cells with its extrapolation mode, the box grown to match."""
        comps = [c.padded(widths).data for c in self.data]
        if isinstance(widths, int):
            widths = [[widths, widths]] * self.rank
        w_lo, w_hi = np.array([w[0] for w in widths]), np.array([w[1] for w in widths])
        box = AABox(self.box.lower - w_lo * self.dx, self.box.upper + w_hi * self.dx)
        return StaggeredGrid(comps, box, extrapolation=self.extrapolation)

    def at(self, other):
        """staggered_grid.py `at`: every component resampled to the matching face points of `other` (a StaggeredGrid over
        any resolution / box).  A component is a cell-centred array over the box grown by half a cell along its own axis."""
        assert isinstance(other, StaggeredGrid)
        comps = []
        for c in (0, 1):
            lower = np.array(self.box.lower, dtype=np.float64).copy()
            lower[c] -= 0.5 * self.dx[c]
            comps.append(_sample_grid(self.data[c].data, lower, self.dx, _face_points(other.box, other.resolution, c), self.extrapolation))
        return StaggeredGrid(stack_staggered_components(comps), other.box, extrapolation=other.extrapolation)

    def at_centers(self):
        """Linear interpolation of both components to the cell centres -> CenteredGrid [1,Ny,Nx,2] (y component first)."""
        v, u = self.data[0].data, self.data[1].data
        vc = 0.5 * (v[:, 1:] + v[:, :-1])
        uc = 0.5 * (u[:, :, 1:] + u[:, :, :-1])
        return CenteredGrid(torch.cat([vc, uc], dim=-1), self.box, self.extrapolation)


class Domain(object):
    """PhiFlow/phi/physics/domain.py:13-209."""

    def __init__(self, resolution, boundaries=OPEN, box=None, **kwargs):
        self.resolution = np.array(resolution).reshape(-1)
        self.boundaries = _collapse(boundaries) if isinstance(boundaries, (tuple, list)) else boundaries
        self.box = AABox.to_box(box, resolution_hint=self.resolution)

    @property
    def dx(self):
        return self.box.size / self.resolution

    @property
    def rank(self):
        return len(self.resolution)

    def staggered_grid(self, data, dtype=None, name=None, batch_size=None, extrapolation=None):
        if extrapolation is None:
            extrapolation = Material.extrapolation_mode(self.boundaries)     # domain.py:173-174
        if isinstance(data, (int, float)):
            ny, nx = self.resolution
            data = torch.zeros((1, ny + 1, nx + 1, 2), dtype=torch.float32, device=default_device()) + data
            data[:, :, nx, 0] = 0
            data[:, ny, :, 1] = 0
        if isinstance(data, StaggeredGrid):
            return data
        return StaggeredGrid(data, self.box, name, extrapolation=extrapolation)

    def centered_grid(self, data, components=1, dtype=None, name=None, batch_size=None, extrapolation=None):
        if isinstance(data, (int, float)):
            ny, nx = self.resolution
            data = torch.zeros((1, ny, nx, components), dtype=torch.float32, device=default_device()) + data
        if isinstance(data, CenteredGrid):
            return data
        return CenteredGrid(data, self.box, extrapolation if extrapolation is not None else "boundary", name)


def placeholder(shape, dtype=None, basename=None):
    """Script compatibility (PhiFlow/phi/tf/util.py placeholder): eager mode has no placeholders -- returns zeros."""
    return torch.zeros(tuple(int(s) for s in shape), dtype=torch.float32, device=default_device())
