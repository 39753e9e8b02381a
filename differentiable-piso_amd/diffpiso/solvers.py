"""Solver classes of the PISO step, backed by libpiso_hip.so (hand-written gfx950 kernels behind a C ABI).

Same class surface as the reference (diffpiso/linear_solver.py, diffpiso/piso_cuda_pressure_solver.py):
  LinearSolver                              base type                               (linear_solver.py:15-30)
  LinearSolverScipy                         host-side direct solve, cross-check only   (linear_solver.py:33-57)
  LinearSolverCudaMultiBicgstabILU          u+v ILU(0)-BiCGStab, adjoint = A^T solve  (linear_solver.py:113-178)
  PisoPressureSolverCudaCustom              A0-weighted 5-diagonal CG, adjoint = same solve (piso_cuda_pressure_solver.py:36-114)
The adjoint linear solves are `torch.autograd.Function` nodes (the reference uses tf.custom_gradient).
There is NO CPU / PyTorch fallback: tensors must live on the GPU and the native library must be present.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N
from .grids import StaggeredGrid, as_tensor
from .stencils import flatten_staggered_data


class LinearSolver(object):
    """diffpiso/linear_solver.py:15-30."""

    def __init__(self, name, supported_devices, supports_guess, supports_batch, solver_type, input_format):
        self.name = name
        self.supported_devices = supported_devices
        self.supports_guess = supports_guess
        self.supports_batch = supports_batch
        self.solver_type = solver_type
        self.input_format = input_format

    def solve(self, *args):
        raise NotImplementedError(self.__class__)

    def __repr__(self):
        return self.name


def _scalar(v):
    if isinstance(v, torch.Tensor):
        return float(v.item())
    return float(v)


class _ScipySolveFn(torch.autograd.Function):
    """solve_call of LinearSolverScipy (diffpiso/linear_solver.py:45-54): direct sparse solve on the host, gradient w.r.t. the
    right-hand side = solve with the transposed matrix."""

    @staticmethod
    def _solve(mv, ci, rp, rhs, transpose):
        import scipy.sparse
        import scipy.sparse.linalg
        m = scipy.sparse.csr_matrix((mv.detach().cpu().numpy(), ci.detach().cpu().numpy(), rp.detach().cpu().numpy()))
        if transpose:
            m = m.transpose()
        x = scipy.sparse.linalg.spsolve(m.tocsr(), rhs.detach().cpu().numpy().reshape(-1))
        return torch.as_tensor(np.asarray(x, dtype=np.float32), device=rhs.device)

    @staticmethod
    def forward(ctx, rhs, mv, ci, rp, transpose):
        ctx.save_for_backward(mv, ci, rp)
        ctx.transpose = transpose
        return _ScipySolveFn._solve(mv, ci, rp, rhs, transpose)

    @staticmethod
    def backward(ctx, ds):
        mv, ci, rp = ctx.saved_tensors
        return _ScipySolveFn._solve(mv, ci, rp, ds, not ctx.transpose), None, None, None, None


class LinearSolverScipy(LinearSolver):
    """diffpiso/linear_solver.py:33-57: scipy.sparse.linalg.spsolve on ONE CSR matrix, on the host (the reference's manual
    cross-check for the CUDA solvers; it wraps it in tf.py_function).  It is a CPU solver by design, exactly as in the
    reference -- `piso_step` never selects it by itself and the HIP solvers never fall back to it."""

    def __init__(self):
        LinearSolver.__init__(self, "Scipy Solver for sparse matrix", supported_devices="CPU", supports_guess=False,
                              supports_batch=False, solver_type="-", input_format="csr")

    def solve(self, matrix_values, row_ptr, col_indices, rhs, transpose=False):
        return _ScipySolveFn.apply(rhs.reshape(-1), matrix_values.reshape(-1), col_indices.reshape(-1), row_ptr.reshape(-1),
                                   bool(transpose))


def multi_bicgstab_ilu_native(values, row_ptr, col_indices, rhs, x0, nx, ny, tol, max_it, transpose, band_rows, warn, slab_comm=None, negate=False):
    """One call of piso_multi_bicgstab_ilu_{f32,f64}. Returns (x, iterations[2]); sets warn[0] in place on NaN input.
    slab_comm (distributed.SlabCommunicator, either transport, more than one rank): the solve is cut into y-slabs over the ranks
    (every rank passes the full arrays and works on its rows; dot products are all-reduced inside the scalar kernels, the edge
    rows of the SpMV inputs travel through the mailboxes) and every rank returns the full solution.
    negate: the system matrix is -values (the sign is applied where the kernel reads the values: no pass over the array for it)."""
    if slab_comm is not None and slab_comm.world > 1:
        from .distributed import multi_bicgstab_ilu_slab, multi_bicgstab_ilu_slab_local
        if slab_comm.sharded:                # slab-decomposed STEP: every array holds the rank's stored rows (sharding.py)
            return multi_bicgstab_ilu_slab_local(slab_comm, values, row_ptr, col_indices, rhs, x0, nx, ny, tol, max_it, transpose, band_rows, warn,
                                                 negate=negate)
        return multi_bicgstab_ilu_slab(slab_comm, values, row_ptr, col_indices, rhs, x0, nx, ny, tol, max_it, transpose, band_rows, warn, negate=negate)
    dt = values.dtype
    assert dt in (torch.float32, torch.float64)
    values, rhs, x0 = values.contiguous(), rhs.to(dt).contiguous(), x0.to(dt).contiguous()
    row_ptr, col_indices = row_ptr.contiguous(), col_indices.contiguous()
    x = torch.empty_like(rhs)
    elem = 8 if dt == torch.float64 else 4
    nbytes = N.lib.piso_bicgstab_workspace_bytes(nx, ny, elem)
    ws = N.workspace(nbytes, rhs.device, "bicgstab")
    its = (C.c_int * 2)()
    fn = N.lib.piso_multi_bicgstab_ilu_f64 if dt == torch.float64 else N.lib.piso_multi_bicgstab_ilu_f32
    st = fn(N.ptr(values), N.ptr(row_ptr), N.ptr(col_indices), N.ptr(rhs), N.ptr(x0), N.ptr(x), nx, ny, C.c_float(tol),
            int(max_it), (1 if transpose else 0) | (2 if negate else 0), int(band_rows), N.ptr(warn), its, N.ptr(ws), C.c_size_t(ws.numel()),
            N.stream_ptr())
    N.check(st, "piso_multi_bicgstab_ilu")
    return x, (its[0], its[1])


class _LinearSolveFn(torch.autograd.Function):
    """solve_call of diffpiso/linear_solver.py:163-175: gradient only w.r.t. the right-hand side, obtained by solving with
    the transposed matrix (same initial guess), multiplied by (1 - warn)."""

    @staticmethod
    def forward(ctx, rhs, values, row_ptr, col_indices, x0, solver, nx, ny, transpose, warn, negate=False):
        tol = _scalar(solver.accuracy)
        x, its = multi_bicgstab_ilu_native(values, row_ptr, col_indices, rhs, x0, nx, ny, tol, solver.max_iterations,
                                           transpose, solver.band_rows, warn, solver.slab_comm, negate=negate)
        solver.last_iterations = its
        solver.stats["solves"] += 1
        solver.stats["iterations"] += max(its)
        # the reference's backward op receives the SAME warn buffer the forward op aliased and mutated
        # (linear_solver.py:165-173, multi_bicgstab_ilu_linear_solve_op.cc:136-140): a forward warning zeroes the gradient
        ctx.save_for_backward(values, row_ptr, col_indices, x0, warn.clone())
        ctx.meta = (solver, nx, ny, transpose, negate)
        return x.to(torch.float32), warn.to(torch.float32)

    @staticmethod
    def backward(ctx, ds, dw):
        values, row_ptr, col_indices, x0, warn_fwd = ctx.saved_tensors
        solver, nx, ny, transpose, negate = ctx.meta
        warn_b = warn_fwd.clone()
        tol = _scalar(solver.accuracy)
        df, its = multi_bicgstab_ilu_native(values, row_ptr, col_indices, ds.to(values.dtype), x0, nx, ny, tol,
                                            solver.max_iterations, not transpose, solver.band_rows, warn_b, solver.slab_comm, negate=negate)
        solver.last_adjoint_iterations = its
        solver.stats["adjoint_solves"] += 1
        solver.stats["adjoint_iterations"] += max(its)
        df = df.to(torch.float32) * (1.0 - warn_b.to(torch.float32)[0])
        return df, None, None, None, None, None, None, None, None, None, None


class LinearSolverCudaMultiBicgstabILU(LinearSolver):
    """diffpiso/linear_solver.py:113-178.  `accuracy` may be a float or a 0-d/1-element tensor (the reference accepts a
    placeholder, lid_driven_cavity_2d.py:12-13) and may be re-assigned between steps.
    band_rows: preconditioner block height of the MI355X engine (0 = automatic, < 0 = one block); not in the reference."""

    def __init__(self, accuracy=1e-5, max_iterations=2000, cast_to_double=False, band_rows=0):
        LinearSolver.__init__(self, "HIP dual iLU-preconditioned BiCGStab solve", supported_devices=("GPU",),
                              supports_guess=True, supports_batch=False, solver_type="iterative", input_format="csr")
        self.max_iterations = max_iterations
        self.cast_to_double = cast_to_double
        self.accuracy = accuracy
        self.band_rows = band_rows
        self.slab_comm = None        # distributed.SlabCommunicator: cut every solve into y-slabs over the ranks
        self.last_iterations = None
        self.last_adjoint_iterations = None
        self.stats = dict(solves=0, iterations=0, adjoint_solves=0, adjoint_iterations=0)   # cumulative (max over u, v per solve)

    def solve(self, matrix_values, row_ptr, col_indices, rhs, staggered_shape, initial_guess=None, offset=0,
              transpose=False, unrolling_step=0, warn=None, negate=False):
        """negate (not in the reference): the system matrix is -matrix_values.  The reference's step passes `-matrix_values`
        (piso_tf.py:41) - a pass over the whole value array per step; `piso_step` here passes the values and this flag instead."""
        dt = torch.float64 if self.cast_to_double else torch.float32
        values = matrix_values.reshape(-1).to(dt)
        flat_rhs = rhs.reshape(-1)
        ny, nx = int(staggered_shape[1]) - 1, int(staggered_shape[2]) - 1
        n_tot = (nx + 1) * ny + nx * (ny + 1)
        if initial_guess is None:
            flat_x = torch.zeros(n_tot, dtype=dt, device=flat_rhs.device)
        else:
            flat_x = initial_guess.reshape(-1).detach().to(dt)
        if warn is None:
            warn = torch.zeros(1, dtype=torch.uint8, device=flat_rhs.device)
        elif warn.dtype != torch.uint8:
            warn = (warn != 0).to(torch.uint8)
        sol, w = _LinearSolveFn.apply(flat_rhs, values.detach(), row_ptr.reshape(-1), col_indices.reshape(-1), flat_x, self,
                                      nx, ny, bool(transpose), warn, bool(negate))
        return [sol, w]


LinearSolverHipMultiBicgstabILU = LinearSolverCudaMultiBicgstabILU


class _SingleSolveFn(torch.autograd.Function):
    """solve_call of diffpiso/linear_solver.py:95-107: gradient w.r.t. the right-hand side by the transposed solve."""

    @staticmethod
    def forward(ctx, rhs_pair, values, row_ptr, col_indices, x0, solver, nx, ny, transpose):
        warn = torch.zeros(1, dtype=torch.uint8, device=rhs_pair.device)
        x, its = multi_bicgstab_ilu_native(values, row_ptr, col_indices, rhs_pair, x0, nx, ny, _scalar(solver.accuracy),
                                           solver.max_iterations, transpose, 0, warn)
        solver.last_iterations = its
        solver.last_warn = warn                              # (device byte: NaN input; `solve` looks at it)
        ctx.save_for_backward(values, row_ptr, col_indices, x0)
        ctx.meta = (solver, nx, ny, transpose)
        return x

    @staticmethod
    def backward(ctx, ds):
        values, row_ptr, col_indices, x0 = ctx.saved_tensors
        solver, nx, ny, transpose = ctx.meta
        warn = torch.zeros(1, dtype=torch.uint8, device=ds.device)
        df, _ = multi_bicgstab_ilu_native(values, row_ptr, col_indices, ds.contiguous(), x0, nx, ny, _scalar(solver.accuracy),
                                          solver.max_iterations, not transpose, 0, warn)
        return df, None, None, None, None, None, None, None, None


class LinearSolverCudaBicgstabILU(LinearSolver):
    """diffpiso/linear_solver.py:60-110: the single-matrix ILU(0)-BiCGStab (superseded in the reference by the pair solver; none
    of its scripts constructs it).  The HIP engine solves the two advection matrices of a staggered grid together and takes its
    structure from the grid, so the ONE matrix must be the u or the v matrix of a grid that is named: `staggered_shape`
    ([1, ny+1, nx+1, 2]), `component` ('u' | 'v') and `bool_periodic` ((y, x)) are additional keyword arguments of `solve`.  The
    other component is filled in with a diffusion matrix of the same grid and a zero right-hand side (it converges at once)."""

    _patterns = {}       # (nx, ny, per_x, per_y, device) -> (stand-in pair values, row pointers, columns, nnz): geometry only, built once

    def __init__(self, accuracy=1e-5, max_iterations=2000, raise_on_nan=True):
        LinearSolver.__init__(self, "HIP iLU-preconditioned BiCGStab solve", supported_devices=("GPU",), supports_guess=True,
                              supports_batch=False, solver_type="iterative", input_format="csr")
        self.accuracy = accuracy
        self.max_iterations = max_iterations
        self.last_iterations = None
        self.last_warn = None           # the pair solver's warning byte of the last solve (1: NaN in the matrix / right-hand side / guess)
        self.raise_on_nan = raise_on_nan

    @classmethod
    def _pattern(cls, nx, ny, per_x, per_y, dev):
        """The pair's pattern and a well-posed stand-in for the other component (diffusion + identity on a resting fluid), cached per
        grid: it depends on (nx, ny, periodicity) only, and checking a caller's pattern against it costs two host synchronisations."""
        from .piso import assemble_from_padded
        key = (nx, ny, per_x, per_y, str(dev))
        hit = cls._patterns.get(key)
        if hit is None:
            n_u, n_v = (nx + 1) * ny, nx * (ny + 1)
            pad = torch.zeros((ny + 2) * (nx + 3) + (ny + 3) * (nx + 2), dtype=torch.float32, device=dev)
            val, rp, col, _, nnz = assemble_from_padded(pad, nx, ny, (1.0, 1.0), per_x, per_y, torch.zeros(n_u + n_v, dtype=torch.uint8, device=dev),
                                                        torch.ones((ny + 2) * (nx + 2), dtype=torch.float32, device=dev), 1.0, None, 1.0)
            if len(cls._patterns) > 8:
                cls._patterns.clear()
            hit = cls._patterns[key] = (-val, rp, col, (int(nnz[0]), int(nnz[1])), set())      # (piso_step hands the solver -matrix_values, piso_tf.py:41)
        return hit

    def solve(self, matrix_values, row_ptr, col_indices, rhs, initial_guess=None, offset=0, transpose=False,
              staggered_shape=None, component=None, bool_periodic=(False, False)):
        if staggered_shape is None or component not in ("u", "v"):
            raise NotImplementedError("LinearSolverCudaBicgstabILU.solve: the MI355X engine takes the matrix structure from the staggered grid - "
                                      "pass staggered_shape=[1, ny+1, nx+1, 2], component='u'|'v' (and bool_periodic), or use "
                                      "LinearSolverCudaMultiBicgstabILU, which is what the reference's scripts use")
        ny, nx = int(staggered_shape[1]) - 1, int(staggered_shape[2]) - 1
        n_u, n_v = (nx + 1) * ny, nx * (ny + 1)
        dev = rhs.device
        per_y, per_x = bool(bool_periodic[0]), bool(bool_periodic[1])
        val, rp, col, (nnz_u, nnz_v), checked = self._pattern(nx, ny, per_x, per_y, dev)
        lo, hi, r0, rows = (0, nnz_u, 0, n_u) if component == "u" else (nnz_u, nnz_u + nnz_v, n_u + 1, n_v)
        mv = matrix_values.reshape(-1).to(torch.float32)
        if mv.numel() != hi - lo or int(row_ptr.numel()) != rows + 1 or int(col_indices.numel()) != hi - lo:
            raise ValueError("LinearSolverCudaBicgstabILU: the matrix is not the %s matrix of a %d x %d staggered grid (%d values, expected %d)"
                             % (component, nx, ny, mv.numel(), hi - lo))
        # (a caller's pattern arrays are compared with the grid's ONCE per pair of tensors: the comparison synchronises the host)
        pat_key = (component, row_ptr.data_ptr(), col_indices.data_ptr(), int(row_ptr._version), int(col_indices._version))
        if pat_key not in checked:
            if not (torch.equal(row_ptr.reshape(-1).to(torch.int32), rp[r0:r0 + rows + 1]) and torch.equal(col_indices.reshape(-1).to(torch.int32), col[lo:hi])):
                raise ValueError("LinearSolverCudaBicgstabILU: row pointers / column indices are not the 5-point pattern of the grid")
            if len(checked) > 16:
                checked.clear()
            checked.add(pat_key)
        val = val.clone()
        val[lo:hi] = mv
        flat_rhs = rhs.reshape(-1).to(torch.float32)
        off = 0 if component == "u" else n_u
        x0 = torch.zeros(n_u + n_v, dtype=torch.float32, device=dev)
        if initial_guess is not None:
            x0[off:off + rows] = initial_guess.reshape(-1).detach().to(torch.float32)
        embed = torch.zeros(n_u + n_v, dtype=torch.float32, device=dev)
        pair_rhs = embed.index_put((torch.arange(off, off + rows, device=dev),), flat_rhs)          # (differentiable w.r.t. rhs)
        sol = _SingleSolveFn.apply(pair_rhs, val, rp, col, x0, self, nx, ny, bool(transpose))
        if self.raise_on_nan and int(self.last_warn.item()) != 0:
            # (the pair solver's failure handling is the reference's - NaN input: warning byte, zero solution; this class has no `warn`
            # output to hand it to, so it says so instead of returning zeros silently)
            raise N.PisoNativeError("LinearSolverCudaBicgstabILU: NaN in the matrix, the right-hand side or the initial guess (the solver's warning byte is set)")
        return sol[off:off + rows]


def mat_vec_mul_csr(matrix_values, row_pointers, column_indices, staggered_field, staggered_shape):
    """diffpiso/linear_solver.py:181-196: [M_u u, M_v v] of the two advection matrices (CSR pair as `advection_matrix_cuda` returns
    it) as a staggered tensor - the HIP gather / segment-sum product (`piso_csr_matvec_f32`)."""
    from .grids import StaggeredGrid
    from .piso import _CsrMatVec
    from .stencils import flatten_staggered_data, stagger_flattened_data
    ny, nx = int(staggered_shape[1]) - 1, int(staggered_shape[2]) - 1
    field = staggered_field if isinstance(staggered_field, StaggeredGrid) else StaggeredGrid(staggered_field)
    flat = flatten_staggered_data(field, coord_flip=True)
    prod = _CsrMatVec.apply(flat, matrix_values.reshape(-1).to(torch.float32), row_pointers.reshape(-1), column_indices.reshape(-1), nx, ny)
    return stagger_flattened_data(prod, tuple(int(v) for v in staggered_shape), coord_flip=True)


def print_residual(matrix_values, row_pointers, column_indices, staggered_field, staggered_shape, rhs):
    """diffpiso/linear_solver.py:198-206: prints sum |M x - rhs| and returns the residual (flat, in the reference's default
    flatten order)."""
    from .grids import StaggeredGrid
    from .stencils import flatten_staggered_data
    prod = mat_vec_mul_csr(matrix_values, row_pointers, column_indices, staggered_field, staggered_shape)
    residual = flatten_staggered_data(StaggeredGrid(prod)) - rhs.reshape(-1)
    print("linsolve residual", float(residual.abs().sum()))
    return residual


# ------------------------------------------------------------------------------------------------------------------
class PoissonSolver(object):
    """PhiFlow/phi/physics/pressuresolver/solver_api.py:10-38 (attribute holder)."""

    def __init__(self, name, supported_devices, supports_guess, supports_loop_counter, supports_continuous_masks):
        self.name = name
        self.supported_devices = supported_devices
        self.supports_guess = supports_guess
        self.supports_loop_counter = supports_loop_counter
        self.supports_continuous_masks = supports_continuous_masks

    def __repr__(self):
        return self.name


def laplace_matrix_native(nx, ny, active, accessible, a0_vfirst, dtype, sharding=None):
    """[ny nx][5] pressure matrix; sharding (the slab-decomposed step): masks and a0 hold the rank's stored rows, the matrix its
    OWNED rows [nyl nx][5] - what the slab CG takes."""
    if sharding is None:
        L = torch.empty(nx * ny * 5, dtype=dtype, device=a0_vfirst.device)
        fn = N.lib.piso_laplace_matrix_f64 if dtype == torch.float64 else N.lib.piso_laplace_matrix_f32
        N.check(fn(nx, ny, N.ptr(active), N.ptr(accessible), N.ptr(a0_vfirst.contiguous()), N.ptr(L), N.stream_ptr()),
                "piso_laplace_matrix")
        return L
    L = torch.empty(nx * sharding.nyl * 5, dtype=dtype, device=a0_vfirst.device)
    fn = N.lib.piso_laplace_matrix_f64_slab if dtype == torch.float64 else N.lib.piso_laplace_matrix_f32_slab
    N.check(fn(nx, ny, N.ptr(active), N.ptr(accessible), N.ptr(a0_vfirst.contiguous()), N.ptr(L), N.stream_ptr(), sharding.slab_ptr),
            "piso_laplace_matrix_slab")
    return L


class DeferredInt(object):
    """An iteration count that still lives on the device (asynchronous solves, piso_cg_solve_async_*): behaves like an int and
    reads its value - ONE synchronising copy - the first time somebody looks at it.  Sums of such counts (solver.stats) stay
    deferred, so a training loop that never looks never waits."""
    __slots__ = ("_base", "_pending")

    def __init__(self, base=0, pending=()):
        self._base, self._pending = int(base), list(pending)

    def device_tensor(self):
        """The count as an int32 device tensor [1] (the reference's `iterations` output) without a host round trip."""
        if len(self._pending) == 1 and self._base == 0:
            return self._pending[0]
        dev = self._pending[0].device if self._pending else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None)
        return torch.tensor([int(self)], dtype=torch.int32, device=dev)

    def __int__(self):
        if self._pending:
            self._base += int(torch.stack([t.reshape(()) for t in self._pending]).sum().item())
            self._pending = []
        return self._base

    __index__ = __int__

    def __add__(self, other):
        if isinstance(other, DeferredInt):
            out = DeferredInt(self._base + other._base, self._pending + other._pending)
        else:
            out = DeferredInt(self._base + int(other), self._pending)
        if len(out._pending) > 64:          # (a long run that never looks: fold the pending counts on the device, still no wait)
            out._pending = [torch.stack([t.reshape(()) for t in out._pending]).sum().to(torch.int64)]
        return out

    __radd__ = __add__

    def __float__(self): return float(int(self))
    def __bool__(self): return int(self) != 0
    def __hash__(self): return hash(int(self))
    def __repr__(self): return repr(int(self))
    def __str__(self): return str(int(self))
    def __format__(self, spec): return format(int(self), spec)
    def __eq__(self, o): return int(self) == o
    def __ne__(self, o): return int(self) != o
    def __lt__(self, o): return int(self) < o
    def __le__(self, o): return int(self) <= o
    def __gt__(self, o): return int(self) > o
    def __ge__(self, o): return int(self) >= o
    def __sub__(self, o): return int(self) - o
    def __rsub__(self, o): return o - int(self)
    def __mul__(self, o): return int(self) * o
    __rmul__ = __mul__
    def __mod__(self, o): return int(self) % o
    def __truediv__(self, o): return int(self) / o
    def __rtruediv__(self, o): return o / int(self)
    def __floordiv__(self, o): return int(self) // o
    def __neg__(self): return -int(self)
    def __abs__(self): return abs(int(self))


class _Stats(dict):
    """Cumulative solver counters.  Counts of asynchronous solves are added as DeferredInt (no host wait); whatever LEAVES the
    solver through the mapping interface - solver.stats["iterations"], dict(solver.stats), .items() - is a plain int (so that
    json.dumps, numpy and torch see numbers), at the price of the one synchronising read that looking costs."""

    def add(self, key, value):
        dict.__setitem__(self, key, dict.__getitem__(self, key) + value)

    @staticmethod
    def _plain(v):
        return int(v) if isinstance(v, DeferredInt) else v

    def __getitem__(self, key):
        return self._plain(dict.__getitem__(self, key))

    def get(self, key, default=None):
        return self._plain(dict.get(self, key, default))

    def items(self):
        return [(k, self._plain(v)) for k, v in dict.items(self)]

    def values(self):
        return [self._plain(v) for v in dict.values(self)]

    def copy(self):
        return dict(self.items())

    def keys(self):
        return dict.keys(self)

    def __iter__(self):
        return dict.__iter__(self)


class _PlainCount(object):
    """Descriptor of `last_iterations` / `last_adjoint_iterations`: stored as handed in (possibly a DeferredInt), read as int."""

    def __init__(self, name):
        self.name = "_" + name

    def __get__(self, obj, cls=None):
        if obj is None:
            return self
        v = getattr(obj, self.name, None)
        if isinstance(v, DeferredInt):
            v = int(v)
            setattr(obj, self.name, v)
        return v

    def __set__(self, obj, value):
        setattr(obj, self.name, value)


_ASYNC_MAX_CELLS = 4608          # csrc/cg_tiny.h: kTinyMaxCells (the library answers PISO_ERR_NEEDS_HOST for anything it cannot run in one launch)


def cg_solve_native(nx, ny, per_x, per_y, L, div, accuracy, max_iterations, rank_deficient, residual_reset):
    """-> (x, iterations).  Grids the library solves in ONE launch (tiny grids: the lid-driven cavity) are queued without waiting
    for the result; `iterations` is then a DeferredInt."""
    dt = L.dtype
    div = div.reshape(-1).to(dt).contiguous()
    x = torch.empty_like(div)
    elem = 8 if dt == torch.float64 else 4
    ws = N.workspace(N.lib.piso_cg_workspace_bytes(nx, ny, elem), div.device, "cg")
    if nx * ny <= _ASYNC_MAX_CELLS:
        it_dev = torch.empty(1, dtype=torch.int32, device=div.device)
        fn = N.lib.piso_cg_solve_async_f64 if dt == torch.float64 else N.lib.piso_cg_solve_async_f32
        st = fn(nx, ny, int(per_x), int(per_y), N.ptr(L), N.ptr(div), N.ptr(x), C.c_float(accuracy), int(max_iterations),
                int(bool(rank_deficient)), int(residual_reset), N.ptr(it_dev), N.ptr(ws), C.c_size_t(ws.numel()), N.stream_ptr())
        if st == 0:
            return x, DeferredInt(0, [it_dev])
        if st != N.ERR_NEEDS_HOST:
            N.check(st, "piso_cg_solve_async")
    it = C.c_int(0)
    fn = N.lib.piso_cg_solve_f64 if dt == torch.float64 else N.lib.piso_cg_solve_f32
    st = fn(nx, ny, int(per_x), int(per_y), N.ptr(L), N.ptr(div), N.ptr(x), C.c_float(accuracy), int(max_iterations),
            int(bool(rank_deficient)), int(residual_reset), C.byref(it), N.ptr(ws), C.c_size_t(ws.numel()), N.stream_ptr())
    N.check(st, "piso_cg_solve")
    return x, it.value


class _PressureSolveFn(torch.autograd.Function):
    """psolve of diffpiso/piso_cuda_pressure_solver.py:90-109: gradient only w.r.t. the divergence = the same CG solve
    applied to the incoming gradient (the operator is symmetric)."""

    @staticmethod
    def forward(ctx, divergence, L, solver, nx, ny, per_x, per_y, rank_deficient):
        x, it = solver._cg(nx, ny, per_x, per_y, L, divergence, _scalar(solver.accuracy), solver.max_iterations,
                           rank_deficient, solver.residual_reset)
        solver.last_iterations = it
        solver.stats.add("solves", 1)
        solver.stats.add("iterations", it)
        ctx.save_for_backward(L)
        ctx.meta = (solver, nx, ny, per_x, per_y, rank_deficient, divergence.shape)
        iterations = it.device_tensor() if isinstance(it, DeferredInt) else torch.tensor([it], dtype=torch.int32, device=divergence.device)
        return x.reshape(divergence.shape).to(torch.float32), iterations

    @staticmethod
    def backward(ctx, dp, di):
        (L,) = ctx.saved_tensors
        solver, nx, ny, per_x, per_y, rank_deficient, shape = ctx.meta
        g, it = solver._cg(nx, ny, per_x, per_y, L, dp.reshape(-1), _scalar(solver.accuracy), solver.max_iterations,
                           rank_deficient, solver.residual_reset)
        solver.last_adjoint_iterations = it
        solver.stats.add("adjoint_solves", 1)
        solver.stats.add("adjoint_iterations", it)
        return g.reshape(shape).to(torch.float32), None, None, None, None, None, None, None


class PisoPressureSolverCudaCustom(PoissonSolver):
    """diffpiso/piso_cuda_pressure_solver.py:36-114."""
    last_iterations = _PlainCount("last_iterations")                   # (int on read; a tiny grid's asynchronous solve hands in a DeferredInt)
    last_adjoint_iterations = _PlainCount("last_adjoint_iterations")

    def __init__(self, dx, accuracy=1e-5, max_iterations=2000, residual_reset=10, randomized_restarts=0,
                 cast_to_double=True):
        PoissonSolver.__init__(self, "HIP Conjugate Gradient", supported_devices=("GPU",), supports_loop_counter=False,
                               supports_guess=True, supports_continuous_masks=False)
        self.accuracy = accuracy
        self.max_iterations = max_iterations
        self.dx = dx
        self.scaling_field = None
        self.solve_count = 0.001
        self.laplace_rank_deficient = None
        self.residual_reset = residual_reset
        assert randomized_restarts >= 0
        if randomized_restarts != 0:
            raise NotImplementedError("randomized_restarts > 0 is never used by the reference's scripts "
                                      "(combined_training_integrated.py:487) and is not implemented")
        self.randomized_restarts = randomized_restarts
        self.cast_to_double = cast_to_double
        self.last_iterations = None
        self.last_adjoint_iterations = None
        self.slab_comm = None        # distributed.SlabCommunicator: decompose the CG into y-slabs over the ranks (fp64 only)
        self.stats = _Stats(solves=0, iterations=0, adjoint_solves=0, adjoint_iterations=0)   # cumulative; callers may reset

    def _cg(self, nx, ny, per_x, per_y, L, div, accuracy, max_iterations, rank_deficient, residual_reset):
        if self.slab_comm is not None and self.slab_comm.sharded and L.dtype != torch.float64:
            # (L and div of a sharded step are valid on this rank's rows only: the one-GPU kernel on the whole grid would read garbage)
            raise N.PisoNativeError("the slab-decomposed pressure CG is fp64 only: a sharded step needs cast_to_double=True")
        if self.slab_comm is not None and self.slab_comm.world > 1 and L.dtype == torch.float64:
            from .distributed import cg_solve_slab, cg_solve_slab_local
            if self.slab_comm.sharded:       # slab-decomposed STEP: L holds the rank's owned rows, div its stored rows; the result stays there
                sh = self.slab_comm.step_sharding
                d_loc = sh.owned_cells(div.reshape(-1).to(torch.float64)).reshape(-1).contiguous()
                x_loc, it = cg_solve_slab_local(self.slab_comm, nx, sh.nyl, per_x, per_y, L, d_loc, accuracy, max_iterations, rank_deficient,
                                                residual_reset, gather=False)
                x = torch.zeros(sh.n_cells, dtype=x_loc.dtype, device=x_loc.device)
                sh.owned_cells(x).copy_(x_loc.view(sh.nyl, nx))
                return x, it
            return cg_solve_slab(self.slab_comm, nx, ny, per_x, per_y, L, div, accuracy, max_iterations, rank_deficient,
                                 residual_reset)
        return cg_solve_native(nx, ny, per_x, per_y, L, div, accuracy, max_iterations, rank_deficient, residual_reset)

    def solve(self, scaling_field, divergence, guess, enable_backprop, simulation_physics, offset=0, unrolling_step=0):
        # `guess` is ignored exactly like in the reference (init_with_zeros=True, piso_cuda_pressure_solver.py:95)
        scaling_field = scaling_field if isinstance(scaling_field, StaggeredGrid) else StaggeredGrid(scaling_field)
        a0 = flatten_staggered_data(scaling_field, coord_flip=False).detach().to(torch.float32)   # v first (:70)
        return self.solve_flat(a0, divergence, simulation_physics, unrolling_step=unrolling_step)

    def solve_flat(self, a0_vfirst, divergence, simulation_physics, unrolling_step=0):
        """`solve` with the face coefficients already in the op's layout (flat, v faces first, :70); used by the fused step."""
        dt = torch.float64 if self.cast_to_double else torch.float32
        sharding = getattr(simulation_physics, "sharding", None)
        ny, nx = (int(divergence.shape[1]), int(divergence.shape[2])) if sharding is None else (sharding.ny, sharding.nx)
        dev = divergence.device
        a0 = a0_vfirst
        if self.laplace_rank_deficient is None:                                     # :84-87
            # (looked at once per solver, on the host: the whole-grid masks of a sharded simulation never go to the device as a whole)
            a, c = (as_tensor(m, dtype=torch.float32, device="cpu") for m in (simulation_physics.accessible_mask, simulation_physics.active_mask))
            prod = a * c + (1 - a) * (1 - c)
            prod = torch.prod(prod[0, 0, 1:-1, 0]) * torch.prod(prod[0, -1, 1:-1, 0]) * \
                torch.prod(prod[0, 1:-1, 0, 0]) * torch.prod(prod[0, 1:-1, -1, 0])
            self.laplace_rank_deficient = bool(prod.item() != 0)
        rank_def = self.laplace_rank_deficient
        if isinstance(rank_def, torch.Tensor):
            rank_def = bool(rank_def.reshape(-1)[0].item())
        if (sharding is not None) != bool(self.slab_comm is not None and self.slab_comm.sharded):
            raise ValueError("pressure solve: the simulation's `sharding` and this solver's slab communicator disagree")
        if sharding is None:
            active = simulation_physics.active_mask_tensor(dev).reshape(-1).contiguous()
            accessible = simulation_physics.accessible_mask_tensor(dev).reshape(-1).contiguous()
        else:
            loc = sharding.sim_tensors(simulation_physics, dev)
            active, accessible = loc["active"], loc["accessible"]
        L = laplace_matrix_native(nx, ny, active, accessible, a0, dt, sharding)
        per_y, per_x = [bool(b) for b in simulation_physics.bool_periodic]          # given (y, x), flipped for the op (:95)
        pressure, iteration = _PressureSolveFn.apply(divergence, L, self, nx, ny, per_x, per_y, rank_def)
        self.solve_count = self.solve_count + .001
        return pressure, iteration, L


PisoPressureSolverHip = PisoPressureSolverCudaCustom
