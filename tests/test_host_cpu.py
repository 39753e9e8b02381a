"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol include/piso_hip.h declares,
the torch glue (layout, padding, stencils, custom adjoints) reproduces the golden vectors of the reference's own Python
helpers, and the solver entry points refuse to run without a GPU (no silent fallback)."""
import ast
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import diffpiso as dp
from diffpiso import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["periodic", "xper_ywall", "open", "spatial_ml", "closed"]
TOL = dict(rtol=2e-6, atol=2e-6)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "piso_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(piso_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 14, names
    lib = ctypes.CDLL(N.LIB_PATH)
    for n in sorted(names):
        assert hasattr(lib, n), "libpiso_hip.so does not export %s" % n
    assert b"gfx950" in N.lib.piso_version()


def test_deferred_iteration_count_behaves_like_an_int():
    """Asynchronous solves (piso_cg_solve_async_*) leave their iteration count on the device: solvers.DeferredInt reads it on first
    use and sums stay deferred (solver.stats in a training loop never waits)."""
    from diffpiso.solvers import DeferredInt
    a = DeferredInt(0, [torch.tensor([7], dtype=torch.int32)])
    b = DeferredInt(0, [torch.tensor([5], dtype=torch.int32)])
    total = 0
    total += a
    total += b
    total += 3
    assert isinstance(total, DeferredInt) and len(total._pending) == 2          # nothing has been read yet
    assert a.device_tensor().dtype == torch.int32 and int(a.device_tensor()[0]) == 7
    assert total == 15 and int(total) == 15 and total._pending == []
    assert a == 7 and a != 8 and a < 8 and a >= 7 and a % 5 == 2 and abs(a - 9) == 2 and a / 2 == 3.5 and max(a, b) == 7
    assert "%d %s" % (a, b) == "7 5" and [0] * 10 and list(range(10))[b] == 5
    many = 0
    for _ in range(200):
        many += DeferredInt(0, [torch.tensor([2], dtype=torch.int32)])
    assert len(many._pending) <= 65 and many == 400



def test_iteration_counts_leave_the_solver_as_plain_ints():
    """`solver.last_iterations` and whatever is read out of `solver.stats` are ints even after asynchronous solves handed in
    DeferredInt counts: json.dumps / numpy / torch see numbers (bench.py dumps exactly this pattern)."""
    import json
    import diffpiso as dp
    from diffpiso.solvers import DeferredInt
    ps = dp.PisoPressureSolverCudaCustom(dx=[])
    ps.stats.add("iterations", DeferredInt(0, [torch.tensor([5], dtype=torch.int32)]))
    ps.stats.add("iterations", 3)
    ps.stats.add("solves", 2)
    ps.last_iterations = DeferredInt(0, [torch.tensor([7], dtype=torch.int32)])
    assert type(ps.last_iterations) is int and ps.last_iterations == 7 and ps.last_adjoint_iterations is None
    assert json.loads(json.dumps(dict(ps.stats))) == dict(solves=2, iterations=8, adjoint_solves=0, adjoint_iterations=0)
    assert json.dumps({"it": ps.stats["iterations"], "last": ps.last_iterations}) == '{"it": 8, "last": 7}'
    assert int(np.asarray(ps.stats["iterations"])) == 8 and int(torch.tensor(ps.last_iterations)) == 7
    for key in ps.stats:                                   # the reset idiom of bench.py
        ps.stats[key] = 0
    assert sum(ps.stats.values()) == 0

def test_device_constant_uploads_once_and_notices_in_place_changes():
    """grids.device_constant: a small numpy array handed over again every step (the reference's scripts pass sim.dirichlet_values to every
    piso_step) is converted once and found again by identity + checksum; an array modified in place is converted again."""
    from diffpiso.grids import device_constant
    a = np.arange(12, dtype=np.float64).reshape(1, 3, 4, 1)
    t1 = device_constant(a, dtype=torch.float32, device="cpu")
    t2 = device_constant(a, dtype=torch.float32, device="cpu")
    assert t1 is t2 and t1.dtype == torch.float32 and np.array_equal(t1.numpy(), a.astype(np.float32))
    a[0, 1, 2, 0] = -5.0
    t3 = device_constant(a, dtype=torch.float32, device="cpu")
    assert t3 is not t1 and float(t3[0, 1, 2, 0]) == -5.0
    b = a.copy()
    assert device_constant(b, dtype=torch.float32, device="cpu") is not t3          # another array object: its own entry
    big = np.zeros(1 << 19, dtype=np.float32)                                        # 2 MiB: too large for the cache, plain conversion
    assert device_constant(big, device="cpu") is not device_constant(big, device="cpu")
    t = torch.ones(3)
    assert device_constant(t, device="cpu") is t                                     # tensors pass through


def test_workspace_queries_and_nnz_closed_form():
    assert N.lib.piso_cg_workspace_bytes(2048, 2048, 8) > 9 * 2048 * 2048 * 8
    assert N.lib.piso_bicgstab_workspace_bytes(64, 64, 4) > 18 * (65 * 64 + 64 * 65) * 4
    for (nx, ny, px, py) in [(4, 3, 1, 1), (4, 3, 0, 0), (64, 65, 0, 0), (256, 256, 1, 1), (512, 256, 1, 0)]:
        a, b = ctypes.c_int(), ctypes.c_int()
        N.lib.piso_csr_nnz(nx, ny, px, py, ctypes.byref(a), ctypes.byref(b))
        from oracle.native import matrix_sizes
        assert (a.value, b.value) == matrix_sizes(nx, ny, px, py)[2:]
    a, b = ctypes.c_int(), ctypes.c_int()
    N.lib.piso_csr_nnz(64, 65, 0, 0, ctypes.byref(a), ctypes.byref(b))
    assert (a.value, b.value) == (20865, 20860)           # SURVEY.md section 8 table, LDC 64


def test_no_cpu_fallback():
    """Solver calls with CPU tensors must fail loudly, never compute."""
    lin = dp.LinearSolverCudaMultiBicgstabILU()
    t = torch.zeros(10)
    with pytest.raises(N.PisoNativeError):
        lin.solve(t, t.int(), t.int(), t, (1, 5, 5, 2))
    if not torch.cuda.is_available():
        bad = N.lib.piso_cg_solve_f64(8, 8, 1, 1, None, None, None, ctypes.c_float(1e-3), 10, 1, 10, None, None, 0, None)
        assert bad != 0


def _load(golden_dir, name):
    d = np.load(os.path.join(golden_dir, "helpers_%s.npz" % name))
    vel_ext = ast.literal_eval(str(d["velocity_extrapolation"]))
    p_ext = ast.literal_eval(str(d["pressure_extrapolation"]))
    ny, nx = d["resolution"]
    dy, dx = d["dx_yx"]
    box = dp.box[0:dy * ny, 0:dx * nx]
    vel = dp.StaggeredGrid(torch.tensor(d["vel_in"]), box, extrapolation=vel_ext)
    p = dp.CenteredGrid(torch.tensor(d["p_in"]), box, extrapolation=p_ext)
    return d, vel, p


@pytest.mark.parametrize("name", CASES)
def test_glue_layout_against_reference_golden(golden_dir, name):
    d, vel, p = _load(golden_dir, name)
    ny, nx = d["resolution"]
    np.testing.assert_array_equal(vel.staggered_tensor().numpy(), d["vel_tensor"])
    np.testing.assert_array_equal(dp.flatten_staggered_data(vel, True).numpy(), d["flat_ufirst"])
    np.testing.assert_array_equal(dp.flatten_staggered_data(vel, False).numpy(), d["flat_vfirst"])
    st = (1, ny + 1, nx + 1, 2)
    np.testing.assert_array_equal(dp.stagger_flattened_data(torch.tensor(d["flat_ufirst"]), st, True).numpy(), d["restagger_ufirst"])
    np.testing.assert_array_equal(dp.stagger_flattened_data(torch.tensor(d["flat_vfirst"]), st, False).numpy(), d["restagger_vfirst"])
    np.testing.assert_array_equal(dp.padded_velocity_flat(vel).numpy(), d["vel_padded_flat"])
    v_pad, u_pad = dp.custom_padded(vel, 1)
    assert tuple(u_pad.shape) == tuple(d["padded_u_shape"]) and tuple(v_pad.shape) == tuple(d["padded_v_shape"])
    got = dp.arrange_rhs_term_tf(torch.tensor(d["rhs_in"]), d["dirichlet_mask"], d["dirichlet_values"], 1.0, coord_flip=True)
    np.testing.assert_allclose(got.numpy(), d["rhs_arranged"], **TOL)


class _Sim(object):
    def __init__(self, acc):
        self.acc = torch.tensor(acc)

    def accessible_mask_tensor(self, device):
        return self.acc


@pytest.mark.parametrize("name", CASES)
def test_glue_stencils_and_custom_adjoints_against_reference_golden(golden_dir, name):
    d, vel, p = _load(golden_dir, name)
    ny, nx = d["resolution"]
    np.testing.assert_allclose(dp.finite_volume_gradient_tensor(p, _Sim(d["accessible_mask"])).numpy(), d["fv_gradient_masked"], **TOL)
    np.testing.assert_allclose(dp.finite_volume_gradient_tensor(p, None).numpy(), d["fv_gradient_nomask"], **TOL)
    t = torch.tensor(d["vel_in"]).requires_grad_(True)
    v = dp.StaggeredGrid(t, vel.box, extrapolation=vel.extrapolation)
    div = dp.finite_volume_divergence(v)
    np.testing.assert_allclose(div.detach().numpy(), d["fv_divergence"], **TOL)
    div.backward(torch.tensor(d["div_adj_in"]))
    np.testing.assert_allclose(t.grad.numpy(), d["div_adj_out"], **TOL)     # includes the reference's periodic quirk (C-7)
    for dim in (1, 2):
        key = "circ_grad_adj_in_dim%d" % dim
        if key in d.files:                                                  # C-8: TF-semantics custom gradient
            from diffpiso.stencils import _PeriodicAxisGradient
            q = torch.tensor(d["p_in"]).requires_grad_(True)
            out = _PeriodicAxisGradient.apply(q, dim)
            np.testing.assert_allclose(out.detach().numpy(), d["circ_grad_fwd_dim%d" % dim], **TOL)
            out.backward(torch.tensor(d[key]))
            np.testing.assert_allclose(q.grad.numpy(), d["circ_grad_adj_out_dim%d" % dim], **TOL)


def test_exact_adjoint_switch_is_a_true_transpose():
    from diffpiso import stencils
    dom = dp.Domain([6, 5], boundaries=dp.PERIODIC)
    rng = np.random.default_rng(0)
    stencils.REFERENCE_ADJOINTS = False
    try:
        t = torch.tensor(rng.standard_normal((1, 7, 6, 2)).astype(np.float32))
        t[0, :, 5, 0] = 0
        t[0, 6, :, 1] = 0
        t.requires_grad_(True)
        div = dp.finite_volume_divergence(dp.StaggeredGrid(t, dom.box, extrapolation="periodic"))
        g = torch.tensor(rng.standard_normal(tuple(div.shape)).astype(np.float32))
        div.backward(g)
        dt = torch.tensor(rng.standard_normal((1, 7, 6, 2)).astype(np.float32))
        dt[0, :, 5, 0] = 0
        dt[0, 6, :, 1] = 0
        lhs = float((dp.finite_volume_divergence(dp.StaggeredGrid(dt, dom.box, extrapolation="periodic")) * g).sum())
        rhs = float((t.grad * dt).sum())
        assert abs(lhs - rhs) < 1e-4 * max(1, abs(lhs))
    finally:
        stencils.REFERENCE_ADJOINTS = True


def test_field_shim_matches_phiflow_conventions(golden_dir):
    d = np.load(os.path.join(golden_dir, "mixing_layer_masks.npz"))
    assert list(dp.calculate_staggered_shape(1, np.array([6, 9]))) == list(d["calc_staggered_shape"])
    assert list(dp.calculate_centered_shape(1, np.array([6, 9]))) == list(d["calc_centered_shape"])
    dom = dp.Domain([6, 9], boundaries=((dp.OPEN, dp.OPEN), (dp.OPEN, dp.CLOSED)), box=dp.box[0:6, 0:9])
    assert dp.Material.extrapolation_mode(dom.boundaries) == ("constant", ("constant", "boundary"))
    assert dp.pressure_extrapolation(dom.boundaries) == ("boundary", ("boundary", "constant"))
    assert dp.pressure_extrapolation(dp.Domain([4, 4], boundaries=dp.PERIODIC).boundaries) == "periodic"
    assert dp.Domain([4, 5], boundaries=(dp.CLOSED, dp.PERIODIC)).boundaries == (dp.CLOSED, dp.PERIODIC)
    g = dom.staggered_grid(1.0)
    assert tuple(g.staggered_tensor().shape) == (1, 7, 10, 2)
    assert float(g.staggered_tensor()[0, :, 9, 0].abs().sum()) == 0
    assert np.allclose(dom.dx, [1.0, 1.0])


def test_mixing_layer_masks_against_reference_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "mixing_layer_masks.npz"))
    ny, nx = 6, 9
    bcy = np.zeros((1, 1, nx + 2, 1), np.float32)
    m, v, n, act, acc = dp.compute_mixingLayer_masks(d["staggered_shape"], ((True, True), (True, False)),
                                                     ((bcy, bcy), (d["bcx"], [])))
    np.testing.assert_array_equal(m, d["dirichlet_mask"])
    np.testing.assert_array_equal(v, d["dirichlet_values"])
    np.testing.assert_array_equal(n, d["neumann_mask"])
    np.testing.assert_array_equal(act, d["active_mask"])
    np.testing.assert_array_equal(acc, d["accessible_mask"])
    upd = dp.update_dirichlet_values(d["dirichlet_values"], ((False, False), (True, False)), (([], []), (d["update_in"], [])))
    np.testing.assert_array_equal(upd, d["updated_values"])


@pytest.mark.parametrize("name", ["periodic_cubic", "xper_ywall_cubic", "spatial_ml", "closed"])
def test_closure_coupling_against_reference_golden(golden_dir, name):
    d, vel, p = _load(golden_dir, name)
    np.testing.assert_allclose(vel.at_centers().data.numpy(), d["at_centers"], **TOL)
    np.testing.assert_allclose(dp.centered_gradient(p).numpy(), d["pressure_gradient"], **TOL)
    np.testing.assert_allclose(dp.centered_to_staggered(torch.tensor(d["nn_out"])).numpy(), d["nn_forcing"], **TOL)


def test_closure_network_shapes_and_gradients():
    net, weights, rbw = dp.initialise_fullyconv_network(None, padding="SAME", seed=0)
    assert [tuple(w.shape) for w in weights] == [(16, 4, 7, 7), (16, 16, 5, 5), (32, 16, 5, 5), (64, 32, 3, 3), (64, 64, 3, 3),
                                                 (64, 64, 1, 1), (2, 64, 1, 1)] and rbw == 9
    x = torch.randn(1, 20, 24, 4)
    y = net(x)
    assert tuple(y.shape) == (1, 20, 24, 2)
    y.sum().backward()
    assert all(w.grad is not None and torch.isfinite(w.grad).all() for w in weights)
    netv, _, rbwv = dp.initialise_fullyconv_network([[2, 2], [1, 3]], padding="VALID", restore_shape=True, seed=0)
    x = torch.randn(1, 40, 44, 4)
    assert tuple(netv(x).shape) == (1, 40, 44, 2) and rbwv == [[11, 11], [10, 12]]
    z = netv(x)
    assert float(z[:, :2].abs().sum()) == 0 and float(z[:, :, :1].abs().sum()) == 0      # buffer zones are zero padded


def test_run_piso_steps_has_the_reference_signature():
    """diffpiso/combined_training_integrated.py:396-397: 14 parameters in this order, the last two optional."""
    import inspect
    sig = inspect.signature(dp.run_piso_steps)
    assert list(sig.parameters) == ["velocity", "pressure", "domain", "physical_parameters", "simulation_parameters", "training_dict",
                                    "neural_network", "neural_network_wrapper", "sim_physics", "viscosity_field", "bcx",
                                    "bc_placeholders", "dirichlet_placeholder_update", "loss_buffer_width"]
    assert sig.parameters["dirichlet_placeholder_update"].default is None and sig.parameters["loss_buffer_width"].default is None
    assert all(p.default is inspect.Parameter.empty for n, p in list(sig.parameters.items())[:12])


def test_linear_solver_scipy_forward_and_transpose_adjoint():
    """diffpiso/linear_solver.py:33-57: direct solve; gradient w.r.t. rhs = transposed solve."""
    import scipy.sparse as sp
    A = (sp.random(30, 30, 0.2, format="csr", random_state=3) + sp.eye(30) * 4).tocsr()
    s = dp.LinearSolverScipy()
    assert s.supported_devices == "CPU" and not s.supports_guess
    mv, rp, ci = torch.tensor(A.data, dtype=torch.float32), torch.tensor(A.indptr), torch.tensor(A.indices)
    b = torch.randn(30, generator=torch.Generator().manual_seed(0)).requires_grad_(True)
    g = torch.randn(30, generator=torch.Generator().manual_seed(1))
    x = s.solve(mv, rp, ci, b)
    np.testing.assert_allclose(A @ x.detach().numpy(), b.detach().numpy(), atol=2e-6)
    (x * g).sum().backward()
    np.testing.assert_allclose(A.T @ b.grad.numpy(), g.numpy(), atol=2e-6)
    xt = s.solve(mv, rp, ci, b.detach(), transpose=True)
    np.testing.assert_allclose(A.T @ xt.numpy(), b.detach().numpy(), atol=2e-6)


def test_frame_file_format(tmp_path):
    """<field>_%06d.npz / arr_0, windows of step_count+1 frames spaced dt_ratio apart, [1,T,...] stacking
    (diffpiso/datamanagement.py:35-57 describes the format; spatial_mixing_layer.py:60-75 writes it)."""
    d = str(tmp_path) + "/"
    for f in range(9):
        dp.save_frame(d, "velocity", f, np.full((1, 3, 4, 2), f, np.float64))
        dp.save_frame(d, "pressure", f, np.full((1, 2, 3, 1), 10 + f, np.float64))
    assert sorted(os.listdir(d))[0] == "pressure_000000.npz"
    assert list(np.load(d + "velocity_000003.npz").keys()) == ["arr_0"]
    lists = dp.data_path_assembler([d], ["velocity", "pressure"], [[(float(i), 0.5) for i in range(9)]], [1], [8], [2], dt_ratio=2)
    assert len(lists) == 3 and len(lists[0]) == 8 - 2 * 2 == len(lists[2])
    assert lists[0][0] == [d + "velocity_%06d.npz" % k for k in (1, 3, 5)]
    assert lists[2][1] == (1.0, 0.5)                        # per-frame characteristics are indexed by the window start
    vel, prs, ch = dp.load_function(lists[0][1], lists[1][1], lists[2][1])
    assert vel.shape == (1, 3, 3, 4, 2) and vel.dtype == np.float32 and prs.shape == (1, 3, 2, 3, 1)
    assert [float(vel[0, t, 0, 0, 0]) for t in range(3)] == [2.0, 4.0, 6.0] and ch.shape == (1, 2)
    lists = dp.data_path_assembler([d], ["velocity"], [7.5], [0], [6], [2])
    assert len(lists[0]) == 4 and lists[1] == [7.5] * 4


def test_build_from_a_tree_without_binaries(tmp_path):
    """__graft_entry__.build() must produce libpiso_hip.so (hipcc, gfx950) and the oracle libraries from sources alone: run it in
    a copy of the tree that holds no .so / .o file (the tree gpurun ships contains prebuilt ones, which proves nothing)."""
    import shutil
    import subprocess
    import sys
    dst = tmp_path / "tree"
    ignore = shutil.ignore_patterns("*.so", "*.o", "_obj", "_build", "_bin", "gpurun_out", ".git", "__pycache__", "profiles", "golden")
    shutil.copytree(ROOT, dst, ignore=ignore)
    assert not list(dst.rglob("*.so"))
    code = "import __graft_entry__ as g; g.build(); import diffpiso._native as N; print(N.LIB_PATH)"
    out = subprocess.run([sys.executable, "-c", code], cwd=dst, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    built = out.stdout.strip().splitlines()[-1]
    assert built.startswith(str(dst)) and os.path.isfile(built)
    assert os.path.isfile(dst / "oracle" / "_build" / "libpiso_oracle.so") and os.path.isfile(dst / "oracle" / "_build" / "libpiso_oracle_omp.so")
    lib = ctypes.CDLL(built)
    assert b"gfx950" in ctypes.cast(lib.piso_version, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_bench_auto_policy_tries_peer_then_rccl_then_reports_replicas():
    """bench.py's `--decomp auto` policy for the N > 1 headline (bench.sharded_headline), with stub attempts: the peer transport
    first; the RCCL transport only if the environment REFUSES the peer transport and every rank has its own GPU; a refused transport
    is reported next to the replicas figure with exit code 0, a run that failed or hung costs exit code 3 - and nothing is tried
    after a failure."""
    import bench
    line = {"value": 30.0, "ms_per_step": 266.0, "sharded": {"ranks_seen": 8}}
    refused = {"sharded_unavailable": "hipIpcGetMemHandle: invalid argument"}

    def script(*answers):
        calls, it = [], iter(answers)

        def attempt(transport, port_offset, limit_s):
            calls.append((transport, port_offset, limit_s))
            return next(it)
        return attempt, calls

    # 1. the peer transport works: its line is the headline, nothing else is tried
    att, calls = script((line, None, True, False))
    assert bench.sharded_headline(att, False) == (line, None, None, 0) and [c[0] for c in calls] == ["peer"]
    # 2. peer refused by the environment, one GPU per rank: RCCL is tried once and carries the headline
    att, calls = script((refused, None, True, True), (line, None, True, False))
    assert bench.sharded_headline(att, False) == (line, None, None, 0) and [c[0] for c in calls] == ["peer", "rccl"]
    assert calls[0][1] != calls[1][1]                                  # the two attempts rendezvous on different ports
    # 3. peer refused and the RCCL attempt fails / hangs: reported, replicas only, exit code 0 (that path cannot be tested before it meets a node)
    att, calls = script((refused, None, True, True), (None, {"error": "timed out after 420 s", "last_stage": "transport set-up (rccl, 8 ranks)"}, False, False))
    child, skipped, err, rc = bench.sharded_headline(att, False)
    assert child is None and err is None and rc == 0 and [c[0] for c in calls] == ["peer", "rccl"]
    assert "hipIpcGetMemHandle" in skipped and "RCCL transport was tried instead" in skipped and "last_stage" in skipped
    # 4. peer refused while the ranks share one GPU (RCCL refuses two ranks per device): nothing else to try
    att, calls = script((refused, None, True, True))
    child, skipped, err, rc = bench.sharded_headline(att, True)
    assert child is None and rc == 0 and "could not be set up" in skipped and [c[0] for c in calls] == ["peer"]
    # 5. the peer run FAILED on some rank (not refused): exit code 3, replicas only, RCCL is not a fall-back for failures
    att, calls = script((None, {"error": "the sharded run (peer transport) ended with code 1", "stderr_tail": "..."}, False, False))
    child, skipped, err, rc = bench.sharded_headline(att, False)
    assert child is None and skipped is None and rc == 3 and "ended with code 1" in err["error"] and [c[0] for c in calls] == ["peer"]
    # 6. ... also when this rank's own attempt looked fine but another rank's did not
    att, calls = script((line, None, False, False))
    assert bench.sharded_headline(att, False)[3] == 3


def test_closure_initialiser_is_tensorflows_glorot_normal():
    """networks.py:57 draws the closure's weights from `tf.glorot_normal_initializer()`: a normal distribution truncated at two standard
    deviations whose standard deviation AFTER the truncation is sqrt(2 / (fan_in + fan_out)).  A random draw cannot be compared value
    for value; its support and its scale can."""
    import numpy as np
    from diffpiso.closure import FullyConvNetwork
    net = FullyConvNetwork(seed=5)
    for w in net.weights:
        cout, cin, k, _ = w.shape
        std = np.sqrt(2.0 / (k * k * cin + k * k * cout))
        sig = std / 0.87962566103423978
        assert float(w.abs().max()) <= 2.0 * sig * (1 + 1e-6)                       # truncated at two sigma
        if w.numel() >= 4096:                                                        # (enough samples for a 5 % statement)
            assert abs(float(w.std()) / std - 1.0) < 0.05, (tuple(w.shape), float(w.std()), std)
