"""Shared synthetic set-ups for the parity tests (inputs only; no reference code).

Each case mirrors a configuration of BASELINE.json / the reference's scripts at test size:
  periodic      decaying turbulence, doubly periodic                       
  xper_ywall    temporally evolving mixing layer: x periodic, y walls, v Dirichlet on the walls
  cavity        lid-driven cavity incl. the solid lid row and no-slip mask   (lid_driven_cavity_2d.py:15-47)
  spatial_ml    spatially evolving mixing layer: inflow/outflow in x, open y (combined_training_integrated.py:481-539)
"""
import numpy as np

f32 = np.float32


def _vel_from_stream(ny, nx, dy, dx, rng, periodic_yx, amp=1.0):
    """Roughly solenoidal random velocity on faces (curl of a smooth random stream function on cell corners)."""
    ky, kx = np.meshgrid(np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
    psi = np.zeros((ny + 1, nx + 1))
    for _ in range(6):
        a, b = rng.integers(1, 4, size=2)
        ph = rng.uniform(0, 2 * np.pi, size=2)
        psi += rng.standard_normal() * np.sin(2 * np.pi * a * ky / ny + ph[0]) * np.sin(2 * np.pi * b * kx / nx + ph[1])
    u = (psi[1:, :] - psi[:-1, :]) / dy          # [ny, nx+1]
    v = -(psi[:, 1:] - psi[:, :-1]) / dx         # [ny+1, nx]
    s = amp / max(1e-12, np.sqrt(np.mean(u ** 2) + np.mean(v ** 2)))
    t = np.zeros((1, ny + 1, nx + 1, 2), f32)
    t[0, :, :nx, 0] = v * s
    t[0, :ny, :, 1] = u * s
    if periodic_yx[1]:
        t[0, :ny, nx, 1] = t[0, :ny, 0, 1]
    if periodic_yx[0]:
        t[0, ny, :nx, 0] = t[0, 0, :nx, 0]
    return t


def make_case(name, ny, nx, seed=0, viscosity=1e-2, cfl=0.5, variable_viscosity=False):
    """Returns a dict of plain numpy inputs understood by both the oracle (OracleSetup) and the product package."""
    rng = np.random.default_rng(seed)
    st = (1, ny + 1, nx + 1, 2)
    ones = np.ones((1, ny + 2, nx + 2, 1), f32)
    c = dict(name=name, ny=ny, nx=nx)
    dmask = np.zeros(st, bool)
    dvals = np.zeros(st, f32)
    no_slip = None
    if name == "periodic":
        L = 2 * np.pi
        c.update(periodic_yx=(True, True), p_ext=("periodic", "periodic"), boundaries="PERIODIC")
        active, accessible = ones.copy(), ones.copy()
        size = (L, L * nx / ny)
    elif name == "xper_ywall":
        c.update(periodic_yx=(False, True), p_ext=(("constant", "constant"), "periodic"), boundaries="(CLOSED, PERIODIC)")
        active, accessible = ones.copy(), ones.copy()
        active[0, 0], active[0, -1], accessible[0, 0], accessible[0, -1] = 0, 0, 0, 0
        dmask[0, 0, :nx, 0] = True
        dmask[0, ny, :nx, 0] = True
        size = (1.0 * ny, 1.0 * nx)
    elif name == "cavity":
        c.update(periodic_yx=(False, False), p_ext=(("boundary", "boundary"), ("boundary", "boundary")), boundaries="OPEN")
        active = np.pad(np.ones((1, ny, nx, 1), f32), ((0, 0), (1, 1), (1, 1), (0, 0)))
        active[0, -2] = 0
        accessible = active.copy()
        dmask[0, 0, :nx, 0] = True
        dmask[0, -2:, :nx, 0] = True
        dmask[0, :ny, 0, 1] = True
        dmask[0, :ny, nx, 1] = True
        dmask[0, ny - 1, :, 1] = True
        dvals[0, ny - 1, :, 1] = 1.0
        ns = np.zeros((ny + 2, nx + 2), bool)
        ns[0, :], ns[-2:, :], ns[:, 0], ns[:, -1] = True, True, True, True
        no_slip = ns.ravel()
        size = (1.0 + 1.0 / (ny - 1), 1.0 * nx / (ny - 1))
    elif name == "spatial_ml":
        c.update(periodic_yx=(False, False), p_ext=(("boundary", "boundary"), ("boundary", "constant")),
                 boundaries="((OPEN, OPEN), (OPEN, CLOSED))")
        active = np.pad(np.ones((1, ny, nx, 1), f32), ((0, 0), (1, 1), (1, 1), (0, 0)))
        accessible = ones.copy()
        accessible[0, :, 0], accessible[0, 0, :], accessible[0, -1, :] = 0, 0, 0
        dmask[0, 0, :nx, 0] = True
        dmask[0, ny, :nx, 0] = True
        dmask[0, :ny, 0, 1] = True
        prof = 0.5 * np.tanh(2.0 * (np.linspace(0, ny, ny + 2)[1:-1] - ny / 2) / (ny / 8)) + 1.0
        dvals[0, :ny, 0, 1] = prof
        no_slip = np.zeros((ny + 2) * (nx + 2), bool)
        size = (1.0 * ny, 1.0 * nx)
    else:
        raise ValueError(name)
    dy, dx = size[0] / ny, size[1] / nx
    vel = _vel_from_stream(ny, nx, dy, dx, rng, c["periodic_yx"])
    if name == "spatial_ml":
        vel[0, :ny, :, 1] += dvals[0, :ny, 0:1, 1]
    if name == "cavity":
        vel[...] *= 0.3
    vel = np.where(dmask, dvals, vel).astype(f32)
    umax = float(np.abs(vel).max())
    dt = cfl * min(dx, dy) / max(umax, 1e-6)
    visc = viscosity
    if variable_viscosity:
        n_u, n_v = (nx + 1) * ny, nx * (ny + 1)
        visc = (viscosity * (1.0 + rng.random(n_u + n_v))).astype(f32)
    p = (0.1 * rng.standard_normal((ny, nx))).astype(f32)
    c.update(dx_yx=(dy, dx), vel=vel, p=p, dt=dt, dirichlet_mask=dmask, dirichlet_values=dvals, active=active,
             accessible=accessible, no_slip=no_slip, viscosity=visc)
    return c


def oracle_setup(c, **kw):
    from oracle.piso_ref import OracleSetup
    return OracleSetup(c["nx"], c["ny"], c["dx_yx"], c["periodic_yx"], c["dirichlet_mask"], c["active"], c["accessible"],
                       no_slip=c["no_slip"], p_extrapolation=c["p_ext"], viscosity=c["viscosity"], **kw)


def product_setup(c, lin_tol=1e-5, lin_max_it=2000, lin_double=False, p_tol=1e-5, p_max_it=2000, p_reset=10, p_double=True,
                  band_rows=0, rank_deficient=None, device="cuda"):
    """The same case as oracle_setup, expressed through the product's drop-in API (diffpiso package)."""
    import torch
    import diffpiso as dp
    ny, nx = c["ny"], c["nx"]
    bnd = eval(c["boundaries"], {"PERIODIC": dp.PERIODIC, "CLOSED": dp.CLOSED, "OPEN": dp.OPEN})
    dy, dx = c["dx_yx"]
    domain = dp.Domain([ny, nx], boundaries=bnd, box=dp.box[0:dy * ny, 0:dx * nx])
    lin = dp.LinearSolverCudaMultiBicgstabILU(accuracy=lin_tol, max_iterations=lin_max_it, cast_to_double=lin_double,
                                              band_rows=band_rows)
    ps = dp.PisoPressureSolverCudaCustom(dx=[], accuracy=p_tol, max_iterations=p_max_it, residual_reset=p_reset,
                                         cast_to_double=p_double)
    if rank_deficient is not None:
        ps.laplace_rank_deficient = rank_deficient
    sim = dp.SimulationParameters(dirichlet_mask=c["dirichlet_mask"], dirichlet_values=c["dirichlet_values"],
                                  active_mask=c["active"], accessible_mask=c["accessible"],
                                  bool_periodic=c["periodic_yx"], no_slip_mask=c["no_slip"], viscosity=c["viscosity"],
                                  linear_solver=lin, pressure_solver=ps)
    dev = torch.device(device)
    vel_t = torch.tensor(c["vel"], device=dev)
    velocity = dp.StaggeredGrid.sample(vel_t, domain=domain)
    p_ext = dp.pressure_extrapolation(domain.boundaries)
    pressure = dp.CenteredGrid(torch.tensor(c["p"], device=dev)[None, :, :, None], box=domain.box, extrapolation=p_ext)
    return dict(domain=domain, sim=sim, velocity=velocity, pressure=pressure, lin=lin, ps=ps, vel_tensor=vel_t)


def pressure_system(nx, ny, walls=False, seed=11):
    """A doubly periodic (or, walls=True, wall-bounded) pressure system at any size for the kernel-level CG tests and benchmarks:
    random A0 face weights in [0.5, 1.5) with consistent periodic duplicates (a symmetric matrix), the HIP Laplace matrix of it
    [nx * ny, 5] float64 on the device and a zero-mean right-hand side."""
    import torch
    from diffpiso.solvers import laplace_matrix_native
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    a0 = 0.5 + torch.rand(nx * (ny + 1) + (nx + 1) * ny, generator=g)
    a0v = a0[:nx * (ny + 1)].view(ny + 1, nx); a0u = a0[nx * (ny + 1):].view(ny, nx + 1)
    a0v[ny] = a0v[0]; a0u[:, nx] = a0u[:, 0]
    a0 = a0.to(dev)
    act = torch.ones((ny + 2, nx + 2))
    if walls:
        act[0, :] = 0; act[-1, :] = 0; act[:, 0] = 0; act[:, -1] = 0
    act = act.reshape(-1).to(dev)
    L = laplace_matrix_native(nx, ny, act, act, a0, torch.float64)
    b = torch.randn(nx * ny, generator=g, dtype=torch.float64).to(dev); b -= b.mean()
    return L, b


# ------------------------------------------------------------------------------------------------ full-size configurations
# (inputs of the full-size oracle fixtures, tests/golden/make_golden_configs.py, and of the GPU tests that load those fixtures)
def tml_case():
    """Config 3 inputs (also used by the GPU test): x periodic, y walls, tanh shear layer + perturbation."""
    ny, nx = 256, 512
    c = make_case("xper_ywall", ny, nx, seed=0, viscosity=1e-3)
    yy = (np.arange(ny) + 0.5) / ny
    c["vel"][0, :ny, :, 1] += np.tanh(2.0 * (yy - 0.5) * 8)[:, None].astype(f32)
    c["vel"] = np.where(c["dirichlet_mask"], c["dirichlet_values"], c["vel"]).astype(f32)
    return c


def sml_case():
    """Config 4 inputs (also used by the GPU test): spatially evolving mixing layer 1024x256 with the sponge viscosity field."""
    from diffpiso.setups import sponge_viscosity_field
    ny, nx = 256, 1024
    c = make_case("spatial_ml", ny, nx, seed=0, viscosity=2e-3)
    c["viscosity"] = sponge_viscosity_field((ny, nx), 2e-3, int(nx * 0.875), 2e-3 * 20)
    return c


CFG4_SIMPAR = dict(HRres=[256, 1024], sponge_ratio=0.875, dx_ratio=1)


def sml_network(dp, torch):
    """The closure of config 4 with seeded weights: VALID padding + restore_shape and zero buffer width
    (spatial_mixing_layer_differentiable_training.py:46,49-50,55), damped so that the forcing stays a perturbation."""
    net, weights, _ = dp.initialise_fullyconv_network([[0, 0], [0, 0]], padding="VALID", restore_shape=True, seed=1,
                                                      initialiser="normal")      # (the draw the committed fixture was made with)
    with torch.no_grad():
        for w in net.weights:
            w.mul_(0.6)
    return net


def sml_wrapper(F):
    def neural_network_wrapper(neural_network, input, fluid, physical_parameters, simulation_parameters, loss_buffer_width, buffer_width):
        # spatial_mixing_layer_differentiable_training.py:6-10: no closure inside the sponge layer
        sponge_start = int(simulation_parameters["HRres"][1] * simulation_parameters["sponge_ratio"]) // simulation_parameters["dx_ratio"]
        out = neural_network(input[:, :, :sponge_start, :])
        return F.pad(out, (0, 0, 0, int(fluid.resolution[1]) - sponge_start))
    return neural_network_wrapper
