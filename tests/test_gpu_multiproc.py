"""The multi-GPU code path of the pressure CG with real PROCESSES: W ranks, one process each, mailboxes mapped
across processes with hipIpc handles, in-kernel exchanges through peer-mapped memory.  On the one-GPU test box all ranks share
cuda:0 (their kernels run concurrently on disjoint CUs); what this cannot cover is the xGMI hop itself."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_parent_gpu_memory():
    """The ranks are child processes on the SAME GPU: hand back what this (pytest) process has cached there - after the full-size
    fixture tests that is tens of GB, and eight children of a 4096^2 step on top of it have run one of them out of memory."""
    import gc
    import torch
    gc.collect()
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()



def run_ranks(world, nx, ny, walls, timeout=600, worker="slab_worker.py", extra=None, env_extra=None):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    _free_parent_gpu_memory()
    tail = [str(nx), str(ny), str(int(walls)), "1"] if extra is None else extra
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", worker), str(r), str(world), str(port)] + tail,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(world)]
    res = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        lines = [l for l in so.splitlines() if l.startswith("SLAB_WORKER ")]
        assert lines, "rank died without a report:\n%s\n%s" % (so[-2000:], se[-4000:])
        res.append(json.loads(lines[-1][len("SLAB_WORKER "):]))
    # a box that cannot map device memory across processes (hipIpc refused) cannot run these tests at all: skipped, not failed
    for r in res:
        if not r.get("ok") and ("piso_comm_peer_create" in r.get("error", "") or "piso_comm_peer_connect" in r.get("error", "")
                                or "the peer transport could not be set up" in r.get("error", "")):
            pytest.skip("peer transport unavailable here: %s" % r["error"][:300])
    return sorted(res, key=lambda r: r["rank"])


@pytest.mark.parametrize("world,nx,ny,walls", [(2, 1024, 1024, False), (2, 2048, 2048, False), (2, 2048, 2048, True),
                                               (4, 2048, 2048, False),
                                               # eight ranks x 32 workgroups = every CU of the one GPU: all eight waves of a workgroup poll a rank each
                                               (8, 2048, 2048, False)])
def test_slab_cg_over_processes(world, nx, ny, walls):
    res = run_ranks(world, nx, ny, walls)
    for r in res:
        assert r["ok"], r
        print(r)
        for label in ("persistent", "two_kernel"):
            # fixed short runs: round-off level (summation grouping; the persistent kernel's merged reductions)
            assert max(r[label]["fixed_run_diffs"]) <= 2e-10, (label, r[label])
            ita, itb = r[label]["converged_its"]
            # (the walled 2048^2 system needs more than the 20000 iterations allowed: then both sides must stop at the cap)
            assert (ita == itb == 20000) or (ita < 20000 and itb < 20000 and abs(ita - itb) <= max(10, 0.1 * ita)), (label, ita, itb)
            assert r[label]["converged_diff"] <= 1e-3          # both stop at max|r| < 1e-7 (tolerance / smallest eigenvalue)
        # the NORMAL iterations really ran inside the persistent slab kernel, and no segment had to be repeated
        assert r["stats"]["transport"] == "peer"
        assert r["stats"]["persistent_iterations"] > 150 and r["stats"]["persistent_fallbacks"] == 0, r["stats"]
        assert r["stats"]["solves_verified"] >= 5 and r["stats"]["verification_failures"] == 0, r["stats"]   # r == b - A^ x checked per solve
    # every rank took the same decisions
    assert len({tuple(r["persistent"]["converged_its"]) for r in res}) == 1


@pytest.mark.parametrize("env,expect", [({"PISO_PEER_MAP": "fd"}, "fd"),                       # the second mechanism on its own
                                        ({"PISO_TEST_REFUSE_PEER": "ipc"}, "fd"),              # ONE rank's hipIpc attempt fails: everybody moves on to it
                                        ({}, "ipc")])
def test_slab_cg_over_processes_mailboxes_mapped_through_file_descriptors(env, expect):
    """The peer transport's SECOND way to map the mailboxes (for nodes that refuse hipIpc handles across ranks): exportable
    virtual-memory allocations (hipMemCreate, uncached), shared as POSIX file descriptors over Unix sockets, imported and mapped by the
    other rank - then the very same kernels: persistent slab CG, verified solves, and the mailbox ping-pong of the hop matrix."""
    res = run_ranks(2, 1024, 1024, False, env_extra=env)
    for r in res:
        assert r["ok"], r
        assert r["peer_map"] == expect, r["peer_map"]
        for label in ("persistent", "two_kernel"):
            assert max(r[label]["fixed_run_diffs"]) <= 2e-10, (label, r[label])
        assert r["stats"]["transport"] == "peer" and r["stats"]["persistent_iterations"] > 150 and r["stats"]["persistent_fallbacks"] == 0
        assert r["stats"]["verification_failures"] == 0
        m = r["hop_us_matrix"]
        print(expect, "hop matrix [us, one way]:", m)
        assert len(m) == 2 and all(len(row) == 2 for row in m)
        assert all(0.05 < m[a][b] < 200.0 for a in range(2) for b in range(2)), m       # measured, symmetric by construction
        assert m[0][1] == m[1][0]
    assert res[0]["hop_us_matrix"] == res[1]["hop_us_matrix"]                              # every rank holds the same matrix


def test_slab_cg_config5_slab_shape_four_row_regions_over_two_processes():
    """The slab kernel instance BASELINE config 5 runs - 4096 columns, regions of FOUR rows, two per wave - over REAL processes: a
    4096 x 512 grid on two ranks (slabs of 4096 x 256: two launches of 128 workgroups fit the one GPU side by side), persistent kernel
    forced, every fixed run up to 150 iterations against the one-GPU two-kernel iteration at 2e-10 - the bar the 2048^2 instance is
    held to.  (Eight slabs of 4096 x 512 need eight GPUs; the 8-rank tests on one GPU run the two-kernel iteration.)"""
    res = run_ranks(2, 4096, 512, False, extra=["4096", "512", "0", "1", "4"])
    for r in res:
        assert r["ok"], r
        print(r)
        for label in ("persistent", "two_kernel"):
            # fixed runs of 1, 2, 7, 45 iterations: round-off level.  After 150 iterations on this 8 : 1 grid two summation orders of the
            # SAME iteration are 1e-8 apart (the system does not converge within 20 000 iterations: round-off is amplified along the
            # way) - the slab's two-kernel iteration, which is not in question, shows exactly that (measured 1.1e-8, the persistent
            # kernel 2.1e-8): the persistent kernel is held to the same order of magnitude there
            assert max(r[label]["fixed_run_diffs"][:4]) <= 2e-10, (label, r[label])
            assert r[label]["fixed_run_diffs"][4] <= 1e-7, (label, r[label])
            ita, itb = r[label]["converged_its"]
            assert (ita == itb == 20000) or (ita < 20000 and itb < 20000 and abs(ita - itb) <= max(10, 0.1 * ita)), (label, ita, itb)
        assert r["persistent"]["fixed_run_diffs"][4] <= 5 * max(r["two_kernel"]["fixed_run_diffs"][4], 1e-9), r      # (a rank whose rows happen to agree to 1e-12 on one path compares noise)
        assert r["stats"]["persistent_iterations"] > 150 and r["stats"]["persistent_fallbacks"] == 0, r["stats"]
        assert r["stats"]["solves_verified"] >= 5 and r["stats"]["verification_failures"] == 0, r["stats"]


def test_slab_cg_four_row_regions_over_eight_processes():
    """Regions of FOUR rows, two per wave (config 5's instance) with EIGHT ranks: a 1024 x 1024 grid, slabs of 128 rows = 16 workgroups
    per rank, all eight persistent kernels side by side on the one GPU; every wave of a workgroup polls a rank of its own."""
    res = run_ranks(8, 1024, 1024, False, extra=["1024", "1024", "0", "1", "4"])
    for r in res:
        assert r["ok"], r
        for label in ("persistent", "two_kernel"):
            assert max(r[label]["fixed_run_diffs"]) <= 2e-10, (label, r[label])
            ita, itb = r[label]["converged_its"]
            assert (ita == itb == 20000) or (ita < 20000 and itb < 20000 and abs(ita - itb) <= max(10, 0.1 * ita)), (label, ita, itb)
            assert r[label]["converged_diff"] <= 1e-3
        assert r["stats"]["persistent_iterations"] > 150 and r["stats"]["persistent_fallbacks"] == 0, r["stats"]
        assert r["stats"]["solves_verified"] >= 5 and r["stats"]["verification_failures"] == 0, r["stats"]
    assert len({tuple(r["persistent"]["converged_its"]) for r in res}) == 1


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N > 1 path end to end (torch.distributed.run, max-over-ranks timing, the slab self-check inside the JSON line)
    with two ranks sharing the box's one GPU (PISO_BENCH_SHARE_GPU=1: gloo instead of RCCL for torch.distributed, the library's
    peer transport for the solver).  The timings of this mode mean nothing; the agreement figures do."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PISO_BENCH_SHARE_GPU="1")
    _free_parent_gpu_memory()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--grid", "1024",
           "--max-iterations", "300"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and lines, (p.returncode, p.stdout[-2000:], p.stderr[-3000:])
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    # the headline of an N > 1 run is the SHARDED step (one 1024 x 2048 box here, a 1024^2 slab per rank), measured in a child process per
    # rank; the replicas figure stands beside it
    assert d.get("sharded_run") is None, d.get("sharded_run")
    assert d["sharded"]["ranks_seen"] == 2 and d["sharded"]["halo_exchanges"] > 0 and d["sharded"]["verification_failures"] == 0
    # the first record from a multi-GPU node diagnoses itself: how the mailboxes were mapped and the measured hop between every pair
    assert d["sharded"]["peer_map"] in ("ipc", "fd") and d["sharded"]["hop_us_matrix_error"] is None
    hop = d["sharded"]["hop_us_matrix"]
    assert len(hop) == 2 and 0.05 < hop[0][1] == hop[1][0] < 200 and d["sharded"]["hop_us"]["ring_neighbours_max"] == hop[0][1]
    assert d["config"]["grid"] == [2048, 1024] and d["replicas"]["value"] > 0 and d["parallel_efficiency_vs_replicas"] > 0
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    chk = d["slab_cg_self_check"]
    print(chk)
    assert chk["ok"] and chk["ok_all_ranks"], chk
    assert chk["strong"]["max_rel_diff_persistent"] < 1e-9 and chk["strong"]["max_rel_diff_two_kernel"] < 1e-9
    assert chk["weak"]["max_rel_diff_vs_single_gpu_tall_grid"] < 1e-9
    assert chk["strong"]["persistent_iterations"] > 0 and chk["persistent_fallbacks"] == 0


@pytest.mark.parametrize("world,name,nx,ny", [(2, "periodic", 48, 64), (2, "spatial_ml", 40, 32), (4, "xper_ywall", 36, 64),
                                              (4, "periodic", 128, 128)])
def test_slab_bicgstab_over_processes(world, name, nx, ny):
    """The distributed ILU(0)-BiCGStab: slabs cut at band edges carry the single-GPU preconditioner, so iteration counts are the
    single-GPU ones and the solutions agree to summation order (float64 1e-9; float32: its +-1, 2e-5 like the oracle test)."""
    res = run_ranks(world, nx, ny, False, worker="bicg_worker.py", extra=[name, str(nx), str(ny)])
    for r in res:
        assert r["ok"], r
        print(r)
        assert r["nan_warn"] == 1
        for run in r["runs"]:
            assert run["warn"] == [0, 0]
            if "float64" in run["dtype"]:
                assert run["its_single"] == run["its_slab"], run
                assert run["rel_diff"] <= 1e-9, run
            else:
                assert max(abs(a - b) for a, b in zip(run["its_single"], run["its_slab"])) <= 1, run
                assert run["rel_diff"] <= 2e-5, run
    assert len({json.dumps(r["runs"]) for r in res}) == 1          # every rank saw the same iteration counts and the same solution


def _bench(env_extra, args, nproc):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    _free_parent_gpu_memory()
    if nproc > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900, cwd=ROOT)
    launcher_trouble = ("EADDRINUSE", "Address already in use", "RendezvousConnectionError", "RendezvousTimeoutError", "Timed out waiting for",
                        "DistStoreError", "Connection reset by peer", "The client socket has timed out", "failed to connect to")
    if p.returncode != 0 and nproc > 1 and any(sig in p.stderr for sig in launcher_trouble) and '"metric"' not in p.stdout:
        # ONE retry on a fresh port, and only for a failure of the LAUNCHER / rendezvous (a port still in use, a store time-out: eight
        # processes starting on one box right behind another multi-process test have tripped over that once in ~20 runs of the suite).
        # Anything the code under test does wrong - a mailbox wait that gave up, a sequence race, exit code 3 of a failed sharded
        # child - fails on the FIRST attempt: intermittent bugs must not be retried into passes.
        print("first attempt failed in the launcher (code %d), stderr tail:\n%s" % (p.returncode, p.stderr[-1500:]))
        s2 = socket.socket()
        s2.bind(("127.0.0.1", 0))
        port2 = s2.getsockname()[1]
        s2.close()
        cmd = [str(port2) if c == str(port) else c for c in cmd]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 and ("piso_comm_peer_create" in p.stderr or "piso_comm_peer_connect" in p.stderr):
        pytest.skip("peer transport unavailable here")
    if p.returncode != 0 or not lines:
        print("bench.py %s ended with code %d; stderr tail:\n%s" % (" ".join(args), p.returncode, p.stderr[-6000:]))     # (untruncated, unlike the assertion's repr)
    assert p.returncode == 0 and lines, (p.returncode, p.stdout[-2000:], p.stderr[-3000:])
    return json.loads(lines[-1])


def test_bench_reports_a_refused_peer_transport_instead_of_failing():
    """Where the environment refuses the peer transport (here: a test knob makes ONE rank's hipIpc set-up fail) the set-up fails as a
    collective on every rank, the sharded child says so and leaves with code 0, and bench.py prints the replicas line with
    `sharded_run.skipped` - exit code 0, nothing hangs.  (With one GPU per rank the RCCL transport would be tried next.)"""
    d = _bench({"PISO_BENCH_SHARE_GPU": "1", "PISO_BENCH_SLAB_CHECK": "0", "PISO_TEST_REFUSE_PEER": "1"},
               ["--gpus", "2", "--steps", "1", "--warmup", "0", "--grid", "256", "--no-cpu-baseline", "--no-extras", "--max-iterations", "100"], 2)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d.get("replicas_only") is True
    # (both mapping mechanisms were tried - hipIpc handles, then shared file descriptors - and the refusing rank is named for each)
    assert "ipc: failed on rank(s) [1]" in d["sharded_run"]["skipped"] and "fd: failed on rank(s) [1]" in d["sharded_run"]["skipped"] and "sharded" not in d


def test_decomposed_step_two_ranks_matches_one_gpu():
    """The whole benchmark step (forward PISO steps + reverse sweep) with BOTH linear solvers cut into two slabs over two
    processes -- pressure CG through the persistent slab kernel, ILU(0)-BiCGStab with all-reduced dot products, forward and
    transposed solves -- against the same run on one GPU: the loss and the CG / BiCGStab iteration counts must agree."""
    common = ["--steps", "2", "--warmup", "0", "--grid", "256", "--no-cpu-baseline", "--no-extras", "--tol", "1e-7"]
    one = _bench({}, ["--gpus", "1"] + common, 1)
    two = _bench({"PISO_BENCH_SHARE_GPU": "1", "PISO_BENCH_SLAB_CHECK": "0"}, ["--gpus", "2", "--decomp", "slab"] + common, 2)
    assert two["scaling"] == "strong" and two["n_gpus"] == 2
    l1, l2 = one["config"]["loss"], two["config"]["loss"]
    print(one["config"], two["config"])
    assert abs(l1 - l2) <= 1e-5 * abs(l1), (l1, l2)
    assert abs(one["config"]["grad_norm"] - two["config"]["grad_norm"]) <= 1e-5 * one["config"]["grad_norm"]     # |dL/du_0|, summed over the ranks
    assert one["config"]["last_bicgstab_iterations"] == two["config"]["last_bicgstab_iterations"]
    # the WHOLE step is sharded: every rank assembled / updated its own rows only and nothing was all-gathered
    assert two["sharded"]["ranks_seen"] == 2 and two["sharded"]["halo_exchanges"] > 0 and two["sharded"]["verification_failures"] == 0
    # (the shifted, rank-deficient operator: CG iteration counts are not reproducible between summation orders - DESIGN.md 4 - and
    # the stopping test runs every 5th iteration)
    assert abs(one["config"]["last_cg_iterations_fwd"] - two["config"]["last_cg_iterations_fwd"]) <= 30


@pytest.mark.eight_ranks
def test_config5_4096_eight_slabs():
    """BASELINE.json config 5 at its size and rank count: decaying turbulence 4096^2, 8-slab decomposition, mailbox halo exchange and
    all-reduced dot products - eight processes (here: sharing the one GPU, so the CG runs its two-kernel iteration; eight
    persistent slab kernels of 128 workgroups cannot be resident side by side on 256 CUs), one PISO step forward + reverse sweep
    with a bounded iteration count, against the same step on one GPU."""
    common = ["--steps", "1", "--warmup", "0", "--grid", "4096", "--no-cpu-baseline", "--no-extras", "--max-iterations", "100"]
    one = _bench({}, ["--gpus", "1"] + common, 1)
    eight = _bench({"PISO_BENCH_SHARE_GPU": "1", "PISO_BENCH_SLAB_CHECK": "0"}, ["--gpus", "8", "--decomp", "slab"] + common, 8)
    assert eight["n_gpus"] == 8 and eight["scaling"] == "strong"
    l1, l8 = one["config"]["loss"], eight["config"]["loss"]
    print(one["config"], eight["config"])
    assert abs(l1 - l8) <= 1e-5 * abs(l1), (l1, l8)
    assert one["config"]["last_bicgstab_iterations"] == eight["config"]["last_bicgstab_iterations"]
    assert eight["config"]["warn"] == 0.0


def test_bench_line_contract_on_one_gpu():
    """`python bench.py` prints ONE JSON line with the keys the driver reads, the roofline block and the CPU baseline (small grid
    here: the shape of the line is what is tested, the numbers are the default run's business)."""
    d = _bench({}, ["--gpus", "1", "--steps", "1", "--warmup", "1", "--grid", "256", "--no-extras"], 1)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "phases"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "steps/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # `bound` names what the kernel's own SQ counters show (the round-5 verdict: "hbm" was a mislabel for a kernel whose vectors never
    # leave the chip); achieved / peak / frac stay the contract's HBM figure and the contract's enum value is kept beside it
    assert r["bound"] == "valu+latency" and r["bound_enum_of_the_contract"].startswith("hbm")
    assert r["unit"] == "GB/s" and 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert d["phases"]["verification_failures"] == 0 and d["phases"]["persistent_cg_fallbacks"] == 0


def test_weak_scaled_decomposed_step_two_ranks_matches_one_gpu():
    """`--decomp slab-weak`: ONE 256 x 512 periodic box, one 256^2 slab per rank, both linear solvers decomposed - against the
    same box on one GPU (`--grid-ny 512`): the loss must agree (rectangular grids, slabs of a taller box)."""
    common = ["--steps", "1", "--warmup", "0", "--grid", "256", "--no-cpu-baseline", "--no-extras", "--tol", "1e-7"]
    one = _bench({}, ["--gpus", "1", "--grid-ny", "512"] + common, 1)
    two = _bench({"PISO_BENCH_SHARE_GPU": "1", "PISO_BENCH_SLAB_CHECK": "0"}, ["--gpus", "2", "--decomp", "slab-weak"] + common, 2)
    assert two["scaling"] == "weak" and two["n_gpus"] == 2 and two["config"]["grid"] == [512, 256] == one["config"]["grid"]
    l1, l2 = one["config"]["loss"], two["config"]["loss"]
    print(one["config"], two["config"])
    assert abs(l1 - l2) <= 1e-5 * abs(l1), (l1, l2)
    assert one["config"]["last_bicgstab_iterations"] == two["config"]["last_bicgstab_iterations"]
    assert abs(two["value"] - 2 * 1e3 / two["ms_per_step"]) <= 1e-6 * two["value"]      # a step of the box counts as two steps at 256^2


def test_sharded_step_four_ranks_weak_box_matches_one_gpu():
    """Four ranks, ONE 128 x 512 periodic box (a 128^2 slab each), two unrolled steps forward + reverse sweep with every kernel of
    the step on the rank's rows: loss and |dL/du_0| against the same box on one GPU."""
    common = ["--steps", "2", "--warmup", "0", "--grid", "128", "--no-cpu-baseline", "--no-extras", "--tol", "1e-7"]
    one = _bench({}, ["--gpus", "1", "--grid-ny", "512"] + common, 1)
    four = _bench({"PISO_BENCH_SHARE_GPU": "1", "PISO_BENCH_SLAB_CHECK": "0"}, ["--gpus", "4", "--decomp", "slab-weak"] + common, 4)
    print(one["config"], four["config"], four["sharded"])
    assert four["config"]["grid"] == [512, 128] == one["config"]["grid"] and four["sharded"]["ranks_seen"] == 4
    assert abs(one["config"]["loss"] - four["config"]["loss"]) <= 1e-5 * abs(one["config"]["loss"])
    assert abs(one["config"]["grad_norm"] - four["config"]["grad_norm"]) <= 1e-5 * one["config"]["grad_norm"]
    assert one["config"]["last_bicgstab_iterations"] == four["config"]["last_bicgstab_iterations"]
    assert four["sharded"]["verification_failures"] == 0 and four["config"]["warn"] == 0.0
