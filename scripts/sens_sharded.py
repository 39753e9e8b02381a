"""How sensitive are the fields of a step with UNCONVERGED pressure solves (K fixed un-shifted CG iterations) to round-off-sized changes?
One-GPU run vs (a) the same with 1e-7 relative white noise on the initial velocity, (b) the sharded run.  rel-L2 of u, p, dL/du_0, dL/dp_0."""
import sys, os, tempfile, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import test_gpu_sharded_fields as T
for n, w, its in ((1024, 2, 100), (1024, 2, 2000)):
    common = ["--steps", "1", "--warmup", "0", "--grid", str(n), "--no-cpu-baseline", "--no-extras", "--max-iterations", str(its), "--unshifted", "--tol", "1e-30", "--lin-tol", "1e-9"]
    d1, d2, d3 = tempfile.mkdtemp(), tempfile.mkdtemp(), tempfile.mkdtemp()
    T._bench_dump({}, ["--gpus", "1", "--cg-persist", "0"] + common, 1, d1)
    T._bench_dump({}, ["--gpus", "1", "--cg-persist", "0", "--perturb-input", "1e-7"] + common, 1, d3)
    T._bench_dump({"PISO_BENCH_SHARE_GPU": "1", "PISO_BENCH_SLAB_CHECK": "0"}, ["--gpus", str(w), "--decomp", "slab"] + common, w, d2)
    a = T._gather(d1, 1, n, n); b = T._gather(d2, w, n, n); c = T._gather(d3, 1, n, n)
    print(n, w, its, "sharded vs one GPU", ["%.2e" % T._rel(b[k], a[k]) for k in range(4)], " perturbed one GPU vs one GPU", ["%.2e" % T._rel(c[k], a[k]) for k in range(4)], flush=True)
