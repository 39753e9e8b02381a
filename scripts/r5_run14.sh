#!/bin/bash
cd $GRAFT_REPO_ROOT
cat > /tmp/chk.py <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "differentiable-piso_amd"); sys.path.insert(0, "tests")
import torch
import diffpiso._native as N
from diffpiso.solvers import cg_solve_native
sys.argv = ["x", "none"]
import importlib.util
spec = importlib.util.spec_from_file_location("ab", "scripts/r5_ab.py"); ab = importlib.util.module_from_spec(spec); spec.loader.exec_module(ab)
for nx, ny, kind in ((1024, 256, "open"), (512, 512, "periodic"), (256, 256, "periodic"), (512, 256, "walls_y")):
    L, b, px, py = ab.system(nx, ny, kind)
    out = {}
    for nq in (0, 1):
        N.set_option("cg_persist_nq", nq)
        x, its = cg_solve_native(nx, ny, px, py, L, b, 1e-30, 300, kind != "open", 1 << 30)
        out[nq] = x.clone()
    N.set_option("cg_persist", 0)
    x2, _ = cg_solve_native(nx, ny, px, py, L, b, 1e-30, 300, kind != "open", 1 << 30)
    N.set_option("cg_persist", -1)
    d = lambda a, b: float((a - b).abs().max() / b.abs().max())
    print(nx, ny, kind, "nq1 vs nq2 %.2e, nq1 vs two-kernel %.2e, nq2 vs two-kernel %.2e" % (d(out[1], out[0]), d(out[1], x2), d(out[0], x2)), flush=True)
PY
timeout 300 python /tmp/chk.py 2>&1 | grep -v "^$" | tail -6
for nq in 0 1; do for half in -1 0; do
echo "--- nq $nq half $half"
PISO_CG_PERSIST_NQ=$nq PISO_CG_PERSIST_HALF=$half timeout 300 python scripts/r5_ab.py small 2>/dev/null | grep "grid"
done; done
