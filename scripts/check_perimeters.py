"""Host-side look into the CG workspace after a failing single-exchange run: no kernel changes, so the failure reproduces."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import torch
import diffpiso._native as N
from diffpiso.solvers import cg_solve_native
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from tests.cases import pressure_system as case
nx = ny = 2048
n = nx * ny
L, b = case(nx, ny)
def arrays():
    ws = N._workspaces[(str(b.device), "cg")]
    base = 56 * n + 256
    names = ["r", "z", "p0", "p1", "zp0", "zp1"]
    return {nm: ws[base + i * 8 * n: base + (i + 1) * 8 * n].view(torch.float64).view(ny, nx).clone() for i, nm in enumerate(names)}
jj = torch.arange(ny, device=b.device).view(-1, 1) % 16
ii = torch.arange(nx, device=b.device).view(1, -1) % 128
for nit in (3,):
    N.set_option("cg_persist", 0)
    xa, _ = cg_solve_native(nx, ny, True, True, L, b, 1e-30, nit, False, 1000)
    ref = arrays()
    N.set_option("cg_persist", 1); N.set_option("cg_persist_r", 16)
    for rep in range(3):
        xb, _ = cg_solve_native(nx, ny, True, True, L, b, 1e-30, nit, False, 1000)
        got = arrays()
        print("nit", nit, "rep", rep, "x err %.2e" % float((xa - xb).abs().max() / xa.abs().max()))
        for nm in ("r", "p0", "p1"):
            for rn in ("r", "p0", "p1"):
                d = (got[nm] - ref[rn]).abs() / ref[rn].abs().max()
                if float(d.max()) < 1e-6:
                    bad = d > 1e-14
                    print("   %s vs ref %s: max %.2e, cells > 1e-14: %d" % (nm, rn, float(d.max()), int(bad.sum())))
        zref = ref["z"]; scale = zref.abs().max()
        d = (got["zp0"] - zref).abs() / scale
        classes = {"bottom row (ring row below)": (jj == 0) & (ii >= 0), "top row (ring row above)": (jj == 15) & (ii >= 0),
                   "left column (edge)": (ii == 0) & (jj > 0) & (jj < 15), "right column (edge)": (ii == 127) & (jj > 0) & (jj < 15)}
        for nm, m in classes.items():
            dm = torch.where(m, d, torch.zeros_like(d))
            bad = dm > 1e-13
            print("   z'_2 perimeter, %s: max %.2e, cells > 1e-13: %d of %d" % (nm, float(dm.max()), int(bad.sum()), int(m.sum())))
            if int(bad.sum()):
                idx = bad.nonzero()[:8]
                print("      e.g. (row, col, got, ref):", [(int(a), int(c), float(got["zp0"][a, c]), float(zref[a, c])) for a, c in idx])
                rows = torch.unique(idx[:, 0] // 16)[:8]; cols = torch.unique(bad.nonzero()[:, 1] // 128)[:16]
                print("      region rows:", rows.tolist(), "strips:", cols.tolist(), " rows hit:", int(torch.unique(bad.nonzero()[:, 0]).numel()))
