"""Lid-driven cavity Re 1000: the centre-line extrema against time (is t = 25 of the reference's script steady?) and against dt / n.
    python scripts/ldc_convergence.py [n] [dt] [t_end] [wall_exact 0|1]"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ldc_example", os.path.join(ROOT, "examples", "lid_driven_cavity_2d.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dt = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
t_end = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
wall_exact = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False


def monitor(i, velocity):
    u, v = mod.centre_lines(velocity, n)
    t = velocity.staggered_tensor()[0]
    print("t %6.1f  u_min %.5f  v_min %.5f  v_max %.5f   max|u-Ghia| %.5f  max|v-Ghia| %.5f" % (
        i * dt, u.min(), v.min(), v.max(), np.abs(u - mod.GHIA_RE1000_U).max(), np.abs(v - mod.GHIA_RE1000_V).max()), flush=True)


vel, _ = mod.run(n=n, reynolds=1000, dt=dt, steps=int(round(t_end / dt)), verbose=False, save_every=int(round(5.0 / dt)), monitor=monitor, wall_exact=wall_exact)
print(mod.ghia_report(vel, n)[0])
