"""Condenses the per-workgroup phase clocks of a diagnostic build (PISO_CG_PERSIST_TIMING=2, scripts/run_diag_wg.sh) by XCD and by band.
Usage: python scripts/wg_table.py gpurun_out/r4_wg_<lib>.log ..."""
import re, sys, collections, statistics as st
for f in sys.argv[1:]:
    rows = []
    for l in open(f):
        m = re.match(r"cg_persist_wg\s+(\d+) xcd (\d+) band\s+(\d+)\s+D ([\d.]+)\s+exchange ([\d.]+)\s+U ([\d.]+)\s+\| drain ([\d.]+)\s+barrier1 ([\d.]+)\s+publish\+poll ([\d.]+)\s+sums ([\d.]+)", l)
        if m:
            rows.append([float(x) for x in m.groups()])
    rows = rows[-256:]
    print(f, len(rows), [l for l in open(f) if l.startswith("grid")][-1].strip())
    byx = collections.defaultdict(list)
    for r in rows:
        byx[int(r[1])].append(r)
    for x in sorted(byx):
        rs = byx[x]
        print("  xcd %d n=%d  D %.2f  U %.2f (min %.2f max %.2f)  barrier1 %.2f  pub+poll %.2f  bands %d..%d" % (
            x, len(rs), st.mean(r[3] for r in rs), st.mean(r[5] for r in rs), min(r[5] for r in rs), max(r[5] for r in rs),
            st.mean(r[7] for r in rs), st.mean(r[8] for r in rs), min(r[2] for r in rs), max(r[2] for r in rs)))
    rs = sorted(rows, key=lambda r: r[2])
    print("  U by band:", " ".join("%.1f" % r[5] for r in rs))
