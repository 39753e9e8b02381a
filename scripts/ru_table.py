"""Resource-usage table of a hipcc -Rpass-analysis=kernel-resource-usage log: python scripts/ru_table.py LOG [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
cur = None
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("sgpr", r"SGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("ssp", r"SGPRs Spill: (\d+)"), ("vsp", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"^void piso::", "", n)
    n = re.sub(r"\(.*$", "", n)
    if flt and flt not in n:
        continue
    print("%-90s sgpr %3d spill %3d | vgpr %3d agpr %3d spill %3d scratch %4d | lds %6d occ %d" % (n[:90], r.get("sgpr", -1), r.get("ssp", -1), r.get("vgpr", -1), r.get("agpr", -1), r.get("vsp", -1), r.get("scratch", -1), r.get("lds", -1), r.get("occ", -1)))
