"""Where the GPU idles: gaps between consecutive kernels of a rocprofv3 kernel trace, summed by (kernel before, kernel after).
Usage: trace_gaps.py <trace dir> [steps]"""
import csv, glob, os, re, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Name")))
rows.sort()
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
short = lambda n: re.sub(r"\(.*$", "", re.sub(r"^void ", "", n))[:60]
gaps, busy, end = {}, 0, rows[0][1]
for i, (s, e, n) in enumerate(rows):
    busy += e - s
    if i:
        g = s - end
        if g > 0:
            k = (short(rows[i - 1][2]), short(n))
            v = gaps.setdefault(k, [0, 0]); v[0] += g; v[1] += 1
        end = max(end, e)
total = rows[-1][1] - rows[0][0]
print("span %.2f ms, busy %.2f ms, idle %.2f ms per step (%g steps)" % (1e-6 * total / steps, 1e-6 * busy / steps, 1e-6 * (total - busy) / steps, steps))
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print("%8.1f us per step in %5.1f gaps per step  | %-60s -> %s" % (1e-3 * v[0] / steps, v[1] / steps, k[0], k[1]))
