"""Condense the SQ / GRBM counter passes of scripts/profile_sq.sh: per counter the mean over the persistent CG kernel's dispatches (the
timed launches of scripts/bench_cg.py: 300 iterations each), then the fractions a reader needs: how much of a wave's life is VALU issue,
parked (s_waitcnt / barrier / s_sleep), issue-stalled, and the VALU instructions per wave and iteration against the ISA count.
Units (MI355X guide): SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_INSTS_* count wave instructions."""
import collections
import csv
import glob
import os
import sys

n = int(sys.argv[1])
ITERS = 300                                        # scripts/bench_cg.py: iterations per timed launch pair (warm-up + timed)
vals = collections.defaultdict(lambda: [0.0, 0])
names = collections.Counter()
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                kn = row.get("Kernel_Name", "?")
                if "cg_persist1" not in kn:
                    continue
                names[kn[:120]] += 1
                k = row.get("Counter_Name", "?")
                vals[k][0] += float(row.get("Counter_Value", 0) or 0)
                vals[k][1] += 1
print("persistent CG kernel, grid %d x %d, %d iterations per launch; kernel names seen:" % (n, n, ITERS))
for k, c in names.most_common(3):
    print("   %6d rows  %s" % (c, k))
mean = {k: s / c for k, (s, c) in vals.items() if c}
for k in sorted(mean):
    print("%-32s mean per dispatch %.6g   (dispatches %d)" % (k, mean[k], vals[k][1]))
g = mean.get
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    print()
    for label, key in (("VALU issue", "SQ_ACTIVE_INST_VALU"), ("any instruction issue", "SQ_ACTIVE_INST_ANY"), ("parked (waitcnt / barrier / sleep)", "SQ_WAIT_ANY"),
                       ("issue stall", "SQ_WAIT_INST_ANY"), ("LDS issue", "SQ_ACTIVE_INST_LDS"), ("LDS issue stall", "SQ_WAIT_INST_LDS"),
                       ("scalar issue", "SQ_ACTIVE_INST_SCA"), ("vector-memory issue", "SQ_ACTIVE_INST_VMEM")):
        if g(key) is not None:
            print("fraction of a wave's cycles in %-36s %.3f   (%s / SQ_WAVE_CYCLES)" % (label + ":", g(key) / wc, key))
if g("SQ_WAVES") and g("SQ_INSTS_VALU"):
    w = g("SQ_WAVES")
    print()
    print("waves per dispatch %.0f (= workgroups x 8)" % w)
    for key in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
        if g(key) is not None:
            print("%-18s per wave and iteration: %.1f" % (key, g(key) / w / ITERS))
if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
    print("LDS bank-conflict cycles / LDS active cycles: %.4f" % (g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
if g("SQ_WAVE_CYCLES") and g("SQ_WAVES"):
    # quad-cycles per wave and iteration -> shader cycles per iteration (a wave lives for the whole launch)
    print("shader cycles per iteration (4 x SQ_WAVE_CYCLES / waves / iterations): %.0f" % (4 * g("SQ_WAVE_CYCLES") / g("SQ_WAVES") / ITERS))
