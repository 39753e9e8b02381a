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
json_out = None
if "--json" in sys.argv:
    k = sys.argv.index("--json")
    json_out = sys.argv[k + 1]
    del sys.argv[k:k + 2]
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
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
    print("L2 (TCC) hit rate: %.3f   (TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum); requests per iteration %.0f)" % (
        g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), (g("TCC_HIT_sum") + g("TCC_MISS_sum")) / ITERS))
if g("TCC_EA0_RDREQ_sum") and g("TCC_EA0_RDREQ_DRAM_sum") is not None:
    # (TCC_EA0_RDREQ_DRAM counts requests ADDRESSED to device memory: the Infinity Cache sits in front of HBM and is not told apart)
    print("L2 read requests to the fabric per iteration: %.0f, of which addressed to device memory %.3f" % (
        g("TCC_EA0_RDREQ_sum") / ITERS, g("TCC_EA0_RDREQ_DRAM_sum") / g("TCC_EA0_RDREQ_sum")))
if g("SQ_WAVE_CYCLES") and g("SQ_WAVES"):
    # quad-cycles per wave and iteration -> shader cycles per iteration (a wave lives for the whole launch)
    print("shader cycles per iteration (4 x SQ_WAVE_CYCLES / waves / iterations): %.0f" % (4 * g("SQ_WAVE_CYCLES") / g("SQ_WAVES") / ITERS))

if json_out and g("SQ_WAVE_CYCLES") and g("SQ_WAVES"):
    import json
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    wc, w = g("SQ_WAVE_CYCLES"), g("SQ_WAVES")
    rec = {"grid": n, "kernel": names.most_common(1)[0][0][:100], "kernel_source_sha": bench.kernel_source_sha(), "iterations_per_launch": ITERS,
           "waves": w, "waves_per_simd": 2,
           "wave_cycle_fractions": {k: (g(v) / wc if g(v) is not None else None) for k, v in (
               ("valu_issue", "SQ_ACTIVE_INST_VALU"), ("any_issue", "SQ_ACTIVE_INST_ANY"), ("parked_waitcnt_barrier_sleep", "SQ_WAIT_ANY"),
               ("issue_stall", "SQ_WAIT_INST_ANY"), ("lds_issue", "SQ_ACTIVE_INST_LDS"), ("scalar_issue", "SQ_ACTIVE_INST_SCA"),
               ("vmem_issue", "SQ_ACTIVE_INST_VMEM"))},
           "valu_pipe_busy": 2 * g("SQ_ACTIVE_INST_VALU") / wc if g("SQ_ACTIVE_INST_VALU") else None,
           "per_wave_and_iteration": {k: (g(k) / w / ITERS if g(k) is not None else None) for k in (
               "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")},
           "valu_cycles_per_instruction": 4 * g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU") if g("SQ_INSTS_VALU") else None,
           "shader_cycles_per_iteration": 4 * wc / w / ITERS,
           "l2_hit_rate": (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))) if (g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0) else None,
           "l2_read_requests_to_dram_share": (g("TCC_EA0_RDREQ_DRAM_sum") / g("TCC_EA0_RDREQ_sum")) if g("TCC_EA0_RDREQ_sum") else None,
           "lds_bank_conflict_share": (g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")) if g("SQ_LDS_IDX_ACTIVE") else None,
           "source": "scripts/profile_sq.sh: rocprofv3 --kernel-trace --pmc <one SQ group per pass> -- python3 scripts/bench_cg.py %d" % n,
           "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; fractions are of a wave's life, "
                    "valu_pipe_busy = 2 waves per SIMD x valu_issue"}
    json.dump({str(n): rec}, open(json_out, "w"), indent=1)          # profiles/sq_counters.json (bench.py: roofline.sq_counters)
