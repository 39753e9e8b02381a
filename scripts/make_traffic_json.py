"""scripts/summarize_pmc.py output -> profiles/traffic.json: fabric bytes per launch of the CG kernels from the FETCH_SIZE /
WRITE_SIZE passes, corrected with the factors measured on scripts/pmc_calib.hip in the same session (known 256 MiB per launch
for every access pattern), stamped with the sha of the kernel sources (bench.py only uses a record whose sha matches)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

txt = open(sys.argv[1]).read()
tag = sys.argv[2] if len(sys.argv) > 2 else "rXX"
rows = {}
for m in re.finditer(r"^(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches=\s*(\d+) mean=([0-9.e+]+)", txt, re.M):
    rows.setdefault(m.group(1).strip(), {})[m.group(2)] = (float(m.group(4)), int(m.group(3)))
CAL_BYTES = 256 << 20


def find(sub):
    for k, v in rows.items():
        if sub in k:
            return v
    return {}


def factor(kernel, counter):
    v = find(kernel).get(counter)
    return (CAL_BYTES / (v[0] * 1024.0)) if v and v[0] > 0 else None


cal = {"read_8B_per_lane": factor("read8<0>", "FETCH_SIZE") or factor("read8ILi0", "FETCH_SIZE"),
       "read_8B_per_lane_sc1": factor("read8<16>", "FETCH_SIZE") or factor("read8ILi16", "FETCH_SIZE"),
       "read_16B_per_lane": factor("read16", "FETCH_SIZE"),
       "write_8B_per_lane_sc1": factor("write8a", "WRITE_SIZE"), "write_16B_per_lane": factor("write16", "WRITE_SIZE")}
out = {"_comment": "fabric bytes = counter [KB] x 1024 x calibration factor of the matching access pattern (scripts/pmc_calib.hip, "
                   "same session); bench.py reads this file and uses a record only if kernel_source_sha matches its sources",
       "calibration_factors": cal, "2048": {}}
sha = bench.kernel_source_sha()
for name, key in (("cg_persist1", "cg_persist"), ("cg_persist<", "cg_persist"), ("cg_persistI", "cg_persist"), ("cg_k1", "cg_k1"), ("cg_k2", "cg_k2")):
    v = find(name)
    if not v or key in out["2048"]:
        continue
    f_kb = v.get("FETCH_SIZE", (0, 0))[0]
    w_kb = v.get("WRITE_SIZE", (0, 0))[0]
    # cg_persist: reads are 8 B per lane (float2 coefficient rows, sc1 perimeters), writes 8 / 16 B per lane sc1
    rf = cal["read_8B_per_lane"] if key == "cg_persist" else cal["read_16B_per_lane"]
    wf = cal["write_8B_per_lane_sc1"] if key == "cg_persist" else cal["write_16B_per_lane"]
    rec = {"kernel": name, "fetch_kb_per_launch": f_kb, "write_kb_per_launch": w_kb, "read_factor": rf, "write_factor": wf,
           "bytes": f_kb * 1024 * (rf or 1.0) + w_kb * 1024 * (wf or 1.0), "kernel_source_sha": sha,
           "source": "profiles/%s_bench2048_pmc_fetch_write_summary.txt" % tag}
    if key == "cg_persist":
        rec["iterations_per_launch"] = 149          # PMC passes run with --max-iterations 150: one segment of 149 iterations
        rec["bytes_per_iteration"] = rec["bytes"] / 149.0
    out["2048"][key] = rec
# BiCGStab (optional third argument: the PMC summary of scripts/bench_bicg.py): all bi_* kernels of the run, per solve.  The run makes
# four fixed-work solves (PISO_BICG_PROFILE=1: nothing else)
if len(sys.argv) > 3:
    btxt = open(sys.argv[3]).read()
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    for m in re.finditer(r"^(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches=\s*(\d+) mean=([0-9.e+]+) sum=([0-9.e+]+)", btxt, re.M):
        if "bi_" in m.group(1):
            tot[m.group(2)] += float(m.group(5))
    # how many solves the profiled run made and how long each was: printed by scripts/bench_bicg.py itself (the JSON line in the run log,
    # argument 4); the PMC passes must run under PISO_BICG_PROFILE=1 (fixed-work solves only: real_solves_run == 0)
    fixed_its, fixed_solves = 20, 4
    if len(sys.argv) > 4:
        line = [l for l in open(sys.argv[4]).read().splitlines() if l.startswith("{") and "fixed_work_solves_run" in l]
        rec = json.loads(line[-1])
        if rec.get("real_solves_run", 0) != 0:
            raise SystemExit("the BiCGStab PMC pass also ran solves to tolerance: run it under PISO_BICG_PROFILE=1")
        fixed_its, fixed_solves = int(rec["iterations_per_fixed_solve"]), int(rec["fixed_work_solves_run"])
    share = 1.0
    rb = tot["FETCH_SIZE"] * 1024 * (cal["read_16B_per_lane"] or 1.0) / fixed_solves
    wb = tot["WRITE_SIZE"] * 1024 * (cal["write_16B_per_lane"] or 1.0) / fixed_solves
    out["2048"]["bicgstab"] = {"kernel": "bi_* (whole fixed-work solve)", "fetch_kb_all_solves": tot["FETCH_SIZE"], "write_kb_all_solves": tot["WRITE_SIZE"],
                               "share_of_the_fixed_work_solves": share, "iterations_per_solve": fixed_its, "bytes_per_solve": rb + wb,
                               "kernel_source_sha": sha, "source": "profiles/%s_bicgstab2048_pmc_fetch_write_summary.txt" % tag}
print(json.dumps(out, indent=1))
