"""profiles/traffic.json must have been measured on the kernel sources of this tree (scripts/profile_bench.sh); with a bench line
as argument: it must carry the counters' traffic, and its headline is printed."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
have, want = t["2048"]["cg_persist"]["kernel_source_sha"], bench.kernel_source_sha()
if have != want:
    sys.exit("profiles/traffic.json was taken on sources %s, this tree has %s" % (have, want))
if len(sys.argv) > 1:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1])
    r = d["roofline"]
    if r["traffic"] is None:
        sys.exit("the bench line carries no PMC traffic: %s" % r["frac_source"])
    print("%s: %.3f steps/s, %.2f us per CG iteration, roofline.frac %.3f (%s)" % (sys.argv[2] if len(sys.argv) > 2 else "", d["value"], r["us_per_iteration"], r["frac"], r["frac_source"]))
