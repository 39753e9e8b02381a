"""Build the library from the CURRENT sources into scripts/_bin/lib<name>.so (A/B runs through PISO_HIP_LIB; the product
library is untouched).  Usage: python scripts/build_variant.py <name> [extra hipcc flags ...]   e.g. ... diag -DPISO_PERSIST_DIAG"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "differentiable-piso_amd", "csrc")
name, flags = sys.argv[1], sys.argv[2:]
objdir = os.path.join(ROOT, "scripts", "_bin", "_obj_" + name)
os.makedirs(objdir, exist_ok=True)
procs, objs = [], []
for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
    obj = os.path.join(objdir, os.path.basename(src) + ".o")
    objs.append(obj)
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                                   "-Wno-int-to-pointer-cast", "-c", src, "-o", obj] + flags, stderr=subprocess.DEVNULL))
assert all(p.wait() == 0 for p in procs)
out = os.path.join(ROOT, "scripts", "_bin", "lib%s.so" % name)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
subprocess.call(["rm", "-rf", objdir])
print(out)
