"""Build libpiso_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  No torch dependency: plain C ABI."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "diffpiso", "libpiso_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.isfile(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


LAST_MODE = None      # "compiled" / "reused": what the last build() call did (recorded by __graft_entry__.build())


def build(force=False, verbose=False):
    global LAST_MODE
    if not force and not needs_build():
        LAST_MODE = "reused"
        return OUT
    LAST_MODE = "compiled"
    objs = []
    os.makedirs(os.path.join(CSRC, "_obj"), exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(CSRC, "_obj", os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", 
               "-ffp-contract=off", "-c", src, "-o", obj] + os.environ.get("PISO_HIPCC_FLAGS", "").split()
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
