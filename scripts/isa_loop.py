"""Static look at a kernel's compiled loop (no GPU needed): registers, spills, scratch and the instruction mix of its largest loop.
Usage: python scripts/isa_loop.py <csrc file, e.g. cg_slab.hip> <substring of the mangled kernel name> [extra hipcc flags ...]
e.g.   python scripts/isa_loop.py cg_slab.hip cg_persist1IdfLi16ELi1ELb1ELb1ELb1
The iteration loop of the persistent kernels is bound by VALU issue: every v_* instruction (v_readlane reloads of spilled SGPRs
included) costs the same ~4.5 SIMD cycles, so `valu`, `readlane`, `scratch` of the loop are the numbers to watch."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "differentiable-piso_amd", "csrc")


def compile_to_asm(src, flags):
    d = tempfile.mkdtemp(prefix="isa_")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-int-to-pointer-cast",
           "-c", os.path.join(CSRC, src), "-o", os.path.join(d, "o.o"), "-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"] + flags
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-4000:])
        raise SystemExit(1)
    asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")]
    return os.path.join(d, asm[0]), p.stderr


def resources(remarks, key):
    out, cur = {}, None
    for l in remarks.split("\n"):
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            cur = m.group(1)
            continue
        m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?): (\d+)", l) or re.search(r"\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", l)
        if m and cur and key in cur:
            out.setdefault(cur, {})[m.group(1).strip()] = int(m.group(2))
    return out


def loops(lines):
    """lines of the largest loop: the blocks the assembler's comments attribute to one loop header ("in Loop: Header=BBn_m" /
    "Parent Loop BBn_m"), header block included - block placement may put unrelated blocks between them."""
    blocks, cur, name = {}, None, None
    for n, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            name, cur = m.group(1)[2:], None
            hdr = None
            for k in range(n, min(n + 4, len(lines))):
                mm = re.search(r"(?:in Loop: Header=|Parent Loop |Loop Header: Depth=1)(BB\d+_\d+)?", lines[k])
                if mm:
                    hdr = mm.group(1) or name
                    break
                if k > n and not lines[k].lstrip().startswith(";"):
                    break
            cur = hdr
        elif re.match(r"^; %bb\.\d+:", l.strip()):
            mm = re.search(r"in Loop: Header=(BB\d+_\d+)", " ".join(lines[n:n + 2]))
            cur = mm.group(1) if mm else None
        if cur:
            blocks.setdefault(cur, []).append(l)
    if not blocks:
        return None
    return max(blocks.values(), key=len)


def mix(body):
    c = collections.Counter()
    for l in body:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        c[l.split()[0]] += 1
    return c


def main():
    src, key, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    asm, remarks = compile_to_asm(src, flags)
    res = resources(remarks, key)
    text = open(asm).read()
    for name in sorted(res):
        i = text.index("\n" + name + ":")
        j = text.index(".Lfunc_end", i)
        lines = text[i:j].split("\n")
        r = res[name]
        print(name[:110])
        print("   VGPRs %s  SGPR spills %s  VGPR spills %s  scratch %s B/lane  LDS %s B" % (
            r.get("VGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]")))
        lp = loops(lines)
        if lp:
            c = mix(lp)
            valu = sum(v for k, v in c.items() if k.startswith("v_"))
            print("   largest loop: %d instructions, valu %d, readlane %d, writelane %d, scratch %d, fp64 %d, cvt %d, dpp %d, s_load %d, vmem %d, ds %d" % (
                sum(c.values()), valu, c["v_readlane_b32"], c["v_writelane_b32"], sum(v for k, v in c.items() if k.startswith("scratch_")),
                sum(v for k, v in c.items() if k.endswith("_f64") or k.endswith("_f64_e32") or k.endswith("_f64_e64")),
                sum(v for k, v in c.items() if k.startswith("v_cvt")), c["v_mov_b32_dpp"], sum(v for k, v in c.items() if k.startswith("s_load")),
                sum(v for k, v in c.items() if k.startswith("buffer_") or k.startswith("global_")), sum(v for k, v in c.items() if k.startswith("ds_"))))
            if os.environ.get("ISA_HIST"):
                print("   opcodes:", ", ".join("%s %d" % kv for kv in c.most_common(60)))
    print("asm:", asm)


if __name__ == "__main__":
    main()
