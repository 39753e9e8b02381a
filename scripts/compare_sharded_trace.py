"""Per kernel: launches and workgroups per launch in two rocprofv3 kernel traces (one process stepping the whole box / rank 1 of the
sharded run on its slab).  Usage: compare_sharded_trace.py <dir of the one-GPU trace> <dir of rank 1's trace>"""
import csv, glob, os, re, sys


def load(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    out = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name") or row.get("Name")
            gx = int(row.get("Grid_Size_X") or row.get("Grid_Size") or 0)
            gy, gz = int(row.get("Grid_Size_Y") or 1), int(row.get("Grid_Size_Z") or 1)
            wx = int(row.get("Workgroup_Size_X") or row.get("Workgroup_Size") or 1)
            wy, wz = int(row.get("Workgroup_Size_Y") or 1), int(row.get("Workgroup_Size_Z") or 1)
            wgs = (gx * gy * gz) // max(wx * wy * wz, 1)
            dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            e = out.setdefault(name, [0, 0, 0])
            e[0] += 1; e[1] += wgs; e[2] += dur
    return out


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:78]


one, r1 = load(sys.argv[1]), load(sys.argv[2])
print("""How to read this table
* Round 5: a rank's tensors hold its STORED rows only (owned rows + halo rows, piso_slab_t in include/piso_hip.h).
* Kernels whose grid follows the problem size (bi_sweep, bi_spmv, bi_factor, csr_matvec, cg_k2, cg_setup_coeffs, cg_init, ...) launch
  HALF the workgroups on rank 1.
* Kernels with a FIXED grid (bi_update_* / bi_residual_init / bi_convert 2048, cg_k1 1024, laplace_kernel 4096, face / divergence /
  gradient kernels 2048: grid-stride loops) show the same workgroup count on both sides; their loops run over the rank's owned rows
  (the *_slab entry points) - compare their time per launch, with a grain of salt: the two ranks of this run SHARE one GPU, so
  rank 1's kernels compete with rank 0's.
* The at::native kernels are torch's element-wise copies / adds / fills between the library's kernels: on tensors of the rank's
  stored rows their workgroup counts are ~(ny / N + 4 ... 6) / ny of the one-GPU run's (0.33 - 0.56 here; rounds 3 - 4, with globally
  indexed arrays: 1.00).
* peer_* / slab_collapse / bi_flags_allreduce: the mailbox traffic of the sharded run (halo rows, all-reduced dot products); their
  time per launch is mostly waiting for the other rank, which here runs on the same GPU.
* The CG of this shared-GPU run iterates on the two-kernel path (two persistent kernels cannot be resident side by side on one GPU);
  one rank per GPU runs cg_persist1<..., SLAB>.
""")
print("kernel | one GPU, 1024 x 2048 box: launches, workgroups per launch, us per launch | rank 1 of 2, its 1024 x 1024 slab: the same | workgroup ratio")
names = sorted(set(one) | set(r1), key=lambda n: -(one.get(n, [0, 0, 0])[2] + r1.get(n, [0, 0, 0])[2]))
tot = [0, 0]
for n in names:
    a, b = one.get(n), r1.get(n)
    fa = "%6d %9.1f %9.1f" % (a[0], a[1] / a[0], 1e-3 * a[2] / a[0]) if a else "     -         -         -"
    fb = "%6d %9.1f %9.1f" % (b[0], b[1] / b[0], 1e-3 * b[2] / b[0]) if b else "     -         -         -"
    ratio = "%.2f" % ((b[1] / b[0]) / (a[1] / a[0])) if a and b and a[1] else "-"
    if a: tot[0] += a[2]
    if b: tot[1] += b[2]
    print("%-78s | %s | %s | %s" % (short(n), fa, fb, ratio))
print("GPU time of all kernels: one GPU %.1f ms, rank 1 %.1f ms" % (1e-6 * tot[0], 1e-6 * tot[1]))
