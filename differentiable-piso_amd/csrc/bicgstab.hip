// ILU(0)-preconditioned BiCGStab for the two advection-diffusion matrices (u and v) of the PISO predictor on MI355X.
//
// Replaces MultiBicgstabIluLinearSolveLauncher / BicgstabIluLinearSolveLauncher[Double]
// (CUDAsrc/multi_bicgstab_ilu_linear_solve_op.cu.cc:85-531, :540-988) and the cuSPARSE / cuBLAS calls inside them
// (csr2csc, csrilu02, csrsv2 x4 per iteration, csrmv x2, dot/nrm2/axpy/scal with a host sync each).
//
// MI355X design (DESIGN.md "BiCGStab"):
//  * both components advance in the SAME launches (blockIdx.y = component) instead of two host threads + two streams;
//  * the CSR input is converted once per solve into a structured "stencil" layout (5 SoA coefficient arrays for the
//    geometric neighbours k-W, k-1, k, k+1, k+W; the few periodic-wrap entries go to a small per-frame-row exception
//    table).  The adjoint (transpose_op) is obtained by gathering the transposed coefficients during that conversion --
//    the role of cusparse?csr2csc;
//  * the preconditioner is a structured block ILU(0): ILU(0) of the near-neighbour part of the matrix restricted to bands
//    of `band_rows` face rows (band_rows < 0: one band = global ILU(0) of the near-neighbour part).  For this triangle-free
//    pattern ILU(0) only modifies the diagonal: d_k = a_kk - a_kW a_Wk / d_W - a_kS a_Sk / d_S.  Inside a band rows are
//    processed top to bottom; along a row the recurrences are solved by parallel scans across the block
//    (Moebius-map scan for the pivots, affine-map scan for the triangular sweeps), carries staged through LDS.
//    Every band is one workgroup: no inter-workgroup dependency, no level schedule;
//  * Krylov scalars live on the device; dot products are per-block partials reduced by 1-block "scalar" kernels.
// The iteration itself (order of updates, absolute ||r||_2 test after each half step, zero + one restart on failure,
// NaN -> warning) is the reference's.
#include "piso_common.h"
#include "options.h"
#include "slab_comm.h"

namespace piso {

constexpr int kBiParts = 1024;    // max blocks per component of a partial-producing kernel
constexpr int kExcSlots = 4;      // exception (wrap) entries per frame row

struct Geo {
  int nx, ny;
  int W[2], H[2], n[2], r0[2];    // face-array dims, rows, row offset of each component in the concatenated vectors
  int xw[2], yw[2];               // periodic wrap distances in x / y (skip the duplicate face in the own direction)
  int F[2], f0[2];                // frame rows (within 2 of a border) and their offset in the exception tables
  int R, nb[2];                   // band height (face rows) and number of bands
};

__host__ __device__ inline int frame_rows(int W, int H) {
  const int wi = W > 4 ? W - 4 : 0, hi = H > 4 ? H - 4 : 0;
  return W * H - wi * hi;
}
// ordinal of a frame row, -1 for interior rows
__device__ __forceinline__ int frame_ordinal(int i, int j, int W, int H) {
  if (H <= 4 || W <= 4) return j * W + i;
  if (j < 2) return j * W + i;
  if (j >= H - 2) return 2 * W + (j - (H - 2)) * W + i;
  if (i < 2) return 4 * W + (j - 2) * 4 + i;
  if (i >= W - 2) return 4 * W + (j - 2) * 4 + 2 + (i - (W - 2));
  return -1;
}

template <typename T>
struct alignas(2 * sizeof(T)) Pair { T a, b; };
template <typename T>
struct Tri { T d, a, b; };

template <typename T>
struct CompScalars {
  T rho, rho_prev, alpha, omega, beta, nrm;
  int done;        // 1: converged (||r|| < tol) -- no more work in this pass
  int it_count;
  int failed;      // set by the host-side restart logic
  int pad;
};

template <typename T>
struct BiArgs {
  Geo g;
  // matrix B (= A or A^T) in stencil form, concatenated components
  T *cS, *cW, *cC, *cE, *cN;
  int* ecol;
  T* eval;
  // preconditioner
  // (what a sweep needs of a row is ONE element: it reads its input and one coefficient stream and writes)
  Pair<T>* L;                     // {LW, LS}
  Tri<T>* U;                      // {1 / d, UE, UN}
  // vectors
  const T* rhs;
  T *x, *r, *rh, *p, *v, *t, *y, *ph, *sh;
  T* parts;                       // [2 comps][4 quantities][kBiParts]: where THIS launch writes its partial sums
  const T* parts_in;              // the partial sums the previous producer wrote (the other of the two buffers: a launch never
                                  // reads the buffer it writes, so its blocks may reduce `parts_in` while others already store)
  CompScalars<T>* sc;             // [2]: the scalars this launch works with
  const CompScalars<T>* sc_prev;  // fold != 0: the record BEFORE the folded stages (`sc` is then written by block 0, read by nobody)
  int fold;                       // scalar stages the blocks of this launch apply themselves before they start (see folded_scalars)
  int fuse_p;                     // forward sweep only: its input is the direction update p = r + beta (p - omega v) (:316-318), formed (and
                                  // stored to p) on the way in - bi_update_p's pass over three vectors and its launch are saved
  int* flags;                     // [0]: unsupported pattern, [1]: NaN seen
  float tol;
  int nparts;                     // blocks per component that write partial records (<= kBiParts; the rest stays zero)
  // slab decomposition (one GPU: everything): the rows [rb, re) and the bands [bb, be) of each component this rank works on.
  // All arrays stay globally indexed; rows outside the range are never written (the SpMV inputs receive their neighbours' edge
  // rows before every product), so slabs cut at band edges reproduce the single-GPU preconditioner exactly.
  int rb[2], re[2], bb[2], be[2];
  // where row `row` (component-local, whole-grid numbering) of component c lives in the concatenated vectors and coefficient arrays:
  // r0[c] + row when the arrays are the whole grid's (one GPU; slab solver on full arrays), the rank's stored rows otherwise
  // (slab-decomposed step, local storage: piso_common.h RowMap)
  RowMap M;
  __device__ __forceinline__ int kx(int c, int row) const { return M.on ? (c ? M.n_u : 0) + M.frow(c, row) : g.r0[c] + row; }
  __device__ __forceinline__ int rpx(int c, int row) const { return M.frow(c, row); }      // place of a row in the component's row pointers
};
// cross-rank part of a scalar kernel (peer transport): the two component blocks add their four partial sums over the ranks
struct BiPeer {
  PeerView pv;
  unsigned seq;
  int* err;
  int on;        // 0: one GPU.  1: peer transport - the sums cross the ranks inside the scalar kernel.  RCCL transport, two launches
                 // around an all-reduce: 2 = write the rank's sums to `gsum` and stop, 3 = continue from the all-reduced `gsum`
  double* gsum;  // [2 components][4 sums]
};

__device__ __forceinline__ bool is_nan(float v) { return v != v; }
__device__ __forceinline__ bool is_nan(double v) { return v != v; }

// ------------------------------------------------------------------------------------------------------------------
// CSR -> stencil conversion (with optional transpose) + exception table + NaN scan of values, rhs, x0
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ bool csr_find(const int* __restrict__ rp, const int* __restrict__ col,
                                         const T* __restrict__ val, int m, int k, T* out, const RowMap& M, int c) {
  const int ml = M.frow(c, m);                               // (row pointers follow the STORED rows; columns keep the whole grid's numbers)
  for (int q = rp[ml]; q < rp[ml + 1]; ++q)
    if (col[q] == k) { *out = val[q]; return true; }
  return false;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void bi_convert(BiArgs<T> a, const T* __restrict__ val_all,
                                                     const int* __restrict__ rp_all, const int* __restrict__ col_all,
                                                     const T* __restrict__ x0, int transpose_flags) {
  // transpose_flags: bit 0 = gather the transposed coefficients (adjoint solve), bit 1 = the matrix is -val (piso_tf.py:41 hands the
  // solver `-matrix_values`: negating here, in the one pass that reads the values anyway, saves the caller a pass over the array)
  const int transpose = transpose_flags & 1;
  const T sgn = (transpose_flags & 2) ? (T)-1 : (T)1;
  const int c = blockIdx.y;
  const Geo& g = a.g;
  const int W = g.W[c], H = g.H[c], n = g.n[c];
  const RowMap& M = a.M;
  const int nu_st = M.on ? M.n_u : g.n[0];                   // u rows the row pointers hold
  const int* rp = rp_all + (c ? nu_st + 1 : 0);
  // the nnz offset of component 1 is the last row pointer of component 0
  const int k0 = c ? rp_all[nu_st] : 0;
  const T* val = val_all + k0;
  const int* col = col_all + k0;
  bool nan_seen = false, bad = false;
  // One GPU: the CSR entries of the block's 256 rows are one run of the arrays.  They are fetched by element (coalesced, all loads of
  // the chunk in flight at once) and wait in LDS for the thread that owns their row: a thread walking its own row's entries makes one
  // dependent round trip to memory per entry.
  constexpr int kPer = 6;                                    // staged entries per thread (5 per row + slack)
  constexpr int kStage = kPer * kBlock;
  __shared__ int lcol[kStage];
  __shared__ T lval[kStage];
  const bool stage_ok = !M.on;
  for (int base = a.rb[c] + blockIdx.x * kBlock; base < a.re[c]; base += gridDim.x * kBlock) {
    const int row = base + (int)threadIdx.x;
    int q0 = 0;
    bool staged = false;
    if (stage_ok) {
      q0 = rp[base];
      const int q1 = rp[min(base + kBlock, a.re[c])];
      staged = q1 - q0 <= kStage;
      __syncthreads();                                       // (the block's previous chunk has been read)
      if (staged) {
        int tc[kPer];
        T tv[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
          const int q = q0 + u * kBlock + (int)threadIdx.x;
          tc[u] = q < q1 ? col[q] : 0;
          tv[u] = q < q1 ? val[q] : (T)0;
        }
#pragma unroll
        for (int u = 0; u < kPer; ++u) { lcol[u * kBlock + threadIdx.x] = tc[u]; lval[u * kBlock + threadIdx.x] = tv[u]; }
      }
      __syncthreads();
    }
    if (row >= a.re[c]) continue;
    const int i = row % W, j = row / W;
    const int fo = frame_ordinal(i, j, W, H);
    T s = 0, w = 0, cc = 0, e = 0, nn = 0;
    int ne = 0;
    int ec[kExcSlots];
    T ev[kExcSlots];
    // classify the entries of A's own row (also validates the pattern in transpose mode)
    const int rl = a.rpx(c, row);
    const int qb = rp[rl], qe = rp[rl + 1];
    auto classify = [&](int cq, T vq) __attribute__((always_inline)) {
      nan_seen |= is_nan(vq);
      int kind;   // 0..4 near slots, 5 exception
      if (cq == row) kind = 2;
      else if (cq == row - 1 && i >= 1) kind = 1;
      else if (cq == row + 1 && i <= W - 2) kind = 3;
      else if (cq == row - W) kind = 0;
      else if (cq == row + W) kind = 4;
      else kind = 5;
      if (kind == 5) {
        const int d = cq - row;
        const bool wrap = (d == g.xw[c] || d == -g.xw[c] || d == g.yw[c] || d == -g.yw[c]);
        if (!wrap || fo < 0) bad = true;
      }
      if (!transpose) {
        if (kind == 0) s = vq; else if (kind == 1) w = vq; else if (kind == 2) cc = vq;
        else if (kind == 3) e = vq; else if (kind == 4) nn = vq;
        else if (ne < kExcSlots) {
#pragma unroll
          for (int x = 0; x < kExcSlots; ++x) if (ne == x) { ec[x] = cq; ev[x] = vq; }
          ++ne;
        }
        else bad = true;
      } else if (kind == 2) cc = vq;
    };
    // (a row of these matrices has at most five entries: they are fetched together and classified without a loop; the loop behind
    // them only runs on a foreign pattern - which the flags above reject anyway)
    {
      int cq5[5];
      T vq5[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int q = qb + u < qe ? qb + u : qb;               // (a row has at least its diagonal)
        cq5[u] = staged ? lcol[q - q0] : col[q];
        vq5[u] = sgn * (staged ? lval[q - q0] : val[q]);
      }
#pragma unroll
      for (int u = 0; u < 5; ++u) if (qb + u < qe) classify(cq5[u], vq5[u]);
    }
    for (int q = qb + 5; q < qe; ++q) classify(staged ? lcol[q - q0] : col[q], sgn * (staged ? lval[q - q0] : val[q]));
    if (transpose) {
      T vq;
      if (i >= 1 && csr_find(rp, col, val, row - 1, row, &vq, M, c)) w = sgn * vq;            // A(k-1, k)
      if (i <= W - 2 && csr_find(rp, col, val, row + 1, row, &vq, M, c)) e = sgn * vq;        // A(k+1, k)
      if (j >= 1 && csr_find(rp, col, val, row - W, row, &vq, M, c)) s = sgn * vq;            // A(k-W, k)
      if (j <= H - 2 && csr_find(rp, col, val, row + W, row, &vq, M, c)) nn = sgn * vq;       // A(k+W, k)
      if (fo >= 0) {
        const int cand[4] = {row - g.xw[c], row + g.xw[c], row - g.yw[c], row + g.yw[c]};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = cand[q];
          if (m < 0 || m >= n || m == row - 1 || m == row + 1 || m == row - W || m == row + W || m == row) continue;
          if (csr_find(rp, col, val, m, row, &vq, M, c)) {
            if (ne < kExcSlots) { ec[ne] = m; ev[ne] = sgn * vq; ++ne; } else bad = true;
          }
        }
      }
    }
    const int k = a.kx(c, row);
    a.cS[k] = s; a.cW[k] = w; a.cC[k] = cc; a.cE[k] = e; a.cN[k] = nn;
    if (fo >= 0) {
      const int base = (g.f0[c] + fo) * kExcSlots;
      for (int q = 0; q < kExcSlots; ++q) {
        a.ecol[base + q] = q < ne ? ec[q] : -1;
        a.eval[base + q] = q < ne ? ev[q] : (T)0;
      }
    }
    nan_seen = nan_seen || is_nan(a.rhs[k]) || is_nan(x0[k]);
    a.x[k] = x0[k];                                                     // cublas copy x_old -> x (:261)
  }
  if (bad) a.flags[0] = 1;
  if (nan_seen) a.flags[1] = 1;
}

// ------------------------------------------------------------------------------------------------------------------
// block-wide exclusive scan of a monoid over kBlock threads (thread order = element order)
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
struct Affine {   // y -> m y + c
  T m, c;
  static __device__ __forceinline__ Affine identity() { return {(T)1, (T)0}; }
  // apply `first`, then `second`
  static __device__ __forceinline__ Affine then(const Affine& first, const Affine& second) {
    return {second.m * first.m, fma(second.m, first.c, second.c)};
  }
  __device__ __forceinline__ Affine shfl_up(int off) const { return {__shfl_up(m, off, kWave), __shfl_up(c, off, kWave)}; }
};

template <typename T>
struct Moebius {  // x -> (a x + b) / (c x + d), kept normalised
  T a, b, c, d;
  static __device__ __forceinline__ Moebius identity() { return {(T)1, (T)0, (T)0, (T)1}; }
  static __device__ __forceinline__ Moebius then(const Moebius& f, const Moebius& s) {   // matrix(s) * matrix(f)
    Moebius o = {fma(s.a, f.a, s.b * f.c), fma(s.a, f.b, s.b * f.d), fma(s.c, f.a, s.d * f.c), fma(s.c, f.b, s.d * f.d)};
    T mx = fmax(fmax(fabs(o.a), fabs(o.b)), fmax(fabs(o.c), fabs(o.d)));
    if (mx > 0 && mx == mx && mx < (T)1e30) { const T inv = (T)1 / mx; o.a *= inv; o.b *= inv; o.c *= inv; o.d *= inv; }
    return o;
  }
  __device__ __forceinline__ Moebius shfl_up(int off) const {
    return {__shfl_up(a, off, kWave), __shfl_up(b, off, kWave), __shfl_up(c, off, kWave), __shfl_up(d, off, kWave)};
  }
};

// Exclusive scan: returns the composition of the values of all threads BEFORE this one (identity for thread 0).
// smem: 4 values of M.  Two __syncthreads.
template <typename M>
__device__ __forceinline__ M block_exclusive_scan(M v, M* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const M o = v.shfl_up(off);
    if (lane >= off) v = M::then(o, v);
  }
  __syncthreads();
  if (lane == 63) smem[wave] = v;
  __syncthreads();
  M prefix = M::identity();
  for (int w = 0; w < wave; ++w) prefix = M::then(prefix, smem[w]);
  M excl = v.shfl_up(1);
  if (lane == 0) excl = M::identity();
  return M::then(prefix, excl);
}

// ------------------------------------------------------------------------------------------------------------------
// structured block ILU(0): pivots d (stored as 1/d) and the sweep coefficient arrays, one workgroup per band
//   d(i,j) = cC - cW(i,j) cE(i-1,j) / d(i-1,j) - [j > band start] cS(i,j) cN(i,j-1) / d(i,j-1)
//   LW = cW(k)/d(k-1), LS = cS(k)/d(k-W), UE = cE(k)/d(k), UN = cN(k)/d(k) (LS / UN zero across band edges)
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int E>
__global__ __launch_bounds__(kBlock) void bi_factor(BiArgs<T> a) {
  __shared__ Moebius<T> smem[4];
  const int c = blockIdx.y;
  const Geo& g = a.g;
  const int band = a.bb[c] + blockIdx.x;
  if (band >= a.be[c]) return;
  const int W = g.W[c], H = g.H[c];
  const int j0 = band * g.R, j1 = min(j0 + g.R, H);
  const int i0 = threadIdx.x * E;
  T d_prev_row[E];          // pivots of the previous row at my columns
  T cN_prev_row[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { d_prev_row[e] = 1; cN_prev_row[e] = 0; }
  for (int j = j0; j < j1; ++j) {
    const int kb = a.kx(c, j * W);
    T A[E], B[E], cw[E], cs[E], ce[E], cn[E];
    Moebius<T> f = Moebius<T>::identity();
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = i0 + e;
      A[e] = 1; B[e] = 0; cw[e] = cs[e] = ce[e] = cn[e] = 0;
      if (i < W) {
        const int k = kb + i;
        cw[e] = a.cW[k]; cs[e] = a.cS[k]; ce[e] = a.cE[k]; cn[e] = a.cN[k];
        const T ce_left = (i >= 1) ? a.cE[k - 1] : (T)0;
        A[e] = a.cC[k] - ((j > j0) ? cs[e] * cN_prev_row[e] / d_prev_row[e] : (T)0);
        B[e] = cw[e] * ce_left;
        const Moebius<T> fi = {A[e], -B[e], (T)1, (T)0};
        f = Moebius<T>::then(f, fi);
      }
    }
    const Moebius<T> pre = block_exclusive_scan(f, smem);
    // pivot just left of my chunk: pre(infinity) = a / c  (thread 0: unused because B == 0 at i == 0)
    T dl = (threadIdx.x == 0 || pre.c == 0) ? (T)1 : pre.a / pre.c;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = i0 + e;
      if (i < W) {
        const int k = kb + i;
        const T d = A[e] - B[e] / dl;
        const T inv = (T)1 / d;
        a.L[k] = Pair<T>{(i >= 1) ? cw[e] / dl : (T)0, (j > j0) ? cs[e] / d_prev_row[e] : (T)0};
        a.U[k] = Tri<T>{inv, ce[e] * inv, (j < j1 - 1) ? cn[e] * inv : (T)0};
        dl = d;
        d_prev_row[e] = d;
        cN_prev_row[e] = cn[e];
      }
    }
  }
}

template <int E>
__device__ __forceinline__ int sweep_slot(int s) { return (E % 2 == 0) ? s + (s >> 5) : s; }
template <typename T, int E>
// (rows of up to 1 024 faces - E <= 4 - are no faster this way: 512^2 89.6 against 97.1 us per iteration, 1024^2 166.0 against 160.9)
constexpr bool kSweepLds = E >= 5 && (size_t)4 * (E * kBlock + E * 8) * sizeof(T) <= (size_t)96 * 1024;

// bi_factor with coalesced memory accesses (see bi_sweep_lds): a row's five coefficients come in element order, are staged in LDS,
// read back E consecutive elements per thread; the five results go the same way back.  Bitwise bi_factor's results.
template <typename T, int E>
__global__ __launch_bounds__(kBlock) void bi_factor_lds(BiArgs<T> a) {
  constexpr int kRow = E * kBlock + E * 8;
  __shared__ Moebius<T> smem[4];
  __shared__ T bw[kRow], bs[kRow], bc[kRow], be[kRow], bn[kRow];
  const int c = blockIdx.y;
  const Geo& g = a.g;
  const int band = a.bb[c] + blockIdx.x;
  if (band >= a.be[c]) return;
  const int W = g.W[c], H = g.H[c];
  const int j0 = band * g.R, j1 = min(j0 + g.R, H);
  const int i0 = threadIdx.x * E;
  T d_prev_row[E], cN_prev_row[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { d_prev_row[e] = 1; cN_prev_row[e] = 0; }
  T lw[E], ls[E], lc[E], le[E], ln[E];
  auto load_row = [&](int j) __attribute__((always_inline)) {
    const int kb = a.kx(c, j * W);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = e * kBlock + threadIdx.x;
      lw[e] = ls[e] = lc[e] = le[e] = ln[e] = 0;
      if (i < W && j < j1) {
        const int k = kb + i;
        lw[e] = a.cW[k]; ls[e] = a.cS[k]; lc[e] = a.cC[k]; le[e] = a.cE[k]; ln[e] = a.cN[k];
      }
    }
  };
  auto stage_row = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = e * kBlock + threadIdx.x;
      if (i < W) {
        const int q = sweep_slot<E>(i);
        bw[q] = lw[e]; bs[q] = ls[e]; bc[q] = lc[e]; be[q] = le[e]; bn[q] = ln[e];
      }
    }
  };
  load_row(j0);
  stage_row();
  __syncthreads();
  for (int j = j0; j < j1; ++j) {
    const int kb = a.kx(c, j * W);
    T A[E], B[E], cw[E], cs[E], ce[E], cn[E];
    Moebius<T> f = Moebius<T>::identity();
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = i0 + e;
      A[e] = 1; B[e] = 0; cw[e] = cs[e] = ce[e] = cn[e] = 0;
      if (i < W) {
        const int q = sweep_slot<E>(i);
        cw[e] = bw[q]; cs[e] = bs[q]; ce[e] = be[q]; cn[e] = bn[q];
        const T ce_left = (i >= 1) ? (e > 0 ? ce[e - 1] : be[sweep_slot<E>(i - 1)]) : (T)0;
        A[e] = bc[q] - ((j > j0) ? cs[e] * cN_prev_row[e] / d_prev_row[e] : (T)0);
        B[e] = cw[e] * ce_left;
        const Moebius<T> fi = {A[e], -B[e], (T)1, (T)0};
        f = Moebius<T>::then(f, fi);
      }
    }
    load_row(j + 1);                              // (in flight during the scan)
    const Moebius<T> pre = block_exclusive_scan(f, smem);      // (its barriers: every thread has read the staged row)
    T dl = (threadIdx.x == 0 || pre.c == 0) ? (T)1 : pre.a / pre.c;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = i0 + e;
      if (i < W) {
        const int q = sweep_slot<E>(i);
        const T d = A[e] - B[e] / dl;
        const T inv = (T)1 / d;
        bw[q] = (i >= 1) ? cw[e] / dl : (T)0;                           // LW
        bs[q] = (j > j0) ? cs[e] / d_prev_row[e] : (T)0;                // LS
        bc[q] = inv;
        be[q] = ce[e] * inv;                                            // UE
        bn[q] = (j < j1 - 1) ? cn[e] * inv : (T)0;                      // UN
        dl = d;
        d_prev_row[e] = d;
        cN_prev_row[e] = cn[e];
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = e * kBlock + threadIdx.x;
      if (i < W) {
        const int q = sweep_slot<E>(i);
        a.L[kb + i] = Pair<T>{bw[q], bs[q]};
        a.U[kb + i] = Tri<T>{bc[q], be[q], bn[q]};
      }
    }
    __syncthreads();                              // (the results are out of LDS)
    stage_row();                                  // row j + 1
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// triangular sweeps inside a band, one workgroup per band, rows sequential, x-recurrence by an affine block scan
//   forward  (L y = in):            y(k) = in(k)         - LW(k) y(k-1) - LS(k) y(k-W)
//   backward (U z = y):             z(k) = y(k) dinv(k)  - UE(k) z(k+1) - UN(k) z(k+W)
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ CompScalars<T> folded_scalars(const BiArgs<T>& a, int c, T* smem);   // (with the scalar stages, below)

template <typename T, int E, bool FWD>
__global__ __launch_bounds__(kBlock) void bi_sweep(BiArgs<T> a, const T* in, T* __restrict__ out) {
  __shared__ Affine<T> smem[4];
  __shared__ T smem_fold[16];
  const int c = blockIdx.y;
  const Geo& g = a.g;
  const CompScalars<T> sc = folded_scalars(a, c, smem_fold);  // (folded: the ||s|| test in front of s_hat; ||r|| test, rho / beta in front of p_hat)
  if (sc.done) return;
  const bool fuse = FWD && a.fuse_p;
  const int band = a.bb[c] + blockIdx.x;
  if (band >= a.be[c]) return;
  const int W = g.W[c], H = g.H[c];
  const int j0 = band * g.R, j1 = min(j0 + g.R, H);
  const Pair<T>* __restrict__ cl = a.L;
  const Tri<T>* __restrict__ cu = a.U;
  // element order along the scan: forward i ascending, backward i descending
  const int i0 = threadIdx.x * E;
  T prev[E];
#pragma unroll
  for (int e = 0; e < E; ++e) prev[e] = 0;
  // The rows of a band are sequential (y(k) needs y(k-W)), but what a row READS from memory does not depend on the
  // recurrence: the inputs of row jj+1 are loaded while row jj is scanned (a row is latency-, not bandwidth-bound).
  T nv[E], na[E], nb[E], nd[E], n2[E], n3[E];
  auto load_row = [&](int jj) __attribute__((always_inline)) {
    const int j = FWD ? j0 + jj : j1 - 1 - jj;
    const int kb = a.kx(c, j * W);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int s = i0 + e;
      const int i = FWD ? s : W - 1 - s;
      nv[e] = 0; na[e] = 0; nb[e] = 0; nd[e] = 0; n2[e] = 0; n3[e] = 0;
      if (s < W && jj < j1 - j0) {
        const int k = kb + i;
        nv[e] = in[k];
        if (fuse) { n2[e] = a.v[k]; n3[e] = a.r[k]; }
        if (FWD) { const Pair<T> cf = cl[k]; na[e] = cf.a; nb[e] = cf.b; }
        else { const Tri<T> cf = cu[k]; nd[e] = cf.d; na[e] = cf.a; nb[e] = cf.b; }      // (y / d is formed where the row is consumed: a product
                                                                                        // here would wait for the loads BEFORE the scan they hide behind)
      }
    }
  };
  load_row(0);
  for (int jj = 0; jj < j1 - j0; ++jj) {
    const int j = FWD ? j0 + jj : j1 - 1 - jj;
    const int kb = a.kx(c, j * W);
    T m[E], cst[E];
    Affine<T> f = Affine<T>::identity();
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int s = i0 + e;                       // position along the scan
      m[e] = 0; cst[e] = 0;
      if (s < W) {
        T v = FWD ? nv[e] : nv[e] * nd[e];
        if (fuse) { v = (nv[e] - sc.omega * n2[e]) * sc.beta + n3[e]; a.p[kb + s] = v; }
        cst[e] = v - nb[e] * prev[e];
        m[e] = -na[e];
        f = Affine<T>::then(f, Affine<T>{m[e], cst[e]});
      }
    }
    load_row(jj + 1);                             // (in flight during the scan below)
    const Affine<T> pre = block_exclusive_scan(f, smem);
    T yl = pre.c;                                 // value just before my chunk (pre applied to 0; first m is 0 anyway)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int s = i0 + e;
      const int i = FWD ? s : W - 1 - s;
      if (s < W) {
        const T y = fma(m[e], yl, cst[e]);
        out[kb + i] = y;
        prev[e] = y;
        yl = y;
      }
    }
  }
}

// The same sweep with every global access COALESCED (round 5).  In bi_sweep a thread loads and stores the E consecutive elements it
// scans: lanes E * 4 bytes apart, every load / store instruction touches E cache lines' worth of address range per 64 lanes and every
// line is visited by E instructions - the texture addresser, not HBM, set the pace (2048^2: 2.0 / 3.6 TB/s forward / backward).  Here
// a row moves between memory and LDS in element order (lane = consecutive element) and between LDS and the scan's registers in
// thread order (E consecutive elements per thread; odd strides are conflict-free, even ones padded by one word per 32).  One more
// barrier per row; the arithmetic, the scan and therefore every result are bitwise bi_sweep's.  2048^2, per sweep: forward 47 -> 29 us,
// backward 68 -> 33 us (of which 59 -> 33 by forming y / d behind the scan instead of in front of it, see load_row).
template <typename T, int E, bool FWD>
__global__ __launch_bounds__(kBlock) void bi_sweep_lds(BiArgs<T> a, const T* in, T* __restrict__ out) {
  constexpr int kRow = E * kBlock + E * 8;
  __shared__ Affine<T> smem[4];
  __shared__ T smem_fold[16];
  __shared__ T bv[kRow], ba[kRow], bb[kRow], by[kRow];
  const int c = blockIdx.y;
  const Geo& g = a.g;
  const CompScalars<T> sc = folded_scalars(a, c, smem_fold);
  if (sc.done) return;
  const bool fuse = FWD && a.fuse_p;
  const int band = a.bb[c] + blockIdx.x;
  if (band >= a.be[c]) return;
  const int W = g.W[c], H = g.H[c];
  const int j0 = band * g.R, j1 = min(j0 + g.R, H);
  const Pair<T>* __restrict__ cl = a.L;
  const Tri<T>* __restrict__ cu = a.U;
  const int i0 = threadIdx.x * E;
  T prev[E];
#pragma unroll
  for (int e = 0; e < E; ++e) prev[e] = 0;
  T nv[E], na[E], nb[E], nd[E], n2[E], n3[E];
  // memory side: column i = e * kBlock + thread (ascending addresses in both directions); its place along the scan is i (forward) or
  // W - 1 - i (backward)
  auto load_row = [&](int jj) __attribute__((always_inline)) {
    const int j = FWD ? j0 + jj : j1 - 1 - jj;
    const int kb = a.kx(c, j * W);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = e * kBlock + threadIdx.x;
      nv[e] = 0; na[e] = 0; nb[e] = 0; nd[e] = 0; n2[e] = 0; n3[e] = 0;
      if (i < W && jj < j1 - j0) {
        const int k = kb + i;
        nv[e] = in[k];
        if (fuse) { n2[e] = a.v[k]; n3[e] = a.r[k]; }
        if (FWD) { const Pair<T> cf = cl[k]; na[e] = cf.a; nb[e] = cf.b; }
        else { const Tri<T> cf = cu[k]; nd[e] = cf.d; na[e] = cf.a; nb[e] = cf.b; }      // (y / d is formed where the row is consumed: a product
                                                                                        // here would wait for the loads BEFORE the scan they hide behind)
      }
    }
  };
  auto stage_row = [&](int jj) __attribute__((always_inline)) {      // the row load_row(jj) asked for
    const int kb = a.kx(c, (FWD ? j0 + jj : j1 - 1 - jj) * W);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = e * kBlock + threadIdx.x;
      if (i < W) {
        const int q = sweep_slot<E>(FWD ? i : W - 1 - i);
        T v = FWD ? nv[e] : nv[e] * nd[e];
        if (fuse && jj < j1 - j0) { v = (nv[e] - sc.omega * n2[e]) * sc.beta + n3[e]; a.p[kb + i] = v; }
        bv[q] = v; ba[q] = na[e]; bb[q] = nb[e];
      }
    }
  };
  load_row(0);
  stage_row(0);
  __syncthreads();
  for (int jj = 0; jj < j1 - j0; ++jj) {
    const int j = FWD ? j0 + jj : j1 - 1 - jj;
    const int kb = a.kx(c, j * W);
    T m[E], cst[E];
    Affine<T> f = Affine<T>::identity();
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int s = i0 + e;                       // position along the scan
      m[e] = 0; cst[e] = 0;
      if (s < W) {
        const int q = sweep_slot<E>(s);
        cst[e] = bv[q] - bb[q] * prev[e];
        m[e] = -ba[q];
        f = Affine<T>::then(f, Affine<T>{m[e], cst[e]});
      }
    }
    load_row(jj + 1);                             // (in flight during the scan below)
    const Affine<T> pre = block_exclusive_scan(f, smem);      // (its barriers: every thread has read the staged row)
    T yl = pre.c;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int s = i0 + e;
      if (s < W) {
        const T y = fma(m[e], yl, cst[e]);
        by[sweep_slot<E>(s)] = y;
        prev[e] = y;
        yl = y;
      }
    }
    stage_row(jj + 1);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = e * kBlock + threadIdx.x;
      if (i < W) out[kb + i] = by[sweep_slot<E>(FWD ? i : W - 1 - i)];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// vector kernels (blockIdx.y = component); partial sums to parts[c][q][blockIdx.x]
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void store_partials(BiArgs<T>& a, int c, T* vals, int count, T* smem, int slot0 = 0) {
  // vals[0..count) already thread-local sums
  T tmp[4] = {0, 0, 0, 0};
  for (int q = 0; q < count; ++q) tmp[q] = vals[q];
  block_sum<T, 4>(tmp, smem);
  if (threadIdx.x == 0)
    for (int q = 0; q < count; ++q) a.parts[(c * 4 + q) * kBiParts + slot0 + blockIdx.x] = tmp[q];
}

// ---- the scalar recurrences between the vector kernels (multi_bicgstab_ilu_linear_solve_op.cu.cc:263-408): one STAGE per reduction
enum { ST_INIT = 0, ST_RHO_BETA = 1, ST_ALPHA = 2, ST_CHECK_S = 3, ST_OMEGA = 4, ST_CHECK_R = 5 };

template <typename T>
__device__ __forceinline__ T sqrt_t(T v);
template <>
__device__ __forceinline__ float sqrt_t<float>(float v) { return sqrtf(v); }
template <>
__device__ __forceinline__ double sqrt_t<double>(double v) { return sqrt(v); }

// q: the (up to four) sums of the previous vector kernel, in its order
template <typename T>
__device__ __forceinline__ void apply_stage(CompScalars<T>& s, int stage, const T (&q)[4], T tol) {
  switch (stage) {
    case ST_INIT:                       // ||r0|| "lucky guess" test (:288-292); rho for the first iteration is rh.r = ||r||^2
      s.nrm = sqrt_t<T>(q[0]);
      if (s.nrm < tol) s.done = 1;
      s.rho_prev = s.rho;               // rho / alpha / omega are NOT reset on a restart (as coded)
      s.rho = q[0];
      s.beta = (s.rho / s.rho_prev) * (s.alpha / s.omega);
      break;
    case ST_RHO_BETA:                   // start of an iteration after the first: rho = rh.r (:309-312)
      s.rho_prev = s.rho;
      s.rho = q[1];
      s.beta = (s.rho / s.rho_prev) * (s.alpha / s.omega);
      break;
    case ST_ALPHA:                      // alpha = rho / rh.v (:336-338)
      s.alpha = s.rho / q[0];
      s.it_count += 1;                  // one ST_ALPHA per started iteration (it_count++, :306)
      break;
    case ST_CHECK_S:                    // ||s|| test (:347-351)
      s.nrm = sqrt_t<T>(q[0]);
      if (s.nrm < tol) s.done = 1;
      break;
    case ST_OMEGA:                      // omega = t.r / t.t (:372-374)
      s.omega = q[0] / q[1];
      break;
    case ST_CHECK_R:                    // ||r|| test (:386-390)
      s.nrm = sqrt_t<T>(q[0]);
      if (s.nrm < tol) s.done = 1;
      break;
  }
}

// the four sums of component c from the partial records of the previous producer, in ONE fixed order (every caller - the scalar
// kernel, every block of a kernel with folded stages - gets bitwise the same sums)
template <typename T>
__device__ __forceinline__ void sum_partials(const BiArgs<T>& a, int c, T (&q)[4], T* smem) {
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] = 0;
  for (int b = threadIdx.x; b < a.nparts; b += kBlock) {       // (four independent loads in flight per pass)
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] += a.parts_in[(c * 4 + k) * kBiParts + b];
  }
  block_sum<T, 4>(q, smem);
}

// The scalars a vector kernel works with.  fold == 0: the record a scalar kernel left in a.sc.  Otherwise (one GPU, small systems -
// a scalar launch of 4 us + its gap is a fifth of the kernels of an iteration there): EVERY block of the consuming kernel applies the
// stages itself - low nibble (stage + 1) first, then the high nibble if set - to the record a.sc_prev with the sums of a.parts_in, all
// in the scalar kernel's order, so every block holds bitwise the same scalars; block 0 of the component stores them to a.sc (the
// OTHER of two records: no block of this launch reads what it writes).  Called by all threads of the block before anything diverges.
template <typename T>
__device__ __forceinline__ CompScalars<T> folded_scalars(const BiArgs<T>& a, int c, T* smem) {
  if (a.fold == 0) return a.sc[c];
  CompScalars<T> s = a.sc_prev[c];
  if (!s.done) {
    T q[4];
    sum_partials(a, c, q, smem);
    apply_stage(s, (a.fold & 15) - 1, q, (T)a.tol);
    if ((a.fold >> 4) != 0 && !s.done) apply_stage(s, (a.fold >> 4) - 1, q, (T)a.tol);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) a.sc[c] = s;
  return s;
}

// p = r + beta (p - omega v)   (:316-318)
template <typename T>
__global__ __launch_bounds__(kBlock) void bi_update_p(BiArgs<T> a) {
  __shared__ T smem[16];
  const int c = blockIdx.y;
  const CompScalars<T> s = folded_scalars(a, c, smem);      // (folded: ||r|| test of the iteration before, then rho / beta)
  if (s.done) return;
  for (int row = a.rb[c] + blockIdx.x * kBlock + threadIdx.x; row < a.re[c]; row += gridDim.x * kBlock) {
    const int k = a.kx(c, row);
    a.p[k] = (a.p[k] - s.omega * a.v[k]) * s.beta + a.r[k];
  }
}

// out = B in ; partials: WHICH 0: rh.out (v = B p_hat)   1: out.r, out.out (t = B s_hat)
// part 0: all owned rows.  Slab mode splits the product so that the halo exchange of `in` overlaps the bulk of it: part 1 = the
// interior (owned rows minus kEdgeRows face rows at either end: they read owned rows only), part 2 = those edge rows (they read
// the neighbours' rows - and, across the periodic seam, the wrap partners v[ny] / v[0] / v[1] / v[ny - 1] of rows 1 / ny - 1 / ny / 0).
constexpr int kEdgeRows = 2;
// Four rows per thread and pass, 256 rows apart (every access of a wave is one run of consecutive elements): all loads of the four
// rows - five coefficients, five values of `in` at clamped addresses, the dot product's partner - are issued before anything is
// consumed, the store of a row cannot stand between the loads of the next (round 5: with one row per pass, a division per row and the
// stores in between the kernel moved 2.85 TB/s at 2048^2).  The row's sum keeps stencil_row's order: S, W, C, E, N, exceptions.
#ifndef PISO_SPMV_ROWS
#define PISO_SPMV_ROWS 4
#endif
constexpr int kSpmvRows = PISO_SPMV_ROWS;
// emit(k, o, d): row at place k of the vectors, o = (B in)(row), d = extra[k] (loaded with the row's other operands)
template <typename T, typename Emit>
__device__ __forceinline__ void stencil_rows(const BiArgs<T>& a, int c, const T* __restrict__ in, const T* __restrict__ extra, int part,
                                             Emit emit) {
  const Geo& g = a.g;
  const int W = g.W[c], H = g.H[c];
  const int eb = kEdgeRows * W;                             // elements of one edge
  const int begin = part == 1 ? a.rb[c] + eb : (part == 2 ? 0 : a.rb[c]);
  const int end = part == 1 ? a.re[c] - eb : (part == 2 ? 2 * eb : a.re[c]);
  const float inv_w = 1.0f / (float)W;
  // Which chunk of kSpmvRows * 256 rows a block takes: workgroups are dealt to the eight XCDs round-robin, and a row's neighbours
  // k - W / k + W sit two chunks away at 2048^2 - with chunks dealt round-robin too, every XCD's L2 fetched `in` three times (its own
  // chunks and both neighbours').  A grid of a multiple of 8 workgroups gives every XCD one contiguous eighth of the rows instead.
  constexpr int kChunk = kSpmvRows * kBlock;
  const int nchunk = (end - begin + kChunk - 1) / kChunk;
  const bool by_xcd = (gridDim.x & 7) == 0 && part != 2;
  const int per = by_xcd ? (nchunk + 7) / 8 : nchunk;                       // chunks of one XCD's eighth
  const int first = by_xcd ? (int)(blockIdx.x & 7) * per : 0;
  const int step = by_xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  for (int cidx = by_xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x; cidx < per; cidx += step) {
    const int base = begin + (first + cidx) * kChunk;
    if (base >= end) break;
    int k[kSpmvRows], fo[kSpmvRows];
    bool on[kSpmvRows], hs[kSpmvRows], hw[kSpmvRows], he[kSpmvRows], hn[kSpmvRows];
    T cs[kSpmvRows], cw[kSpmvRows], cc[kSpmvRows], ce[kSpmvRows], cn[kSpmvRows];
    T xs[kSpmvRows], xw[kSpmvRows], xc[kSpmvRows], xe[kSpmvRows], xn[kSpmvRows], d[kSpmvRows];
#pragma unroll
    for (int u = 0; u < kSpmvRows; ++u) {
      const int idx = base + u * kBlock + (int)threadIdx.x;
      on[u] = idx < end;
      const int idc = on[u] ? idx : end - 1;                  // (a row of the range: its loads are harmless, nothing is stored)
      const int row = part == 2 ? (idc < eb ? a.rb[c] + idc : a.re[c] - 2 * eb + idc) : idc;
      int j = (int)((float)row * inv_w), i = row - j * W;     // (rows < 2^31, j < 2^22: the float quotient is off by one at most)
      if (i < 0) { --j; i += W; } else if (i >= W) { ++j; i -= W; }
      fo[u] = frame_ordinal(i, j, W, H);
      k[u] = a.kx(c, row);
      hs[u] = j >= 1; hw[u] = i >= 1; he[u] = i <= W - 2; hn[u] = j <= H - 2;
      cs[u] = a.cS[k[u]]; cw[u] = a.cW[k[u]]; cc[u] = a.cC[k[u]]; ce[u] = a.cE[k[u]]; cn[u] = a.cN[k[u]];
      xs[u] = in[hs[u] ? k[u] - W : k[u]]; xw[u] = in[hw[u] ? k[u] - 1 : k[u]]; xc[u] = in[k[u]];
      xe[u] = in[he[u] ? k[u] + 1 : k[u]]; xn[u] = in[hn[u] ? k[u] + W : k[u]];
      d[u] = extra[k[u]];
    }
#pragma unroll
    for (int u = 0; u < kSpmvRows; ++u) {
      T o = 0;
      o = hs[u] ? fma(cs[u], xs[u], o) : o;
      o = hw[u] ? fma(cw[u], xw[u], o) : o;
      o = fma(cc[u], xc[u], o);
      o = he[u] ? fma(ce[u], xe[u], o) : o;
      o = hn[u] ? fma(cn[u], xn[u], o) : o;
      if (fo[u] >= 0) {
        const int eb0 = (g.f0[c] + fo[u]) * kExcSlots;
#pragma unroll
        for (int q = 0; q < kExcSlots; ++q) {
          const int ec = a.ecol[eb0 + q];
          if (ec >= 0) o = fma(a.eval[eb0 + q], in[a.kx(c, ec)], o);
        }
      }
      if (on[u]) emit(k[u], o, d[u]);
    }
  }
}

template <typename T, int WHICH>
__global__ __launch_bounds__(kBlock) void bi_spmv(BiArgs<T> a, const T* __restrict__ in, T* __restrict__ out, int part, int slot0) {
  __shared__ T smem[16];
  const int c = blockIdx.y;
  if (a.sc[c].done) return;
  T acc[2] = {0, 0};
  stencil_rows(a, c, in, WHICH == 0 ? a.rh : a.r, part, [&](int k, T o, T d) __attribute__((always_inline)) {
    out[k] = o;
    if (WHICH == 0) acc[0] = fma(d, o, acc[0]);
    else { acc[0] = fma(o, d, acc[0]); acc[1] = fma(o, o, acc[1]); }
  });
  store_partials(a, c, acc, WHICH == 0 ? 1 : 2, smem, slot0);
}

// r = rhs - B x ; rh = r ; p = v = 0 ; partial ||r||^2   (:266-300)
template <typename T>
__global__ __launch_bounds__(kBlock) void bi_residual_init(BiArgs<T> a) {
  __shared__ T smem[16];
  const int c = blockIdx.y;
  if (a.sc[c].done) return;
  T acc[1] = {0};
  stencil_rows(a, c, a.x, a.rhs, 0, [&](int k, T o, T d) __attribute__((always_inline)) {
    const T r = d - o;
    a.r[k] = r; a.rh[k] = r; a.p[k] = 0; a.v[k] = 0;
    acc[0] = fma(r, r, acc[0]);
  });
  store_partials(a, c, acc, 1, smem);
}

// x += coef * dir ; r -= coef * w ; partials ||r||^2, rh.r     (WHICH 0: alpha, p_hat, v ; 1: omega, s_hat, t)
template <typename T, int WHICH>
__global__ __launch_bounds__(kBlock) void bi_update_xr(BiArgs<T> a) {
  __shared__ T smem[16];
  const int c = blockIdx.y;
  const CompScalars<T> s = folded_scalars(a, c, smem);      // (folded: alpha / omega from the product's sums)
  if (s.done) return;
  const T coef = WHICH == 0 ? s.alpha : s.omega;
  const T* __restrict__ dir = WHICH == 0 ? a.ph : a.sh;
  const T* __restrict__ w = WHICH == 0 ? a.v : a.t;
  T acc[2] = {0, 0};
  for (int row = a.rb[c] + blockIdx.x * kBlock + threadIdx.x; row < a.re[c]; row += gridDim.x * kBlock) {
    const int k = a.kx(c, row);
    a.x[k] = a.x[k] + coef * dir[k];
    const T r = a.r[k] - coef * w[k];
    a.r[k] = r;
    acc[0] = fma(r, r, acc[0]);
    acc[1] = fma(a.rh[k], r, acc[1]);
  }
  store_partials(a, c, acc, 2, smem);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void bi_zero_x(BiArgs<T> a, int comp_mask) {
  const int c = blockIdx.y;
  if (!((comp_mask >> c) & 1)) return;
  for (int row = a.rb[c] + blockIdx.x * kBlock + threadIdx.x; row < a.re[c]; row += gridDim.x * kBlock) a.x[a.kx(c, row)] = 0;
}

// ------------------------------------------------------------------------------------------------------------------
// scalar kernels: one block per component reduces the partials of the previous vector kernel and advances the recurrences
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void bi_scalar(BiArgs<T> a, int stage, BiPeer bp) {
  __shared__ T smem[16];
  const int c = blockIdx.x;
  CompScalars<T> s = a.sc[c];
  if (s.done) {                   // (the same decision on every rank: `done` follows from all-reduced sums)
    if (bp.on == 2 && threadIdx.x < 4) bp.gsum[c * 4 + threadIdx.x] = 0;
    return;
  }
  T q[4] = {0, 0, 0, 0};
  if (bp.on == 3) {
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = (T)bp.gsum[c * 4 + k];
  } else {
    sum_partials(a, c, q, smem);
  }
  if (bp.on == 2) {
    if (threadIdx.x < 4) bp.gsum[c * 4 + threadIdx.x] = (double)(threadIdx.x == 0 ? q[0] : (threadIdx.x == 1 ? q[1] : (threadIdx.x == 2 ? q[2] : q[3])));
    return;
  }
  if (bp.on == 1 && threadIdx.x < 64) {
    // the distributed dot products: component c's four sums travel in words [8 c, 8 c + 8) of the all-reduce records
    const int lane = threadIdx.x;
    const int vq = (lane >> 1) & 3;
    bool good = true;
    const double acc = peer_wave_sum(bp.pv, (double)(vq == 0 ? q[0] : (vq == 1 ? q[1] : (vq == 2 ? q[2] : q[3]))), 8, 8 * c, bp.seq, &good);
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = (T)__shfl(acc, 2 * k, 64);
    if (!good && lane == 0) *bp.err = 1;
  }
  if (threadIdx.x != 0) return;
  apply_stage(s, stage, q, (T)a.tol);
  a.sc[c] = s;
}

template <typename T>
__global__ void bi_init_scalars(BiArgs<T> a) {
  if (threadIdx.x < 2) {
    CompScalars<T> s;
    s.rho = 1; s.rho_prev = 1; s.alpha = 1; s.omega = 1; s.beta = 0; s.nrm = 0;
    s.done = 0; s.it_count = 0; s.failed = 0; s.pad = 0;
    a.sc[threadIdx.x] = s;
  }
  for (int i = threadIdx.x; i < 2 * 2 * 4 * kBiParts; i += blockDim.x) a.parts[i] = 0;      // (both buffers)
  if (threadIdx.x == 0) { a.flags[0] = 0; a.flags[1] = 0; }
}

// slab mode: "unsupported pattern" / "NaN seen" of ANY rank's rows count for every rank
template <typename T>
__global__ void bi_flags_allreduce(BiArgs<T> a, BiPeer bp) {
  const int lane = threadIdx.x;
  bool good = true;
  const double acc = peer_wave_sum(bp.pv, lane < 4 ? (double)a.flags[lane >> 1] : 0.0, 4, 0, bp.seq, &good);
  if (lane < 4 && (lane & 1) == 0) a.flags[lane >> 1] = acc > 0 ? 1 : 0;
  if (!good && lane == 0) *bp.err = 1;
}

template <typename T>
__global__ void bi_set_done(BiArgs<T> a, int done0, int done1) {
  if (threadIdx.x == 0) { a.sc[0].done = done0; a.sc[1].done = done1; }
}

// ------------------------------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------------------------------
static Geo make_geo(int nx, int ny, int band_rows) {
  Geo g;
  g.nx = nx; g.ny = ny;
  g.W[0] = nx + 1; g.H[0] = ny; g.W[1] = nx; g.H[1] = ny + 1;
  for (int c = 0; c < 2; ++c) {
    g.n[c] = g.W[c] * g.H[c];
    g.xw[c] = g.W[c] - 1 - (c == 0);
    g.yw[c] = g.W[c] * (g.H[c] - 1 - (c == 1));
    g.F[c] = frame_rows(g.W[c], g.H[c]);
  }
  g.r0[0] = 0; g.r0[1] = g.n[0];
  g.f0[0] = 0; g.f0[1] = g.F[0];
  int R = band_rows;
  if (R < 0) R = (ny + 1);                                   // one band: global structured ILU(0)
  // automatic (2048^2: 8 rows).  The rows of a band are sequential and a band is one workgroup: 2048^2 has 257 bands of 16 rows = ONE workgroup
  // of four waves per CU, and a sweep is then bound by the latency of its row chain (75 / 100 us); 513 bands of 8 rows keep two
  // workgroups per CU busy and halve the chain - 591 instead of 713 us per iteration, the SAME iteration counts (the matrices are
  // strongly diagonally dominant: 3 iterations to 1e-6, 5 to 1e-9 with bands of 4 .. 32 rows; a round-4 A/B script, results in profiles/README.md).  Bands of 4
  // rows gain nothing more at 2048^2 (the sweeps then move ~6 TB/s) and cost an iteration at 256^2.
  // Round 5: smaller grids get lower bands by the same argument - a band is one workgroup, and two components x ny / R bands should be
  // about two workgroups per CU: ny >= 2048: 8 rows, >= 1024: 4, >= 256: 2.  Measured (round 5, solve to 1e-6, same
  // iteration counts): 1024^2 0.833 -> 0.767 ms, 512^2 0.519 -> 0.429, 256^2 0.440 -> 0.366.
  // (grids of fewer than 256 rows - the lid-driven cavity - keep 8: nothing there is bound by the bands' parallelism, and at the
  // reference script's loose 1e-3 the preconditioner decides which iterate inside the tolerance a solve stops at)
  if (R == 0) R = ny >= 2048 ? 8 : (ny >= 1024 ? 4 : (ny >= 256 ? 2 : 8));
  if (R > ny + 1) R = ny + 1;
  g.R = R;
  for (int c = 0; c < 2; ++c) g.nb[c] = (g.H[c] + R - 1) / R;
  return g;
}

template <typename T>
static size_t bi_workspace_bytes(int nx, int ny, const piso_slab_t* slab = nullptr) {
  const Geo g = make_geo(nx, ny, 8);
  size_t ntot = (size_t)g.n[0] + g.n[1];
  if (slab) { const RowMap M = make_row_map(slab, nx, ny); ntot = (size_t)M.n_u + M.n_v; }      // (local storage: the rank's stored rows)
  size_t b = 0;
  b += 18 * align_up(ntot * sizeof(T), 256);
  b += align_up((size_t)(g.F[0] + g.F[1]) * kExcSlots * sizeof(int), 256);
  b += align_up((size_t)(g.F[0] + g.F[1]) * kExcSlots * sizeof(T), 256);
  b += align_up(2 * 2 * 4 * kBiParts * sizeof(T), 256) + align_up(2 * 2 * sizeof(CompScalars<T>), 256) + 256;
  return b + 8192;
}

// slab mode: the halo exchange of an SpMV input runs on this stream while the interior rows are multiplied on the caller's
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t ready = nullptr, halo = nullptr;
};
// one per device and thread: a thread that later solves on another GPU must not launch the exchange on a stream of the first
constexpr int kMaxDevices = 16;
static thread_local SideStream tl_side_dev[kMaxDevices];

template <typename T>
struct BiHost {
  CompScalars<T> sc[2];
  int flags[2];
};

template <typename T, int E>
static void launch_factor(const BiArgs<T>& a, dim3 gb, hipStream_t s) {
  if constexpr (kSweepLds<T, E> && (size_t)5 * (E * kBlock + E * 8) * sizeof(T) <= (size_t)96 * 1024) {
    if (opt(OPT_BICG_SWEEP_LDS) != 0) { bi_factor_lds<T, E><<<gb, kBlock, 0, s>>>(a); return; }
  }
  bi_factor<T, E><<<gb, kBlock, 0, s>>>(a);
}
template <typename T, int E>
static void launch_sweeps(const BiArgs<T>& aL, const BiArgs<T>& aU, dim3 gb, const T* in, T* out, hipStream_t s) {
  if constexpr (kSweepLds<T, E>) {
    if (opt(OPT_BICG_SWEEP_LDS) != 0) {
      bi_sweep_lds<T, E, true><<<gb, kBlock, 0, s>>>(aL, in, aL.y);
      bi_sweep_lds<T, E, false><<<gb, kBlock, 0, s>>>(aU, aU.y, out);
      return;
    }
  }
  bi_sweep<T, E, true><<<gb, kBlock, 0, s>>>(aL, in, aL.y);
  bi_sweep<T, E, false><<<gb, kBlock, 0, s>>>(aU, aU.y, out);
}

// pc = NULL: one GPU.  Else: this rank works on the face rows of its y-slab of cell rows.  slab = NULL: val / rowptr / col / rhs / x0
// are the FULL arrays on every rank (a replicated assembly), x_out is valid on the owned rows only.  slab != NULL (the slab-decomposed
// step, round 5): every array - the caller's and the workspace - holds the rank's STORED rows (piso_slab_t), 1 / ranks of the grid.
template <typename T>
static int bi_solve(const T* val, const int* rowptr, const int* col, const T* rhs, const T* x0, T* x_out, int nx,
                    int ny, float tol, int max_it, int transpose, int band_rows, uint8_t* warning,
                    int* iterations_out, void* ws, size_t ws_bytes, piso_stream_t stream_, PisoComm* pc = nullptr,
                    const piso_slab_t* slab_rows = nullptr, int per_x = 0, int per_y = 0) {
  if (nx < 4 || ny < 4 || !val || !rowptr || !col || !rhs || !x0 || !x_out || !ws || max_it < 0 || !slab_ok(slab_rows, ny) || (slab_rows && !pc)) {
    set_error_msg("piso_multi_bicgstab_ilu: invalid argument (need nx, ny >= 4 and non-NULL arrays)");
    return PISO_ERR_INVALID_ARG;
  }
  // `transpose` is a set of flags (bit 0: A^T, bit 1: the matrix is -csr_val); a caller that still means "any non-zero = transpose"
  // (2, -1, ...) would silently get a different system: everything outside the two bits is refused
  if ((transpose & ~3) != 0) {
    set_error_msg("piso_multi_bicgstab_ilu: transpose must be a combination of bit 0 (A^T) and bit 1 (negated matrix)");
    return PISO_ERR_INVALID_ARG;
  }
  if (ws_bytes < bi_workspace_bytes<T>(nx, ny, slab_rows)) {
    set_error_msg("piso_multi_bicgstab_ilu: workspace too small");
    return PISO_ERR_INVALID_ARG;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  BiArgs<T> a;
  a.g = make_geo(nx, ny, band_rows);
  a.M = make_row_map(slab_rows, nx, ny, per_x, per_y);
  const Geo& g = a.g;
  const size_t ntot = slab_rows ? (size_t)a.M.n_u + a.M.n_v : (size_t)g.n[0] + g.n[1];
  Arena ar(ws, ws_bytes);
  auto tk = [&](size_t n) { return ar.take<T>(n); };
  a.cS = tk(ntot); a.cW = tk(ntot); a.cC = tk(ntot); a.cE = tk(ntot); a.cN = tk(ntot);
  a.L = reinterpret_cast<Pair<T>*>(tk(2 * ntot)); a.U = reinterpret_cast<Tri<T>*>(tk(3 * ntot));
  a.r = tk(ntot); a.rh = tk(ntot); a.p = tk(ntot); a.v = tk(ntot); a.t = tk(ntot);
  a.y = tk(ntot); a.ph = tk(ntot); a.sh = tk(ntot);
  a.ecol = ar.take<int>((size_t)(g.F[0] + g.F[1]) * kExcSlots);
  a.eval = ar.take<T>((size_t)(g.F[0] + g.F[1]) * kExcSlots);
  T* const pbuf0 = ar.take<T>(2 * 2 * 4 * kBiParts);                      // two buffers of partial sums (see BiArgs::parts_in)
  CompScalars<T>* const scbuf0 = ar.take<CompScalars<T>>(2 * 2);         // two scalar records (see BiArgs::sc_prev)
  a.parts = pbuf0; a.parts_in = pbuf0; a.sc = scbuf0; a.sc_prev = scbuf0; a.fold = 0; a.fuse_p = 0;
  a.flags = ar.take<int>(2);
  a.rhs = rhs; a.x = x_out; a.tol = tol;
  if (!ar.ok()) { set_error_msg("piso_multi_bicgstab_ilu: workspace too small"); return PISO_ERR_INVALID_ARG; }

  for (int c = 0; c < 2; ++c) { a.rb[c] = 0; a.re[c] = g.n[c]; a.bb[c] = 0; a.be[c] = g.nb[c]; }
  BiPeer bp;
  bp.on = 0; bp.seq = 0; bp.err = nullptr; bp.gsum = nullptr;
  HaloMsg to_upper = {}, to_lower = {}, from_lower = {}, from_upper = {};
  const bool slab = pc && (pc->world > 1 || opt(OPT_SLAB_FORCE) > 0);      // (slab_force: test knob - one rank, a ring with itself)
  const bool rccl = pc && pc->transport == TRANSPORT_RCCL;                   // halo rows by send / recv, sums by all-reduce (slab_comm.h)
  if (pc) {
    if (pc->transport == TRANSPORT_PEER && !pc->connected) { set_error_msg("piso_multi_bicgstab_ilu_slab: the peer communicator is not connected"); return PISO_ERR_INVALID_ARG; }
    const int world = pc->world, rank = pc->rank;
    // (a product is split into interior rows and kEdgeRows face rows at either end of the slab: thinner slabs would make the two
    // edge ranges overlap and count their rows twice in the dot products)
    if (ny % world != 0 || (ny / world) % g.R != 0 || ny / world < 2 * kEdgeRows) {
      set_error_msg("piso_multi_bicgstab_ilu_slab: the slabs (ny / ranks cell rows) must be whole preconditioner bands of at least 4 rows");
      return PISO_ERR_INVALID_ARG;
    }
    if (!rccl && (size_t)(3 * nx + 1) > pc->row_cap) { set_error_msg("piso_multi_bicgstab_ilu_slab: communicator row_capacity < 3 nx + 1"); return PISO_ERR_INVALID_ARG; }
    const int nyl = ny / world, jb = rank * nyl, jt = jb + nyl - 1;
    const bool last = rank == world - 1;
    if (slab_rows && (slab_rows->row_begin != jb || slab_rows->row_end != jb + nyl || (slab_rows->owns_last_face_row != 0) != last)) {
      set_error_msg("piso_multi_bicgstab_ilu_slab: the slab does not match the communicator's rank");
      return PISO_ERR_INVALID_ARG;
    }
    a.rb[0] = jb * g.W[0]; a.re[0] = (jb + nyl) * g.W[0];
    a.rb[1] = jb * g.W[1]; a.re[1] = (jb + nyl + (last ? 1 : 0)) * g.W[1];      // (the duplicate face row v[ny] lives on the last slab)
    a.bb[0] = a.bb[1] = jb / g.R;
    a.be[0] = (jb + nyl) / g.R;
    a.be[1] = last ? g.nb[1] : (jb + nyl) / g.R;
    bp.pv = make_view(pc, true);       // always a ring: without periodic y the wrap rows travel but no matrix entry reads them
    bp.err = pc->err; bp.on = slab ? 1 : 0;
    if (rccl) { bp.gsum = ar.take<double>(8); if (!ar.ok()) { set_error_msg("piso_multi_bicgstab_ilu_slab: workspace too small"); return PISO_ERR_INVALID_ARG; } }
    // Edge rows of an SpMV input; they land at the same global offsets on the receiver.  Downwards go u[jb], v[jb] and v[jb + 1],
    // upwards u[jt], v[jt] - and, across the periodic seam, the duplicate row v[ny] as well: in A the row v[ny] reads v[1] and
    // v[0] reads v[ny - 1] (the wrap skips the duplicate face, central_difference_csr_op.cu.cc:259-264), in A^T it is v[1] that
    // reads v[ny] and v[ny - 1] that reads v[0].
    const int lo = bp.pv.lower, up = bp.pv.upper;
    const int jt_lo = lo * nyl + nyl - 1, jb_up = up * nyl;
    // (where a row starts in the concatenated vectors: the whole grid's offsets, or the stored rows' - rows that follow each other
    // around the ring - v[ny - 1], v[ny], v[0], v[1] - are neighbours in the stored arrays as well)
    auto at = [&](int c, int j) -> int { return a.M.on ? (c ? a.M.n_u : 0) + a.M.frow(c, j * g.W[c]) : g.r0[c] + j * g.W[c]; };
    to_lower = {2, {at(0, jb), at(1, jb), 0}, {g.W[0], 2 * g.W[1], 0}};
    to_upper = {2, {at(0, jt), at(1, jt), 0}, {g.W[0], (last ? 2 : 1) * g.W[1], 0}};
    from_lower = {2, {at(0, jt_lo), at(1, jt_lo), 0}, {g.W[0], (lo == world - 1 ? 2 : 1) * g.W[1], 0}};
    from_upper = {2, {at(0, jb_up), at(1, jb_up), 0}, {g.W[0], 2 * g.W[1], 0}};
  }
  auto next_seq = [&]() -> BiPeer { BiPeer b = bp; if (slab) b.seq = ++pc->seq_ar; return b; };
  const HaloMsg msgs[4] = {to_upper, to_lower, from_lower, from_upper};
  constexpr int kDtype = sizeof(T) == 8 ? 1 : 0;
  auto exchange_on = [&](T* vec, hipStream_t st) -> int {
    if (rccl) return comm_rccl_exchange_segments(pc, vec, kDtype, msgs, st);
    peer_exchange_segments<T><<<2, 256, 0, st>>>(bp.pv, vec, to_upper, to_lower, from_lower, from_upper, ++pc->seq_ex, pc->err);
    return PISO_OK;
  };
  auto halo = [&](T* vec) -> int { return slab ? exchange_on(vec, stream) : PISO_OK; };
  // one stage of the scalar recurrences (peer transport / one GPU: one launch; RCCL: the ranks' sums, an all-reduce, the rest)
  // Arguments of the NEXT launch.  Two buffers of partial sums: a launch reads the one the last producer wrote and writes the other;
  // two scalar records: a launch with folded stages (folded_scalars) reads one and leaves the other as the current one.
  int cur = 0, pw = 0;
  auto next = [&](int fold, bool produces) -> BiArgs<T> {
    BiArgs<T> b = a;
    b.parts_in = pbuf0 + (size_t)pw * (2 * 4 * kBiParts);
    b.parts = pbuf0 + (size_t)(pw ^ 1) * (2 * 4 * kBiParts);
    b.fold = fold;
    b.sc_prev = scbuf0 + 2 * cur;
    if (fold) cur ^= 1;
    b.sc = scbuf0 + 2 * cur;
    if (produces) pw ^= 1;
    return b;
  };
  auto scalar = [&](int stage) -> int {
    const BiArgs<T> sa = next(0, false);
    if (slab && rccl) {
      BiPeer b = bp;
      b.on = 2; bi_scalar<T><<<2, kBlock, 0, stream>>>(sa, stage, b);
      { const int rc = comm_rccl_allreduce_f64(pc, bp.gsum, 8, stream); if (rc != PISO_OK) return rc; }
      b.on = 3; bi_scalar<T><<<2, kBlock, 0, stream>>>(sa, stage, b);
    } else {
      bi_scalar<T><<<2, kBlock, 0, stream>>>(sa, stage, next_seq());
    }
    return PISO_OK;
  };
  int dev_now = 0;
  if (slab) { PISO_HIP_CHECK(hipGetDevice(&dev_now)); if (dev_now < 0 || dev_now >= kMaxDevices) { set_error_msg("piso_multi_bicgstab_ilu_slab: device ordinal out of range"); return PISO_ERR_INVALID_ARG; } }
  SideStream& tl_side = tl_side_dev[dev_now];
  if (slab && !tl_side.stream) {
    PISO_HIP_CHECK(hipStreamCreateWithFlags(&tl_side.stream, hipStreamNonBlocking));
    PISO_HIP_CHECK(hipEventCreateWithFlags(&tl_side.ready, hipEventDisableTiming));
    PISO_HIP_CHECK(hipEventCreateWithFlags(&tl_side.halo, hipEventDisableTiming));
  }

  const int own0 = a.re[0] - a.rb[0], own1 = a.re[1] - a.rb[1];
  const int nmax = own0 > own1 ? own0 : own1;
  int gv = (nmax + kBlock * 4 - 1) / (kBlock * 4);
  gv = (gv + 7) & ~7;                                       // (a multiple of the XCD count: stencil_rows deals the rows by XCD)
  if (gv > kBiParts) gv = kBiParts;
  if (gv < 1) gv = 1;
  // slab mode: a product is two launches (interior, edge rows) that write the partial slots [0, gv) and [gv, gv + ge); every other
  // kernel runs gv + ge blocks so that it rewrites ALL slots the scalar kernels add up
  int ge = 0;
  if (slab) {
    ge = (2 * kEdgeRows * g.W[0] + kBlock * 4 - 1) / (kBlock * 4);
    if (gv + ge > kBiParts) gv = kBiParts - ge;
  }
  const dim3 grid_v(gv + ge, 2);
  a.nparts = gv + ge;
  const dim3 grid_vs(gv, 2), grid_e(ge > 0 ? ge : 1, 2);
  // y = B in with the edge rows of `in` fetched from the neighbours meanwhile (one GPU: one launch)
  auto spmv = [&](int which, T* in, T* out) -> int {
    const BiArgs<T> a = next(0, true);                        // (slab mode: both launches write the same buffer of partial sums)
    if (!slab) {
      if (which == 0) bi_spmv<T, 0><<<grid_v, kBlock, 0, stream>>>(a, in, out, 0, 0);
      else bi_spmv<T, 1><<<grid_v, kBlock, 0, stream>>>(a, in, out, 0, 0);
      return PISO_OK;
    }
    PISO_HIP_CHECK(hipEventRecord(tl_side.ready, stream));                       // `in` is complete on the owned rows
    PISO_HIP_CHECK(hipStreamWaitEvent(tl_side.stream, tl_side.ready, 0));
    { const int rc = exchange_on(in, tl_side.stream); if (rc != PISO_OK) return rc; }
    PISO_HIP_CHECK(hipEventRecord(tl_side.halo, tl_side.stream));
    if (which == 0) bi_spmv<T, 0><<<grid_vs, kBlock, 0, stream>>>(a, in, out, 1, 0);
    else bi_spmv<T, 1><<<grid_vs, kBlock, 0, stream>>>(a, in, out, 1, 0);
    PISO_HIP_CHECK(hipStreamWaitEvent(stream, tl_side.halo, 0));
    if (which == 0) bi_spmv<T, 0><<<grid_e, kBlock, 0, stream>>>(a, in, out, 2, gv);
    else bi_spmv<T, 1><<<grid_e, kBlock, 0, stream>>>(a, in, out, 2, gv);
    return PISO_OK;
  };
  const int nb0 = a.be[0] - a.bb[0], nb1 = a.be[1] - a.bb[1];
  const int nbmax = nb0 > nb1 ? nb0 : nb1;
  const dim3 grid_b(nbmax, 2);
  const int Wmax = nx + 1;
  const int need = (Wmax + kBlock - 1) / kBlock;
  if (need > 32) { set_error_msg("piso_multi_bicgstab_ilu: nx > 8191 not supported"); return PISO_ERR_INVALID_ARG; }

  bi_init_scalars<T><<<1, 256, 0, stream>>>(a);
  bi_convert<T><<<grid_v, kBlock, 0, stream>>>(a, val, rowptr, col, x0, transpose & 3);
  if (slab && rccl) { const int rc = comm_rccl_allreduce_i32(pc, a.flags, 2, stream); if (rc != PISO_OK) return rc; }   // (sums: non-zero = set)
  else if (slab) bi_flags_allreduce<T><<<1, 64, 0, stream>>>(a, next_seq());
  if (need <= 1) launch_factor<T, 1>(a, grid_b, stream);
  else if (need <= 2) launch_factor<T, 2>(a, grid_b, stream);
  else if (need <= 3) launch_factor<T, 3>(a, grid_b, stream);             // (W = nx + 1 with nx a power of two: 2^k / 256 + 1)
  else if (need <= 4) launch_factor<T, 4>(a, grid_b, stream);
  else if (need <= 5) launch_factor<T, 5>(a, grid_b, stream);
  else if (need <= 8) launch_factor<T, 8>(a, grid_b, stream);
  else if (need <= 9) launch_factor<T, 9>(a, grid_b, stream);
  else if (need <= 16) launch_factor<T, 16>(a, grid_b, stream);
  else launch_factor<T, 32>(a, grid_b, stream);
  PISO_LAUNCH_CHECK();

  auto precond = [&](const T* in, T* out, int fold, int fuse_p) {       // (fold: scalar stages the blocks of the forward sweep apply first)
    BiArgs<T> aL = next(fold, false);
    aL.fuse_p = fuse_p;
    const BiArgs<T> aU = next(0, false);
    if (need <= 1) launch_sweeps<T, 1>(aL, aU, grid_b, in, out, stream);
    else if (need <= 2) launch_sweeps<T, 2>(aL, aU, grid_b, in, out, stream);
    else if (need <= 3) launch_sweeps<T, 3>(aL, aU, grid_b, in, out, stream);
    else if (need <= 4) launch_sweeps<T, 4>(aL, aU, grid_b, in, out, stream);
    else if (need <= 5) launch_sweeps<T, 5>(aL, aU, grid_b, in, out, stream);
    else if (need <= 8) launch_sweeps<T, 8>(aL, aU, grid_b, in, out, stream);
    else if (need <= 9) launch_sweeps<T, 9>(aL, aU, grid_b, in, out, stream);
    else if (need <= 16) launch_sweeps<T, 16>(aL, aU, grid_b, in, out, stream);
    else launch_sweeps<T, 32>(aL, aU, grid_b, in, out, stream);
  };
  // Scalar stages folded into their consumers (folded_scalars) on one GPU: 14 -> 9 launches per iteration.  Every block of a vector
  // kernel re-reads the partial records in passing (<= 16 KB, L2-resident); at 2048^2 the five launches saved are worth 2.5 % of the
  // iteration (606 -> 590 us, round 5), more on smaller grids.  Option bicg_fold: 0 never.
  const bool fold_ok = !slab && opt(OPT_BICG_FOLD) != 0;
  auto F = [&](int stage) -> int { return fold_ok ? stage + 1 : 0; };
  // p = r + beta (p - omega v) inside the forward sweep of p_hat (BiArgs::fuse_p): 9 -> 8 launches per iteration.  Option bicg_fuse_p 0: never.
  const int fuse_p = opt(OPT_BICG_FUSE_P) != 0;

  BiHost<T> host;
  auto fetch = [&]() -> int {
    PISO_HIP_CHECK(hipMemcpyAsync(host.sc, scbuf0 + 2 * cur, 2 * sizeof(CompScalars<T>), hipMemcpyDeviceToHost, stream));
    PISO_HIP_CHECK(hipMemcpyAsync(host.flags, a.flags, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
    PISO_HIP_CHECK(hipStreamSynchronize(stream));
    return PISO_OK;
  };

  int failed_mask = 0;      // components that already used their one restart and failed again
  bool pattern_checked = false;
  for (int restart = 0; restart < 2; ++restart) {
    // r = b - B x, rh = r, p = v = 0, ||r|| test, first rho / beta
    { const int rc = halo(a.x); if (rc != PISO_OK) return rc; }
    bi_residual_init<T><<<grid_v, kBlock, 0, stream>>>(next(0, true));
    { const int rc = scalar(ST_INIT); if (rc != PISO_OK) return rc; }
    PISO_LAUNCH_CHECK();
    int it = 0, look = ntot < 32768 ? 1 : 2;                 // (tiny systems - the lid-driven cavity converges in one iteration: a second one is 13 launches for nothing)
    bool all_done = false;
    while (it < max_it && !all_done) {
      // iterations between host looks: 2, 2, 4, 8, 16, 16, ... - a solve of 3 iterations (the 2048^2 benchmark) still stops at
      // once, a solve of 100 (lid-driven cavity) synchronises 9 times instead of 50; launches of a converged component return early
      const int chunk = (max_it - it) < look ? (max_it - it) : look;
      if (look < 2) look = 2; else if (it >= 4 && look < 16) look *= 2;
      for (int q = 0; q < chunk; ++q, ++it) {
        // (folded: the ||r|| test that ended the iteration before - inside a chunk nobody else has applied it yet - then rho / beta)
        int fold_p = 0;
        if (it > 0) {
          if (fold_ok) fold_p = q > 0 ? (F(ST_CHECK_R) | (F(ST_RHO_BETA) << 4)) : F(ST_RHO_BETA);
          else { const int rc = scalar(ST_RHO_BETA); if (rc != PISO_OK) return rc; }
        }
        if (fuse_p) precond(a.p, a.ph, fold_p, 1);
        else {
          bi_update_p<T><<<grid_v, kBlock, 0, stream>>>(next(fold_p, false));
          precond(a.p, a.ph, 0, 0);
        }
        { const int rc = spmv(0, a.ph, a.v); if (rc != PISO_OK) return rc; }
        if (!fold_ok) { const int rc = scalar(ST_ALPHA); if (rc != PISO_OK) return rc; }
        bi_update_xr<T, 0><<<grid_v, kBlock, 0, stream>>>(next(F(ST_ALPHA), true));
        if (!fold_ok) { const int rc = scalar(ST_CHECK_S); if (rc != PISO_OK) return rc; }
        precond(a.r, a.sh, F(ST_CHECK_S), 0);
        { const int rc = spmv(1, a.sh, a.t); if (rc != PISO_OK) return rc; }
        if (!fold_ok) { const int rc = scalar(ST_OMEGA); if (rc != PISO_OK) return rc; }
        bi_update_xr<T, 1><<<grid_v, kBlock, 0, stream>>>(next(F(ST_OMEGA), true));
        if (!fold_ok || q == chunk - 1) { const int rc = scalar(ST_CHECK_R); if (rc != PISO_OK) return rc; }   // (the host looks at it)
      }
      PISO_LAUNCH_CHECK();
      { const int rc = fetch(); if (rc != PISO_OK) return rc; }
      if (!pattern_checked) {
        pattern_checked = true;
        if (host.flags[0]) {
          set_error_msg("piso_multi_bicgstab_ilu: CSR input is not a 5-point staggered-grid matrix");
          return PISO_ERR_UNSUPPORTED_PATTERN;
        }
      }
      all_done = host.sc[0].done && host.sc[1].done;
    }
    if (max_it == 0 || !pattern_checked) { const int rc = fetch(); if (rc != PISO_OK) return rc; }
    // failure test per component (:392-407): ||r|| > 100 tol or NaN -> x = 0 and one more pass from x = 0
    int fail_now = 0;
    for (int c = 0; c < 2; ++c) {
      const T nrm = host.sc[c].nrm;
      if (nrm > (T)tol * 100 || nrm != nrm) fail_now |= 1 << c;
    }
    if (!fail_now) break;
    bi_zero_x<T><<<grid_v, kBlock, 0, stream>>>(next(0, false), fail_now);
    if (restart == 0) {
      bi_set_done<T><<<1, 64, 0, stream>>>(next(0, false), !(fail_now & 1), !((fail_now >> 1) & 1));
    } else {
      failed_mask = fail_now;
    }
  }
  (void)failed_mask;
  if (slab && !rccl) {
    int herr = 0;
    peer_agree_on_error<><<<1, 64, 0, stream>>>(bp.pv, pc->err, ++pc->seq_ar);      // every rank returns the same status
    PISO_HIP_CHECK(hipMemcpyAsync(&herr, pc->err, sizeof(int), hipMemcpyDeviceToHost, stream));
    PISO_HIP_CHECK(hipStreamSynchronize(stream));
    if (herr) {
      PISO_HIP_CHECK(hipMemsetAsync(pc->err, 0, sizeof(int), stream));
      set_error_msg("piso_multi_bicgstab_ilu_slab: a wait on a peer's mailbox gave up (peer process gone or not running?)");
      return PISO_ERR_HIP;
    }
  }
  PISO_HIP_CHECK(hipStreamSynchronize(stream));
  if (host.flags[1] && warning) {
    const uint8_t one = 1;
    PISO_HIP_CHECK(hipMemcpyAsync(warning, &one, 1, hipMemcpyHostToDevice, stream));
    PISO_HIP_CHECK(hipStreamSynchronize(stream));
  }
  if (iterations_out) { iterations_out[0] = host.sc[0].it_count; iterations_out[1] = host.sc[1].it_count; }
  return PISO_OK;
}

// y = A x or A^T x on the concatenated CSR (second-corrector H product and its adjoint)
__global__ __launch_bounds__(kBlock) void csr_matvec_kernel(const float* __restrict__ val_all, const int* __restrict__ rp_all,
                                                            const int* __restrict__ col_all, const float* __restrict__ x,
                                                            float* __restrict__ y, Geo g, int transpose, int lo0, int hi0, int lo1, int hi1, RowMap M) {
  const int c = blockIdx.y;
  const int n = g.n[c], W = g.W[c];
  const int row_lo = c ? lo1 : lo0, row_hi = c ? hi1 : hi0;     // (slab-decomposed step: this rank's face rows)
  const int nu_st = M.on ? M.n_u : g.n[0];
  const int* rp = rp_all + (c ? nu_st + 1 : 0);
  const int k0 = c ? rp_all[nu_st] : 0;
  const float* val = val_all + k0;
  const int* col = col_all + k0;
  auto kx = [&](int row) { return M.on ? (c ? M.n_u : 0) + M.frow(c, row) : g.r0[c] + row; };
  for (int row = row_lo + blockIdx.x * kBlock + threadIdx.x; row < row_hi; row += gridDim.x * kBlock) {
    float acc = 0.f;
    if (!transpose) {
      const int rl = M.frow(c, row);
      for (int q = rp[rl]; q < rp[rl + 1]; ++q) acc = fmaf(val[q], x[kx(col[q])], acc);
    } else {
      // gather over every row that can hold an entry in column `row` (ascending row order = csr2csc order)
      const int cand[9] = {row - g.yw[c], row - W, row - g.xw[c], row - 1, row, row + 1, row + g.xw[c], row + W, row + g.yw[c]};
      int last = -1;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int m = cand[q];
        if (m < 0 || m >= n || m <= last) continue;
        last = m;
        float v;
        if (csr_find<float>(rp, col, val, m, row, &v, M, c)) acc = fmaf(v, x[kx(m)], acc);
      }
    }
    y[kx(row)] = acc;
  }
}

}  // namespace piso

using namespace piso;

extern "C" {

size_t piso_bicgstab_workspace_bytes(int nx, int ny, int elem_size) {
  return elem_size == 8 ? bi_workspace_bytes<double>(nx, ny) : bi_workspace_bytes<float>(nx, ny);
}

int piso_multi_bicgstab_ilu_f32(const float* csr_val, const int* csr_rowptr, const int* csr_col, const float* rhs,
                                const float* x0, float* x_out, int nx, int ny, float tol, int max_it, int transpose,
                                int band_rows, uint8_t* warning, int* iterations_out, void* workspace,
                                size_t workspace_bytes, piso_stream_t stream) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  return bi_solve<float>(csr_val, csr_rowptr, csr_col, rhs, x0, x_out, nx, ny, tol, max_it, transpose, band_rows, warning,
                         iterations_out, workspace, workspace_bytes, stream);
}

int piso_multi_bicgstab_ilu_f64(const double* csr_val, const int* csr_rowptr, const int* csr_col, const double* rhs,
                                const double* x0, double* x_out, int nx, int ny, float tol, int max_it, int transpose,
                                int band_rows, uint8_t* warning, int* iterations_out, void* workspace,
                                size_t workspace_bytes, piso_stream_t stream) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  return bi_solve<double>(csr_val, csr_rowptr, csr_col, rhs, x0, x_out, nx, ny, tol, max_it, transpose, band_rows, warning,
                          iterations_out, workspace, workspace_bytes, stream);
}

int piso_multi_bicgstab_ilu_slab_f32(void* comm, const float* csr_val, const int* csr_rowptr, const int* csr_col, const float* rhs,
                                     const float* x0, float* x_out, int nx, int ny, float tol, int max_it, int transpose,
                                     int band_rows, uint8_t* warning, int* iterations_out, void* workspace, size_t workspace_bytes,
                                     piso_stream_t stream) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  if (!comm) { set_error_msg("piso_multi_bicgstab_ilu_slab_f32: NULL communicator"); return PISO_ERR_INVALID_ARG; }
  return bi_solve<float>(csr_val, csr_rowptr, csr_col, rhs, x0, x_out, nx, ny, tol, max_it, transpose, band_rows, warning, iterations_out,
                         workspace, workspace_bytes, stream, static_cast<PisoComm*>(comm));
}

int piso_multi_bicgstab_ilu_slab_f64(void* comm, const double* csr_val, const int* csr_rowptr, const int* csr_col, const double* rhs,
                                     const double* x0, double* x_out, int nx, int ny, float tol, int max_it, int transpose,
                                     int band_rows, uint8_t* warning, int* iterations_out, void* workspace, size_t workspace_bytes,
                                     piso_stream_t stream) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  if (!comm) { set_error_msg("piso_multi_bicgstab_ilu_slab_f64: NULL communicator"); return PISO_ERR_INVALID_ARG; }
  return bi_solve<double>(csr_val, csr_rowptr, csr_col, rhs, x0, x_out, nx, ny, tol, max_it, transpose, band_rows, warning, iterations_out,
                          workspace, workspace_bytes, stream, static_cast<PisoComm*>(comm));
}

size_t piso_bicgstab_slab_workspace_bytes(int nx, int ny, int elem_size, const piso_slab_t* slab) {
  if (!slab_ok(slab, ny)) return 0;
  return elem_size == 8 ? bi_workspace_bytes<double>(nx, ny, slab) : bi_workspace_bytes<float>(nx, ny, slab);
}
int piso_multi_bicgstab_ilu_slab_local_f32(void* comm, const float* csr_val, const int* csr_rowptr, const int* csr_col, const float* rhs,
                                           const float* x0, float* x_out, int nx, int ny, int periodic_x, int periodic_y, float tol, int max_it,
                                           int transpose, int band_rows, uint8_t* warning, int* iterations_out, void* workspace,
                                           size_t workspace_bytes, piso_stream_t stream, const piso_slab_t* slab) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  if (!comm || !slab) { set_error_msg("piso_multi_bicgstab_ilu_slab_local_f32: NULL communicator / slab"); return PISO_ERR_INVALID_ARG; }
  return bi_solve<float>(csr_val, csr_rowptr, csr_col, rhs, x0, x_out, nx, ny, tol, max_it, transpose, band_rows, warning, iterations_out,
                         workspace, workspace_bytes, stream, static_cast<PisoComm*>(comm), slab, periodic_x ? 1 : 0, periodic_y ? 1 : 0);
}
int piso_multi_bicgstab_ilu_slab_local_f64(void* comm, const double* csr_val, const int* csr_rowptr, const int* csr_col, const double* rhs,
                                           const double* x0, double* x_out, int nx, int ny, int periodic_x, int periodic_y, float tol, int max_it,
                                           int transpose, int band_rows, uint8_t* warning, int* iterations_out, void* workspace,
                                           size_t workspace_bytes, piso_stream_t stream, const piso_slab_t* slab) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  if (!comm || !slab) { set_error_msg("piso_multi_bicgstab_ilu_slab_local_f64: NULL communicator / slab"); return PISO_ERR_INVALID_ARG; }
  return bi_solve<double>(csr_val, csr_rowptr, csr_col, rhs, x0, x_out, nx, ny, tol, max_it, transpose, band_rows, warning, iterations_out,
                          workspace, workspace_bytes, stream, static_cast<PisoComm*>(comm), slab, periodic_x ? 1 : 0, periodic_y ? 1 : 0);
}

static int csr_matvec_impl(const float* csr_val, const int* csr_rowptr, const int* csr_col, const float* x, float* y,
                           int nx, int ny, int periodic_x, int periodic_y, int transpose, piso_stream_t stream, const piso_slab_t* slab) {
  if (nx < 4 || ny < 4 || !csr_val || !csr_rowptr || !csr_col || !x || !y || !slab_ok(slab, ny)) {
    set_error_msg("piso_csr_matvec_f32: invalid argument");
    return PISO_ERR_INVALID_ARG;
  }
  const Geo g = make_geo(nx, ny, 8);
  const RowMap M = make_row_map(slab, nx, ny, periodic_x ? 1 : 0, periodic_y ? 1 : 0);
  const FaceWin fw = face_window(M);                       // component-local row ranges (whole components on one GPU)
  const int lo0 = fw.u_lo, hi0 = fw.u_lo + fw.cu, lo1 = fw.v_lo - g.n[0], hi1 = lo1 + fw.cv;
  const int nmax = fw.cu > fw.cv ? fw.cu : fw.cv;
  int gv = (nmax + kBlock * 2 - 1) / (kBlock * 2);
  if (gv > 4096) gv = 4096;
  if (gv < 1) gv = 1;
  csr_matvec_kernel<<<dim3(gv, 2), kBlock, 0, static_cast<hipStream_t>(stream)>>>(csr_val, csr_rowptr, csr_col, x, y, g,
                                                                                  transpose ? 1 : 0, lo0, hi0, lo1, hi1, M);
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}
int piso_csr_matvec_f32(const float* csr_val, const int* csr_rowptr, const int* csr_col, const float* x, float* y,
                        int nx, int ny, int transpose, piso_stream_t stream) {
  return csr_matvec_impl(csr_val, csr_rowptr, csr_col, x, y, nx, ny, 0, 0, transpose, stream, nullptr);
}
int piso_csr_matvec_f32_slab(const float* csr_val, const int* csr_rowptr, const int* csr_col, const float* x, float* y,
                             int nx, int ny, int periodic_x, int periodic_y, int transpose, piso_stream_t stream, const piso_slab_t* slab) {
  return csr_matvec_impl(csr_val, csr_rowptr, csr_col, x, y, nx, ny, periodic_x, periodic_y, transpose, stream, slab);
}
}
