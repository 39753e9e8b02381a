// Pressure-Poisson CG for MI355X (gfx950): the dominant kernel pair of the PISO step.
//
// Replaces LaunchPressureKernel + calcZ_v4 / initVariablesWithGuess / checkResiduum and the cuBLAS BLAS-1 calls of
// CUDAsrc/pressure_solve_op.cu.cc:57-418 (double) / :420-696 (float).  Same iteration, same restart / stopping rules,
// re-cut for HBM: one CG iteration is TWO launches and no host round trip,
//
//   K1  p <- r + beta p   (recomputed on the fly for the 5 stencil points, never a separate pass)
//       z' <- L p         (5-point variable-coefficient stencil, SoA coefficients, register sliding window over rows)
//       partials: sum(p), p.r, p.z'
//   K2  alpha from K1's partials;  r <- r - alpha (z' + c sum(p));  partials: r.z', sum(r), #cells with |r| >= accuracy
//       (x <- x + alpha p is done by the NEXT K1, which reads p anyway)
//
// The rank-1 shift c*sum(p) of the reference (z = L p + c sum p, pressure_solve_op.cu.cc:277-286) is carried as a scalar:
// p.z = p.z' + c sum(p)^2, r.z = r.z' + c sum(p) sum(r).  Per-block partial sums are reduced in a fixed order by every
// block of the NEXT kernel (deterministic, L2-served) -- scalars never visit the host.  The stopping test
// (every 5th iteration, max|r| < accuracy, with the reference's flag semantics) is evaluated on the device; the host only
// polls a 16-byte state record per batch of iterations.
//
// Traffic per cell and iteration with fp64 vectors: K1 reads r, p, 5 coefficients, writes p, z' (9 words), K2 reads
// p, z', x, r, writes x, r (6 words) = 120 B against the 128 B "algorithmic" figure of SURVEY.md 8(d).
// (Later in round 1: exact float32 off-diagonals, diagonal recomputed on the fly, x update moved into K1 -- see DESIGN.md.)
#pragma once
#include <stdlib.h>

#include "piso_common.h"

namespace piso {

// state record, double-buffered by version parity (a kernel reads version v and, if it changes it, writes v + 1)
struct CgState {
  int flag;        // the reference's device-side threshold_reached
  int done;        // the reference's threshold_reached_cpu after a successful test
  int iterations;  // what the reference writes to iterations_gpu when it stops early
  int pad;
};

enum { MODE_NORMAL = 0, MODE_INIT = 1, MODE_RESET = 2 };
enum { SC_C = 0, SC_PZ = 1, SC_VS = 2, SC_ALPHA = 3, SC_COUNT = 8 };

template <typename T>
struct CgArgs {
  const T* cC;                       // SoA stencil coefficients: diagonal in T,
  const void *oS, *oW, *oE, *oN;     //   off-diagonals in the kernel's CT (float when exactly representable, else T)
  const T* b;
  T *x, *r, *z;
  T* p[2];                           // ping-pong search direction
  T* zp[2];                          // cg_persist1: ping-pong buffers of the published z' perimeters (agent-scope accesses ONLY)
  T* partsA;                         // K1 partials  [3][kMaxPartials]: sum p, p.r, p.z'
  T* partsB;                         // K2 partials  [3][kMaxPartials]: r.z', sum r, max|r|
  T* partsS;                         // setup partials [kMaxPartials]: sum |diag|
  T* scal;                           // SC_* scalars
  CgState* state;                    // [2]
  int nx, ny, per_x, per_y;          // per_y: 0 none, 1 wrap, 2 halo rows stored at row -1 / ny of r, p[], x (slab mode)
  const T* gA;                       // slab mode: all-reduced K1 sums [3] (NULL: reduce partsA)
  const T* gB;                       // slab mode: all-reduced K2 sums [3] (NULL: reduce partsB)
  int ntx, nty, rows_per_wave;
  int nA, nB;                        // blocks (= partial records) of K1 / K2
  float accuracy;
  int nt;                            // tuning bits: 1 K2 stores, 2 K2 loads, 4 K1 stores, 8 K1 loads non-temporal
  // Padded-grid mode (cg.hip): a wall-bounded nx_true x ny_true system embedded in the nx x ny grid the persistent kernel can
  // tile.  The padding has zero coefficients and a zero right-hand side, so it stays zero by itself - except under the rank-1
  // shift c sum(p), which the reference adds to EVERY cell: the kernels that apply it skip the padding.  0: not padded.
  int nx_true, ny_true;
  double ncells;                     // cells the shift constant and the sum(r) recurrence count (0: nx * ny)
};
// is flat cell i of the (possibly padded) grid a cell of the true system?
template <typename T>
__device__ __forceinline__ bool true_cell(const CgArgs<T>& a, size_t i) {
  if (!a.nx_true) return true;
  return (int)(i % (size_t)a.nx) < a.nx_true && (int)(i / (size_t)a.nx) < a.ny_true;
}

template <typename T, int V>
struct Vec {
  T v[V];
};

// V elements of S from / to an address aligned to min(16, sizeof(S) * V) bytes; 16-byte lane accesses wherever possible
template <typename S, int V>
__device__ __forceinline__ Vec<S, V> ldc(const S* __restrict__ p) {
  Vec<S, V> o;
  constexpr int B = sizeof(S) * V;
  using raw4 = __attribute__((ext_vector_type(4))) unsigned int;
  using raw2 = __attribute__((ext_vector_type(2))) unsigned int;
  if constexpr (B % 16 == 0) {
#pragma unroll
    for (int q = 0; q < B / 16; ++q) {
      const raw4 t = reinterpret_cast<const raw4*>(p)[q];
      __builtin_memcpy(reinterpret_cast<char*>(&o) + 16 * q, &t, 16);
    }
  } else if constexpr (B == 8) {
    const raw2 t = *reinterpret_cast<const raw2*>(p);
    __builtin_memcpy(&o, &t, 8);
  } else {
    static_assert(V == 1, "unsupported vector width");
    o.v[0] = p[0];
  }
  return o;
}
template <typename T, int V>
__device__ __forceinline__ Vec<T, V> ldv(const T* __restrict__ p) { return ldc<T, V>(p); }

// non-temporal (streaming) flavours: the data is not re-used before it leaves the caches anyway
template <typename S, int V>
__device__ __forceinline__ Vec<S, V> ldc_nt(const S* __restrict__ p) {
  Vec<S, V> o;
  constexpr int B = sizeof(S) * V;
  using raw4 = __attribute__((ext_vector_type(4))) unsigned int;
  using raw2 = __attribute__((ext_vector_type(2))) unsigned int;
  if constexpr (B % 16 == 0) {
#pragma unroll
    for (int q = 0; q < B / 16; ++q) {
      const raw4 t = __builtin_nontemporal_load(reinterpret_cast<const raw4*>(p) + q);
      __builtin_memcpy(reinterpret_cast<char*>(&o) + 16 * q, &t, 16);
    }
  } else if constexpr (B == 8) {
    const raw2 t = __builtin_nontemporal_load(reinterpret_cast<const raw2*>(p));
    __builtin_memcpy(&o, &t, 8);
  } else {
    o.v[0] = __builtin_nontemporal_load(p);
  }
  return o;
}
template <typename T, int V>
__device__ __forceinline__ void stv_nt(T* __restrict__ p, const Vec<T, V>& o) {
  constexpr int B = sizeof(T) * V;
  using raw4 = __attribute__((ext_vector_type(4))) unsigned int;
  using raw2 = __attribute__((ext_vector_type(2))) unsigned int;
  if constexpr (B % 16 == 0) {
#pragma unroll
    for (int q = 0; q < B / 16; ++q) {
      raw4 t;
      __builtin_memcpy(&t, reinterpret_cast<const char*>(&o) + 16 * q, 16);
      __builtin_nontemporal_store(t, reinterpret_cast<raw4*>(p) + q);
    }
  } else if constexpr (B == 8) {
    raw2 t;
    __builtin_memcpy(&t, &o, 8);
    __builtin_nontemporal_store(t, reinterpret_cast<raw2*>(p));
  } else {
    __builtin_nontemporal_store(o.v[0], p);
  }
}
template <typename T, int V>
__device__ __forceinline__ void stv(T* __restrict__ p, const Vec<T, V>& o) {
  constexpr int B = sizeof(T) * V;
  using raw4 = __attribute__((ext_vector_type(4))) unsigned int;
  using raw2 = __attribute__((ext_vector_type(2))) unsigned int;
  if constexpr (B % 16 == 0) {
#pragma unroll
    for (int q = 0; q < B / 16; ++q) {
      raw4 t;
      __builtin_memcpy(&t, reinterpret_cast<const char*>(&o) + 16 * q, 16);
      reinterpret_cast<raw4*>(p)[q] = t;
    }
  } else if constexpr (B == 8) {
    raw2 t;
    __builtin_memcpy(&t, &o, 8);
    *reinterpret_cast<raw2*>(p) = t;
  } else {
    p[0] = o.v[0];
  }
}

template <typename T>
__device__ __forceinline__ T absval(T v) { return v < 0 ? -v : v; }

// ---------------------------------------------------------------------------------------------------------------
// Prologue shared by K1 / K2: every block reduces the previous kernel's partial records.  Only wave 0 does the loads,
// the other waves go straight to their first data loads and meet wave 0 at the barrier, so the L2 round trip of the
// reduction overlaps the first HBM loads of the body.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int NV>
__device__ __forceinline__ void prologue_reduce(const T* __restrict__ parts, int count, const T* __restrict__ global,
                                                T (&sum)[NV], T* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave == 0) {
    T s[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) s[q] = 0;
    if (global) {                                   // slab mode: the sums were all-reduced across ranks already
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) s[q] = global[q];
      }
    } else {
      for (int b = lane; b < count; b += 64) {
#pragma unroll
        for (int q = 0; q < NV; ++q) s[q] += parts[q * kMaxPartials + b];
      }
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) s[q] = wave_sum(s[q]);
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < NV; ++q) smem[q] = s[q];
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NV; ++q) sum[q] = smem[q];
  __syncthreads();          // smem is reused by the epilogue reduction
}

// ---------------------------------------------------------------------------------------------------------------
// K1: fused direction update + stencil + dots (+ the x update of the PREVIOUS iteration, whose direction it reads anyway).
// One wave owns a strip of 64*V columns and walks `rows_per_wave` rows keeping three rows of the NEW direction in
// registers; x-neighbours come from lane shuffles, strip-edge columns from two extra scalar loads.  Tiles are dealt to
// blocks XCD-contiguously so halo rows hit the same L2; odd waves walk their rows downwards so that the row shared with
// the neighbouring wave is touched by both at about the same time.
//   mode NORMAL: p_new = r + beta p_old (beta from K2's partials), x += alpha_prev p_old   (pressure_solve_op.cu.cc:302-303, 345-354)
//        INIT  : p_new = r                                                                  (initVariablesWithGuess, :104-114)
//        RESET : apply the operator to x (z' = L x, partial sum(x)); x was flushed by the host beforehand   (:260-274)
//   do_check : this launch is the first of iteration k and evaluates the stopping test of iteration k-1 (:312-335)
//   pending  : a direction of the previous iteration still has to be added to x (false right after INIT / RESET)
// CT = storage type of the off-diagonals; RECON = the diagonal is recomputed as -(S + N + W + E) (setup verified that this
// reproduces the stored fp64 diagonal bit for bit) instead of being read.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, typename CT, int V, bool RECON>
__global__ __launch_bounds__(kBlock) void cg_k1(CgArgs<T> a, int k, int mode, int sv, int do_check, int pending) {
  __shared__ T smem[16];
  const bool writes_state = do_check || mode == MODE_RESET;
  const int nx = a.nx, ny = a.ny;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* __restrict__ pin = a.p[k & 1];
  T* __restrict__ pout = a.p[(k + 1) & 1];
  const T* __restrict__ src = (mode == MODE_RESET) ? a.x : a.r;
  const bool use_pin = (mode == MODE_NORMAL);      // RESET reads a fully updated x (the host flushes it first)

  // raw loads of the own columns of row j (rows outside the domain wrap or read as zero); combined later with beta
  struct Raw { Vec<T, V> s, q; };
  auto load_raw = [&](int j, int c0, bool active) -> Raw {
    Raw o;
#pragma unroll
    for (int e = 0; e < V; ++e) { o.s.v[e] = 0; o.q.v[e] = 0; }
    if (!active) return o;
    if (a.per_y != 2) {                             // slab mode: rows -1 and ny are real halo rows in memory
      if (j < 0) { if (!a.per_y) return o; j = ny - 1; }
      if (j >= ny) { if (!a.per_y) return o; j = 0; }
    }
    const ptrdiff_t i = (ptrdiff_t)j * nx + c0;
    o.s = ldv<T, V>(src + i);
    if (use_pin) o.q = ldv<T, V>(pin + i);
    return o;
  };

  // ---- everything that does not depend on the scalars is issued first: state, scalars, the first rows of my first tile
  const CgState st = a.state[sv & 1];
  const T sc_pz = a.scal[SC_PZ], sc_vs = a.scal[SC_VS], sc_alpha = a.scal[SC_ALPHA];
  const int ntiles = a.ntx * a.nty;
  const XcdRange tr = xcd_range(ntiles);
  const int dir = (wave & 1) ? -1 : 1;
  auto tile_geom = [&](int t, int& jb, int& je, int& c0, bool& active) {
    const int ty = t / a.ntx, tx = t - ty * a.ntx;
    jb = (ty * 4 + wave) * a.rows_per_wave;
    je = min(jb + a.rows_per_wave, ny);
    c0 = (tx * 64 + lane) * V;
    active = c0 < nx;                                 // nx % V == 0 => all V columns valid together
  };
  Raw rawB, rawC;
  {
    int jb, je, c0; bool active;
    if (tr.begin < tr.end) {
      tile_geom(tr.begin, jb, je, c0, active);
      if (jb < ny) {
        const int j = (dir > 0) ? jb : je - 1;
        rawB = load_raw(j - dir, c0, active);
        rawC = load_raw(j, c0, active);
      }
    }
  }

  T beta = 0;
  CgState nst = st;
  if (do_check) {                                     // uniform: every block reduces, also when already done (cheap, rare)
    T pb[3];
    prologue_reduce<T, 3>(a.partsB, a.nB, a.gB, pb, smem);               // r.z', sum r, #cells with !(|r| < accuracy)
    if (!st.done) {
      if (k > 0 && (k % 5) == 0) {
        const int exceeded = pb[2] > 0;              // checkResiduum (:94-102) clears the flag if any |r| >= accuracy
        if (st.flag && !exceeded) { nst.done = 1; nst.iterations = k; }
        else nst.flag = 1;                           // cudaMemset(threshold_reached, 1) after a failed test (:334)
      }
      if (mode == MODE_NORMAL) beta = -(pb[0] + sc_vs * pb[1]) / sc_pz;   // -r.z / p.z, unguarded as coded (:351-352)
    }
  }
  if (mode == MODE_RESET && !nst.done) nst.flag = 0; // initVariablesWithGuess clears the device flag
  // the state always moves to the next version slot, also once done (later launches read that slot)
  if (writes_state && blockIdx.x == 0 && threadIdx.x == 0) a.state[(sv + 1) & 1] = nst;
  if (st.done) return;
  const T alpha_prev = pending ? sc_alpha : (T)0;
  const bool only_flush = nst.done;                  // converged: just add the last direction to x, then stop

  auto combine = [&](const Raw& w) -> Vec<T, V> {
    Vec<T, V> o = w.s;
    if (use_pin) {
#pragma unroll
      for (int e = 0; e < V; ++e) o.v[e] = fma(beta, w.q.v[e], o.v[e]);
    }
    return o;
  };
  auto val_at = [&](int j, int c) -> T {             // value at (row j, column c); rows outside wrap or read as zero
    if (a.per_y != 2) {
      if (j < 0) { if (!a.per_y) return (T)0; j = ny - 1; }
      if (j >= ny) { if (!a.per_y) return (T)0; j = 0; }
    }
    const ptrdiff_t i = (ptrdiff_t)j * nx + c;
    T v = src[i];
    if (use_pin) v = fma(beta, pin[i], v);
    return v;
  };

  T acc_p = 0, acc_pr = 0, acc_pz = 0;
  for (int t = tr.begin; t < tr.end; t += tr.step) {
    int jb, je, c0; bool active;
    tile_geom(t, jb, je, c0, active);
    if (jb >= ny) continue;
    if (only_flush) {                                     // x += alpha_prev * p_old on my cells, nothing else
      if (active && pending && use_pin)
        for (int j = jb; j < je; ++j) {
          const size_t i = (size_t)j * nx + c0;
          Vec<T, V> xv = ldv<T, V>(a.x + i);
          const Vec<T, V> q = ldv<T, V>(pin + i);
#pragma unroll
          for (int e = 0; e < V; ++e) xv.v[e] = fma(alpha_prev, q.v[e], xv.v[e]);
          stv<T, V>(a.x + i, xv);
        }
      continue;
    }
    int j = (dir > 0) ? jb : je - 1;
    if (t != tr.begin) {
      rawB = load_raw(j - dir, c0, active);
      rawC = load_raw(j, c0, active);
    }
    Vec<T, V> behind = combine(rawB);
    Vec<T, V> cur = combine(rawC);
    Vec<T, V> praw = rawC.q;
    if (a.per_y == 2 && active && mode != MODE_RESET && (j - dir < 0 || j - dir >= ny))
      stv<T, V>(pout + (ptrdiff_t)(j - dir) * nx + c0, behind);
    for (int cnt = jb; cnt < je; ++cnt, j += dir) {
      const Raw rawA = load_raw(j + dir, c0, active);
      const Vec<T, V> ahead = combine(rawA);
      if (a.per_y == 2 && active && mode != MODE_RESET && (j + dir < 0 || j + dir >= ny))
        stv<T, V>(pout + (ptrdiff_t)(j + dir) * nx + c0, ahead);
      T left = __shfl_up(cur.v[V - 1], 1, kWave);
      T right = __shfl_down(cur.v[0], 1, kWave);
      if (active) {
        if (lane == 0) {
          const int c = c0 - 1;
          left = (c >= 0) ? val_at(j, c) : (a.per_x ? val_at(j, nx - 1) : (T)0);
        }
        if (lane == 63 || c0 + V >= nx) {
          const int c = c0 + V;
          right = (c < nx) ? val_at(j, c) : (a.per_x ? val_at(j, 0) : (T)0);
        }
        const size_t i = (size_t)j * nx + c0;
        Vec<CT, V> kS, kW, kE, kN;
        if (a.nt & 8) {
          kS = ldc_nt<CT, V>(static_cast<const CT*>(a.oS) + i); kW = ldc_nt<CT, V>(static_cast<const CT*>(a.oW) + i);
          kE = ldc_nt<CT, V>(static_cast<const CT*>(a.oE) + i); kN = ldc_nt<CT, V>(static_cast<const CT*>(a.oN) + i);
        } else {
          kS = ldc<CT, V>(static_cast<const CT*>(a.oS) + i); kW = ldc<CT, V>(static_cast<const CT*>(a.oW) + i);
          kE = ldc<CT, V>(static_cast<const CT*>(a.oE) + i); kN = ldc<CT, V>(static_cast<const CT*>(a.oN) + i);
        }
        Vec<T, V> kC;
        if constexpr (RECON) {
#pragma unroll
          for (int e = 0; e < V; ++e) {                   // accumulation order of calcPISOLaplaceMatrix (laplace_op.cu.cc:118-135)
            T d = 0;
            d -= (T)kS.v[e]; d -= (T)kN.v[e]; d -= (T)kW.v[e]; d -= (T)kE.v[e];
            kC.v[e] = d;
          }
        } else {
          kC = ldv<T, V>(a.cC + i);
        }
        Vec<T, V> rr = cur;
        if (mode == MODE_NORMAL) rr = rawC.s;               // the residual of my cells (INIT: p == r; RESET: unused)
        if (pending && use_pin) {                           // x <- x + alpha_prev p_old  (cublasDaxpy :303 of iteration k-1)
          Vec<T, V> xv = ldv<T, V>(a.x + i);
#pragma unroll
          for (int e = 0; e < V; ++e) xv.v[e] = fma(alpha_prev, praw.v[e], xv.v[e]);
          stv<T, V>(a.x + i, xv);
        }
        Vec<T, V> zz;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const T pw = (e == 0) ? left : cur.v[e > 0 ? e - 1 : 0];
          const T pe = (e == V - 1) ? right : cur.v[e < V - 1 ? e + 1 : 0];
          const T ps = (dir > 0) ? behind.v[e] : ahead.v[e];
          const T pnn = (dir > 0) ? ahead.v[e] : behind.v[e];
          // summation order of calcZ_v4 (:81-90): -y, -x, diag, +x, +y
          T tmp = 0;
          tmp = fma((T)kS.v[e], ps, tmp);
          tmp = fma((T)kW.v[e], pw, tmp);
          tmp = fma(kC.v[e], cur.v[e], tmp);
          tmp = fma((T)kE.v[e], pe, tmp);
          tmp = fma((T)kN.v[e], pnn, tmp);
          zz.v[e] = tmp;
          acc_p += cur.v[e];
          acc_pr = fma(cur.v[e], rr.v[e], acc_pr);
          acc_pz = fma(cur.v[e], tmp, acc_pz);
        }
        if (a.nt & 4) { stv_nt<T, V>(a.z + i, zz); if (mode != MODE_RESET) stv_nt<T, V>(pout + i, cur); }
        else { stv<T, V>(a.z + i, zz); if (mode != MODE_RESET) stv<T, V>(pout + i, cur); }
      }
      behind = cur;
      cur = ahead;
      praw = rawA.q;
      rawC = rawA;
    }
  }
  if (only_flush) return;
  T part[3] = {acc_p, acc_pr, acc_pz};
  block_sum<T, 3>(part, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q) a.partsA[q * kMaxPartials + blockIdx.x] = part[q];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K2: alpha, residual update, dots for beta and the stopping test.  Pure stream over flat cells, XCD-chunked like K1.
// (x is updated by the next K1, which reads the direction anyway.)
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ __launch_bounds__(kBlock) void cg_k2(CgArgs<T> a, int k, int sv) {
  __shared__ T smem[16];
  constexpr int D = 4;                                    // chunks kept in flight per thread
  const T* __restrict__ zp = a.z;
  T* __restrict__ rp = a.r;
  const size_t n = (size_t)a.nx * a.ny;
  const int nchunks = (int)((n / V + kBlock - 1) / kBlock);   // chunks of kBlock * V cells
  const XcdRange cr = xcd_range(nchunks);
  // issued before anything depends on the scalars: state, shift, the first D chunks
  const CgState st = a.state[sv & 1];
  const T sc_c = a.scal[SC_C];
  Vec<T, V> zq[D], rq[D];
  size_t iq[D];
  bool okq[D];
  int ch = cr.begin;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int c = ch + d * cr.step;
    iq[d] = ((size_t)c * kBlock + threadIdx.x) * V;
    okq[d] = c < cr.end && iq[d] < n;
    if (okq[d]) {
      if (a.nt & 2) { zq[d] = ldc_nt<T, V>(zp + iq[d]); rq[d] = ldc_nt<T, V>(rp + iq[d]); }
      else { zq[d] = ldv<T, V>(zp + iq[d]); rq[d] = ldv<T, V>(rp + iq[d]); }
    }
  }

  T pa[3];
  prologue_reduce<T, 3>(a.partsA, a.nA, a.gA, pa, smem);
  if (st.done) return;
  const T vs = sc_c * pa[0];                              // vectorSum = c * sum(p)  (:279)
  const T pz = pa[2] + vs * pa[0];
  T alpha = 0;
  if (absval(pz) > 0) alpha = pa[1] / pz;                 // :301-302
  if (blockIdx.x == 0 && threadIdx.x == 0) { a.scal[SC_PZ] = pz; a.scal[SC_VS] = vs; a.scal[SC_ALPHA] = alpha; }

  T acc_rz = 0, acc_r = 0, acc_ex = 0;
  const T accuracy = (T)a.accuracy;
  while (ch < cr.end) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      // finish chunk d, then refill its slot with the chunk D steps ahead
      if (okq[d]) {
        Vec<T, V> r0 = rq[d];
#pragma unroll
        for (int e = 0; e < V; ++e) {
          r0.v[e] = fma(-alpha, zq[d].v[e] + (true_cell(a, iq[d] + e) ? vs : (T)0), r0.v[e]);
          acc_rz = fma(r0.v[e], zq[d].v[e], acc_rz);
          acc_r += r0.v[e];
          acc_ex += (absval(r0.v[e]) < accuracy) ? (T)0 : (T)1;     // NaN counts as exceeding
        }
        if (a.nt & 1) stv_nt<T, V>(rp + iq[d], r0); else stv<T, V>(rp + iq[d], r0);
      }
      const int c = ch + (d + D) * cr.step;
      iq[d] = ((size_t)c * kBlock + threadIdx.x) * V;
      okq[d] = c < cr.end && iq[d] < n;
      if (okq[d]) {
        if (a.nt & 2) { zq[d] = ldc_nt<T, V>(zp + iq[d]); rq[d] = ldc_nt<T, V>(rp + iq[d]); }
        else { zq[d] = ldv<T, V>(zp + iq[d]); rq[d] = ldv<T, V>(rp + iq[d]); }
      }
    }
    ch += D * cr.step;
  }
  T part[3] = {acc_rz, acc_r, acc_ex};
  block_sum<T, 3>(part, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q) a.partsB[q * kMaxPartials + blockIdx.x] = part[q];
  }
}

// x <- x + alpha p for the direction of the LAST executed iteration (the loop ended without a following K1)
template <typename T>
__global__ __launch_bounds__(kBlock) void cg_flush_x(CgArgs<T> a, int k_last, int sv) {
  if (a.state[sv & 1].done) return;                       // a converged solve was flushed by the K1 that detected it
  const T alpha = a.scal[SC_ALPHA];
  const T* __restrict__ p = a.p[(k_last + 1) & 1];
  const size_t n = (size_t)a.nx * a.ny;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    a.x[i] = fma(alpha, p[i], a.x[i]);
}

// r <- b - (z' + c sum x) after a MODE_RESET application of K1 to x (pressure_solve_op.cu.cc:260-274)
template <typename T>
__global__ __launch_bounds__(kBlock) void cg_reset_residual(CgArgs<T> a, int sv) {
  __shared__ T smem[16];
  if (a.state[sv & 1].done) return;
  T pa[1];
  reduce_partials<T, 1>(a.partsA, a.nA, pa, smem);
  if (a.gA) pa[0] = a.gA[0];                              // slab mode: sum(x) over all ranks
  const T vs = a.scal[SC_C] * pa[0];
  const size_t n = (size_t)a.nx * a.ny;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    a.r[i] = a.b[i] - (a.z[i] + (true_cell(a, i) ? vs : (T)0));
}

// L [N][5] -> SoA coefficients (off-diagonals in T and in float); partial sums of |diag| for the shift (cublasDasum,
// :165-168); flags[0] is set if some off-diagonal is not exactly representable as float (then the T arrays are used),
// flags[1] if some diagonal is not bit-for-bit -(S + N + W + E) of the float off-diagonals (then it is read, not recomputed).
// flags[2] is set unless the matrix is symmetric bit for bit: N of a cell equals S of the cell above, E equals W of the cell to
// the right (periodic wrap, or 0 at a wall); nx = 0 skips the check and sets the flag; per_y = 2 (one slab of a decomposed grid)
// checks the pairs inside the slab only.
// ldx > 0 (padded-grid mode): the SoA outputs are laid out for a grid of ldx columns and n_out cells (zeroed by the caller).
template <typename T>
__global__ __launch_bounds__(kBlock) void cg_setup_coeffs(const T* __restrict__ L, T* cC, T* oT, float* oF, T* parts,
                                                           int* flags, size_t n, int nx = 0, int ny = 0, int per_x = 0,
                                                           int per_y = 0, int ldx = 0, size_t n_out = 0) {
  __shared__ T smem[16];
  T acc = 0;
  bool bad = false, bad_recon = false, bad_sym = (nx == 0);
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const T* row = L + i * 5;
    if (nx > 0) {
      const int ci = (int)(i % (size_t)nx), cj = (int)(i / (size_t)nx);
      const T e_nb = (ci + 1 < nx) ? L[(i + 1) * 5 + 1] : (per_x ? L[(i - (size_t)(nx - 1)) * 5 + 1] : (T)0);
      const T n_nb = (cj + 1 < ny) ? L[(i + (size_t)nx) * 5 + 0] : (per_y == 1 ? L[(size_t)ci * 5 + 0] : (per_y == 2 ? row[4] : (T)0));
      bad_sym |= !(row[3] == e_nb) || !(row[4] == n_nb);      // (per_y = 2, a slab: the last row's N pairs with a remote S - not checked)
    }
    const T o[4] = {row[0], row[1], row[3], row[4]};
    const size_t io = ldx ? (i / (size_t)nx) * (size_t)ldx + (i % (size_t)nx) : i;   // where cell i lives in the output layout
    const size_t no = ldx ? n_out : n;
    cC[io] = row[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float f = (float)o[q];
      oT[q * no + io] = o[q];
      oF[q * no + io] = f;
      bad |= !((T)f == o[q]);
    }
    // can the diagonal be recomputed from the float off-diagonals exactly as calcPISOLaplaceMatrix accumulated it?
    T d = 0;
    d -= (T)(float)o[0]; d -= (T)(float)o[3]; d -= (T)(float)o[1]; d -= (T)(float)o[2];
    bad_recon |= !(d == row[2]);
    acc += absval(row[2]);
  }
  if (bad) flags[0] = 1;
  if (bad_recon) flags[1] = 1;
  if (bad_sym) flags[2] = 1;
  T part[1] = {acc};
  block_sum<T, 1>(part, smem);
  if (threadIdx.x == 0) parts[blockIdx.x] = part[0];
}

// x = 0, r = b, p = 0 (both buffers), shift c, state  (pressure_solve_op.cu.cc:161-190, :104-114)
template <typename T>
__global__ __launch_bounds__(kBlock) void cg_init(CgArgs<T> a, int rank_deficient, const T* global_diag_sum = nullptr,
                                                   double global_cells = 0) {
  __shared__ T smem[16];
  T pa[1];
  reduce_partials<T, 1>(a.partsS, kMaxPartials, pa, smem);
  size_t n = (size_t)a.nx * a.ny;
  double ncells = (double)n;
  if (global_diag_sum) { pa[0] = global_diag_sum[0]; ncells = global_cells; }   // slab mode: all-reduced sum |diag|
  else if (a.ncells > 0) ncells = a.ncells;                                     // padded-grid mode: the cells of the true system
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    a.x[i] = 0;
    a.r[i] = a.b[i];
    a.p[0][i] = 0;
    a.p[1][i] = 0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.scal[SC_C] = rank_deficient ? pa[0] * (T)(.1 / ncells) : (T)0;
    a.scal[SC_PZ] = 1;
    a.scal[SC_VS] = 0;
    const CgState s = {0, 0, 0, 0};
    a.state[0] = s;
    a.state[1] = s;
  }
}

// ---- run-time verification of a solve that used the persistent kernel (cg.hip): the CG recurrences keep r = b - A^ x in exact
// arithmetic, and in floating point to ~ eps * condition * |b|; a perimeter value read before it was visible would break that
// identity for good (nothing re-synchronises r with x between residual resets).  Two passes over the state the solve ended in:
// sum(x), then the largest |b - (L x + c sum x) - r| and the largest |b|, as non-negative floats through atomicMax on their bit
// patterns.  Plain gathers - not on any hot path (one stencil pass per solve).
template <typename T>
__global__ __launch_bounds__(kBlock) void cg_verify_sum_x(CgArgs<T> a, T* parts) {
  __shared__ T smem[16];
  const size_t n = (size_t)a.nx * a.ny;
  T acc[1] = {0};
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) acc[0] += a.x[i];
  block_sum<T, 1>(acc, smem);
  if (threadIdx.x == 0) parts[blockIdx.x] = acc[0];
}
template <typename T, typename CT>
__global__ __launch_bounds__(kBlock) void cg_verify_gap(CgArgs<T> a, const T* parts, int nparts, unsigned* out2, const T* global_sum = nullptr) {
  __shared__ T smem[16];
  T sx[1] = {0};
  for (int b = threadIdx.x; b < nparts; b += kBlock) sx[0] += parts[b];
  block_sum<T, 1>(sx, smem);
  if (global_sum) sx[0] = global_sum[0];                  // slab mode: sum(x) over all ranks
  const T vs = a.scal[SC_C] * sx[0];
  const int nx = a.nx, ny = a.ny;
  const CT *oS = static_cast<const CT*>(a.oS), *oW = static_cast<const CT*>(a.oW), *oE = static_cast<const CT*>(a.oE), *oN = static_cast<const CT*>(a.oN);
  const size_t n = (size_t)nx * ny;
  float gap = 0.f, scale = 0.f;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const int ci = (int)(i % (size_t)nx), cj = (int)(i / (size_t)nx);
    auto at = [&](int ii, int jj) -> T {
      if (ii < 0) { if (!a.per_x) return (T)0; ii = nx - 1; }
      if (ii >= nx) { if (!a.per_x) return (T)0; ii = 0; }
      if (a.per_y != 2) {                                 // (slab mode: rows -1 and ny are halo rows in memory)
        if (jj < 0) { if (!a.per_y) return (T)0; jj = ny - 1; }
        if (jj >= ny) { if (!a.per_y) return (T)0; jj = 0; }
      }
      return a.x[(ptrdiff_t)jj * nx + ii];
    };
    T z = 0;
    z = fma((T)oS[i], at(ci, cj - 1), z);
    z = fma((T)oW[i], at(ci - 1, cj), z);
    z = fma(a.cC[i], a.x[i], z);
    z = fma((T)oE[i], at(ci + 1, cj), z);
    z = fma((T)oN[i], at(ci, cj + 1), z);
    const T d = a.b[i] - (z + (true_cell(a, i) ? vs : (T)0)) - a.r[i];
    const float g = (float)absval(d), sc = (float)absval(a.b[i]);
    if (g == g) gap = g > gap ? g : gap;                 // (NaN data: nothing to verify - the solve reports NaN as the reference does)
    if (sc == sc) scale = sc > scale ? sc : scale;
  }
  gap = wave_max(gap); scale = wave_max(scale);
  if ((threadIdx.x & 63) == 0) { atomicMax(out2, __float_as_uint(gap)); atomicMax(out2 + 1, __float_as_uint(scale)); }
}

template <typename T>
__global__ void cg_zero_partials(T* partsA, T* partsB, T* partsS) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * kMaxPartials) { partsA[i] = 0; partsB[i] = 0; }
  if (i < kMaxPartials) partsS[i] = 0;
}


}  // namespace piso
