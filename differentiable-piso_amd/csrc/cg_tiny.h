// Pressure CG of a TINY grid (at most kTinyMaxCells = 4 608 cells: the lid-driven cavity of BASELINE.json's config 1, 64 x 65) inside ONE
// workgroup: no grid-wide exchange, no launch per iteration - the ONE reduction of a CG iteration (merged as in cg_persist1.h) is
// a block reduction, and the residual resets of the reference's default residual_reset = 10 happen in the same launch.
//
// Why: on such a grid an iteration of the chip-wide paths is nothing but latency - two dependent launches (~13 us) or one grid
// exchange (~3.5 us even with all workgroups on one XCD) for 4 160 cells of arithmetic; here it is ~2 us
// (cg_tiny_cols below, the lid-driven cavity's shape) to ~4 us (cg_tiny, any shape).
//
// Same iteration and the same control flow as cg_k1 / cg_k2 driven by cg_run (cg_kernels.h, cg.hip), i.e. as
// pressure_solve_op.cu.cc:257-357: x0 = 0, r0 = b, the rank-1 shift c sum(p) with c = 0.1 mean|diag| (:161-190, :277-286), the
// stopping test of iteration k - 1 evaluated at the top of iteration k for k % 5 == 0 with the device-flag semantics (:312-335),
// beta unguarded (:351-352), alpha guarded (:301-302), restart r = b - (L x + c sum x), p = r when (k + 1) % reset == 0 (:260-274).
// The matrix is taken as given ([N][5] = -y, -x, diag, +x, +y, any values): the coefficients of a thread's cells, its part of the
// direction and z' live in its registers; r, x, b and the direction (the stencil reads its neighbours there) in LDS.  Summation order inside a cell as calcZ_v4
// (:81-90): S, W, C, E, N with explicit fma.
#pragma once
#include "cg_kernels.h"
#include "cg_persist.h"
#include "cg_persist1.h"

namespace piso {

constexpr int kTinyThreads = 512;             // 8 waves = 2 per SIMD: 256 VGPRs per lane hold the coefficients, p and z' of 9 cells
constexpr int kTinyCellsPerThread = 9;
constexpr int kTinyMaxCells = 4608;                                    // four fp64 vectors of the grid fit the LDS (4 x 36 KB)

// sum of NV values over the workgroup, the same bits in every thread (fixed order: lanes by butterfly, then the 16 waves in order)
template <typename T, int NV>
__device__ __forceinline__ void tiny_block_sum(T (&v)[NV], T* smem /* [NV * 16] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = (T)wave_sum_uniform((double)v[q]);   // DPP network: no LDS round trips in the dependent chain
  __syncthreads();                                          // (smem may still be read from the previous reduction)
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < NV; ++q) smem[q * 16 + wave] = v[q];
  }
  __syncthreads();
  // a fixed TREE over the waves' sums (the loop runs on the latency of dependent fp64 additions: depth 3 instead of 8)
  constexpr int NW = kTinyThreads / 64;
  static_assert(NW == 8, "the tree below adds eight wave sums");
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const T* w = smem + q * 16;
    v[q] = ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
  }
}

// out: CgState {flag, done, iterations}; total = iterations to run at most; reset <= 0: never restart (fixed-work mode)
template <typename T>
__global__ __launch_bounds__(kTinyThreads) void cg_tiny(const T* __restrict__ L, const T* __restrict__ b, T* __restrict__ x_out, int nx, int ny,
                                                         int per_x, int per_y, float accuracy_f, int total, int reset, int rank_deficient,
                                                         CgState* state_out, int* iterations_dev) {
  constexpr int C = kTinyCellsPerThread;
  __shared__ T pbuf[kTinyMaxCells + 1];                     // the direction (slot n: the zero every missing neighbour reads)
  __shared__ T rbuf[kTinyMaxCells], xbuf[kTinyMaxCells];    // r and x of a thread's own cells (168 VGPRs per lane hold the coefficients,
                                                            // p and z' of six cells; r and x on top of them spill)
  __shared__ T bbuf[kTinyMaxCells];                         // the right-hand side (read again at every residual reset)
  __shared__ T smem[8 * 16];                                 // (block sums: up to 16 waves)
  const int n = nx * ny;
  const int t = threadIdx.x;
  T cS[C], cW[C], cC[C], cE[C], cN[C], pm[C];
  unsigned iSW[C], iEN[C];                                  // LDS slots of the four neighbours, two 16-bit slots per register (n: none)
  bool own[C];
  T dsum[1] = {0};
#pragma unroll
  for (int m = 0; m < C; ++m) {
    const int i = t + m * kTinyThreads;
    own[m] = i < n;
    cS[m] = cW[m] = cC[m] = cE[m] = cN[m] = 0; pm[m] = 0;
    iSW[m] = iEN[m] = (unsigned)n | ((unsigned)n << 16);
    if (own[m]) {
      const T* row = L + (size_t)i * 5;
      cS[m] = row[0]; cW[m] = row[1]; cC[m] = row[2]; cE[m] = row[3]; cN[m] = row[4];
      const T bi = b[i];
      rbuf[i] = bi;
      bbuf[i] = bi;
      xbuf[i] = 0;
      dsum[0] += absval(row[2]);
      const int ci = i % nx, cj = i / nx;
      const unsigned w_ = (unsigned)(ci > 0 ? i - 1 : (per_x ? i + nx - 1 : n)), e_ = (unsigned)(ci + 1 < nx ? i + 1 : (per_x ? i - (nx - 1) : n));
      const unsigned s_ = (unsigned)(cj > 0 ? i - nx : (per_y ? i + (ny - 1) * nx : n)), n_ = (unsigned)(cj + 1 < ny ? i + nx : (per_y ? i - (ny - 1) * nx : n));
      iSW[m] = s_ | (w_ << 16);
      iEN[m] = e_ | (n_ << 16);
      pbuf[i] = 0;
    }
  }
  if (t == 0) pbuf[n] = 0;
  tiny_block_sum<T, 1>(dsum, smem);
  const T sc_c = rank_deficient ? dsum[0] * (T)(.1 / (double)n) : (T)0;            // cg_init
  const T accuracy = (T)accuracy_f;
  // z' = L v of my cells, v read from pbuf
  auto stencil = [&](int m) __attribute__((always_inline)) -> T {
    T z = 0;
    z = fma(cS[m], pbuf[iSW[m] & 0xffffu], z);
    z = fma(cW[m], pbuf[iSW[m] >> 16], z);
    z = fma(cC[m], pm[m], z);
    z = fma(cE[m], pbuf[iEN[m] & 0xffffu], z);
    z = fma(cN[m], pbuf[iEN[m] >> 16], z);
    return z;
  };
  CgState st = {0, 0, 0, 0};
  // ONE block reduction per iteration, as in cg_persist1.h: D(k) sums  sum p, p.r, p.z', r.z', z'.z', sum z'  and, left over from
  // U(k-1),  sum r_k, #{|r_k| >= accuracy};  alpha_k, then  r_{k+1}.z'_k = r.z' - alpha (z'.z' + vs sum z')  and
  // sum r_{k+1} = sum r_k - alpha (sum z' + N vs)  give beta_{k+1} without a second reduction (one-step recurrences from directly
  // summed quantities).  The stopping test of iteration k is evaluated behind that reduction, before x and r move on: a converged
  // solve stops in exactly the reference's state (x_k, iterations = k).
  T pz = 1, vs = 0, rz_next = 0, sumr = 0;                  // (cg_init: SC_PZ = 1, SC_VS = 0)
  T lr = 0, lc = 0;                                         // this thread's part of sum r_k and of #{|r_k| >= accuracy}
#pragma unroll
  for (int m = 0; m < C; ++m) if (own[m]) { const T rv = rbuf[t + m * kTinyThreads]; lr += rv; lc += (absval(rv) < accuracy) ? (T)0 : (T)1; }
  const T ncells = (T)n;
  for (int k = 0; k < total && !st.done; ++k) {
    const bool is_reset = reset > 0 && ((k + 1) % reset == 0);
    T beta = 0;
    if (k > 0 && !is_reset) beta = -(rz_next + vs * sumr) / pz;     // unguarded, as coded (:351-352)
    bool tested = false;
    T z[C];
    if (is_reset) {
      // the test of iteration k belongs in front of the restart (cg_k1 MODE_RESET with do_check): #{|r_k| >= accuracy} travels with sum x
      T sx[2] = {0, lc};
#pragma unroll
      for (int m = 0; m < C; ++m) { pm[m] = 0; if (own[m]) { const T xv = xbuf[t + m * kTinyThreads]; pm[m] = xv; pbuf[t + m * kTinyThreads] = xv; sx[0] += xv; } }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < C; ++m) z[m] = stencil(m);
      tiny_block_sum<T, 2>(sx, smem);                       // (its barriers also end the stencil's reads of x)
      if (k > 0 && (k % 5) == 0) {
        const int exceeded = sx[1] > 0;
        if (st.flag && !exceeded) { st.done = 1; st.iterations = k; }
        else st.flag = 1;
      }
      if (st.done) break;
      tested = true;
      st.flag = 0;                                          // initVariablesWithGuess clears the device flag
      // r = b - (L x + c sum x), then the common path with beta = 0: p = r                (:260-274)
      const T vsx = sc_c * sx[0];
      lr = 0; lc = 0;
#pragma unroll
      for (int m = 0; m < C; ++m) {
        if (own[m]) {
          const int i = t + m * kTinyThreads;
          const T rn = bbuf[i] - (z[m] + vsx);
          rbuf[i] = rn; lr += rn; lc += (absval(rn) < accuracy) ? (T)0 : (T)1;
        }
        pm[m] = 0;                                          // (p = r + 0 p below)
      }
    }
    // (no barrier here: the block sum of the previous iteration separates its stencil reads from these writes)
#pragma unroll
    for (int m = 0; m < C; ++m) {
      if (own[m]) { const int i = t + m * kTinyThreads; pm[m] = fma(beta, pm[m], rbuf[i]); pbuf[i] = pm[m]; }   // p = r + beta p (k = 0, resets: p = r)
    }
    __syncthreads();
    // ---- D(k): z' = L p and the sums
    T sD[8] = {0, 0, 0, 0, 0, 0, lr, lc};
#pragma unroll
    for (int m = 0; m < C; ++m) {
      z[m] = stencil(m);
      if (own[m]) {
        const T rv = rbuf[t + m * kTinyThreads];
        sD[0] += pm[m];
        sD[1] = fma(pm[m], rv, sD[1]);
        sD[2] = fma(pm[m], z[m], sD[2]);
        sD[3] = fma(rv, z[m], sD[3]);
        sD[4] = fma(z[m], z[m], sD[4]);
        sD[5] += z[m];
      }
    }
    tiny_block_sum<T, 8>(sD, smem);
    // ---- the stopping test of iteration k (:312-335), behind the reduction but before anything moves
    if (!tested && k > 0 && (k % 5) == 0) {
      const int exceeded = sD[7] > 0;
      if (st.flag && !exceeded) { st.done = 1; st.iterations = k; }
      else st.flag = 1;
    }
    if (st.done) break;
    // ---- alpha (:301-302) and what beta of the next iteration needs
    sumr = sD[6];
    vs = sc_c * sD[0];
    pz = sD[2] + vs * sD[0];
    const T alpha = (absval(pz) > 0) ? sD[1] / pz : (T)0;
    rz_next = sD[3] - alpha * (sD[4] + vs * sD[5]);
    sumr = sumr - alpha * (sD[5] + ncells * vs);
    // ---- U(k): x += alpha p; r -= alpha (z' + c sum p)
    lr = 0; lc = 0;
#pragma unroll
    for (int m = 0; m < C; ++m) {
      if (own[m]) {
        const int i = t + m * kTinyThreads;
        xbuf[i] = fma(alpha, pm[m], xbuf[i]);
        const T rn = fma(-alpha, z[m] + vs, rbuf[i]);
        rbuf[i] = rn;
        lr += rn;
        lc += (absval(rn) < accuracy) ? (T)0 : (T)1;       // NaN counts as exceeding
      }
    }
  }
#pragma unroll
  for (int m = 0; m < C; ++m) if (own[m]) x_out[t + m * kTinyThreads] = xbuf[t + m * kTinyThreads];
  if (t == 0) { *state_out = st; if (iterations_dev) *iterations_dev = st.done ? st.iterations : total; }
}

// ---- the same solve with a COLUMN layout: lane = x (nx <= 64), wave w owns the rows [9 w, 9 w + 9) (ny <= 72: the lid-driven
// cavity is 64 x 65).  p, r, z' and the five coefficients of a thread's nine vertically adjacent cells live in registers (x and b in
// LDS, off the dependent chain); the south / north neighbours of the direction are the thread's own registers - for a wave's edge
// rows: RING COPIES of the neighbouring wave's row, advanced with the owner's arithmetic (see cols_solve) - west / east come from
// the neighbouring lanes through the DPP wavefront shifts.  LDS traffic per iteration: x read + write, 2 z' row writes + 2 reads
// per wave and one line of wave sums (cg_tiny: ~90 accesses per thread); ONE barrier per iteration.  The block reduction is the exchange of cg_persist1.h in small: reduce-scatter butterfly of the eight partial sums
// inside a wave (lane l ends with value l & 7), one LDS line [8 values][8 waves], ONE barrier, one LDS read per lane and a
// three-step tree over the waves - every wave computes the same bits.
// PERX (periodic x) needs nx == 64: the wavefront ROTATES are the wrap-around.
#ifndef PISO_TINY_COLS_FULL
#define PISO_TINY_COLS_FULL 0                 // 1: waves whose nine rows and 64 lanes are all cells run an instance without masks (measured: it spills)
#endif
constexpr int kColsRows = 9;                  // rows per wave
constexpr int kColsMaxNy = 8 * kColsRows;

template <bool UP, bool WRAP, typename S>
__device__ __forceinline__ S cols_shift(S v) {      // lane l <- lane l - 1 (UP) / l + 1; the lane without a source: 0, or the other end (WRAP)
  constexpr int ctrl = WRAP ? (UP ? 0x13C /* wave_ror:1 */ : 0x134 /* wave_rol:1 */) : (UP ? 0x138 /* wave_shr:1 */ : 0x130 /* wave_shl:1 */);
  if constexpr (sizeof(S) == 8) {
    const unsigned long long b = (unsigned long long)__double_as_longlong((double)v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, ctrl, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), ctrl, 0xf, 0xf, false);
    return (S)__longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  } else {
    return (S)__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int((float)v), ctrl, 0xf, 0xf, false));
  }
}

// eight partial sums per thread -> the block totals, the same bits in every thread (v[] is overwritten)
template <typename T>
__device__ __forceinline__ void cols_block_sum8(T (&v)[8], double* red /* [64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double d[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) d[q] = (double)v[q];
  const double mine = wave_reduce_scatter8(d);              // lane l: this wave's total of value l & 7
  if (lane < 8) red[lane * 8 + wave] = mine;
  __syncthreads();
  double t = red[lane];                                     // lane l: value l >> 3 of wave l & 7
  t += dpp_move<0xB1>(t);                                   // (w0 + w1), (w2 + w3), ...
  t += dpp_move<0x4E>(t);                                   // ((w0 + w1) + (w2 + w3)), ...
  t += lanes_xor4(t);                                       // all eight waves: lanes 8 q .. 8 q + 7 hold the total of value q
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = (T)read_lane_c(t, 8 * q);
}

// FULL: all nine rows and all 64 lanes of this wave are cells (no masks); the waves of a workgroup may run different instances -
// they execute the same sequence of barriers.
template <typename T, bool PERX, bool FULL>
__device__ __forceinline__ void cols_solve(const T* __restrict__ L, const T* __restrict__ b, T* __restrict__ x_out, int nx, int ny, int per_y,
                                           float accuracy_f, int total, int reset, int rank_deficient, CgState* state_out, int* iterations_dev,
                                           T (*halo)[kTinyThreads / 64][64], T (*zhalo)[2][kTinyThreads / 64][64], double (*red2)[64],
                                           T* bbuf, T* xbuf) {
  constexpr int C = kColsRows;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j0 = wave * C;
  const int rows = FULL ? C : min(max(ny - j0, 0), C);      // (wave-uniform)
  const bool col = FULL || lane < nx;
  const int n = nx * ny;
  const int wlast = (ny - 1) / C;                           // the wave that owns the top row of the grid
  // where my south / north halo rows come from (-1: nowhere - a wall or an open boundary, the coefficient there multiplies 0)
  const int srcS = rows == 0 ? -1 : (wave > 0 ? wave - 1 : (per_y ? wlast : -1));
  const int srcN = rows == 0 ? -1 : (j0 + rows < ny ? wave + 1 : (per_y ? 0 : -1));
  T cS[C], cW[C], cC[C], cE[C], cN[C], p[C], r[C], z[C];
  T* xmine = xbuf + j0 * 64 + lane;                         // x of my cells: read and written once per iteration, off the dependent chain
  T* bmine = bbuf + j0 * 64 + lane;                         // the right-hand side of my cells (read again at every residual reset)
  T dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int m = 0; m < C; ++m) {
    cS[m] = cW[m] = cC[m] = cE[m] = cN[m] = 0; p[m] = r[m] = z[m] = 0;
    xmine[m * 64] = 0;
    if (m < rows && col) {
      const int i = (j0 + m) * nx + lane;
      const T* row = L + (size_t)i * 5;
      cS[m] = row[0]; cW[m] = row[1]; cC[m] = row[2]; cE[m] = row[3]; cN[m] = row[4];
      r[m] = b[i]; bmine[m * 64] = r[m];
      dsum[0] += absval(row[2]);
    }
  }
  int par = 0;                                              // parity of the block sums: `red2` and `zhalo` are double-buffered by it
  cols_block_sum8<T>(dsum, red2[par]); par ^= 1;
  const T sc_c = rank_deficient ? dsum[0] * (T)(.1 / (double)n) : (T)0;            // cg_init
  const T accuracy = (T)accuracy_f;
  // RING COPIES (as cg_persist1.h keeps them around a region): r and p of the row below my first row and of the row above my last
  // one live in MY registers too and are advanced with the same arithmetic as their owner advances them - p = r + beta p needs
  // nothing from outside, r -= alpha (z' + c sum p) needs the neighbour's z' of that row, which travels through LDS in the shadow of
  // the block reduction's barrier.  The stencil's south / north inputs of my edge rows are then registers: ONE barrier per
  // iteration instead of two (the rows of a fresh vector - x at a residual reset - still go through `halo` with a barrier).
  const bool hasS = srcS >= 0, hasN = srcN >= 0;            // (wave-uniform)
  const int jS = j0 > 0 ? j0 - 1 : ny - 1, jN = j0 + rows < ny ? j0 + rows : 0;      // the ring rows' global row numbers (if they exist)
  T rS = (hasS && col) ? bbuf[jS * 64 + lane] : (T)0, rN = (hasN && col) ? bbuf[jN * 64 + lane] : (T)0;   // r0 = b (written before the barrier above)
  T pS = 0, pN = 0;
  // z' = L v for my cells, v in registers (phantom rows and lanes hold 0), hS / hN: v on the rows below / above mine
  auto stencil = [&](const T (&v)[C], T hS, T hN) __attribute__((always_inline)) {
#pragma unroll
    for (int mm = 0; mm < C; ++mm) {
      const int m = mm < C - 2 ? mm + 1 : (mm == C - 2 ? 0 : C - 1);     // the rows next to the halo rows last: their LDS reads are in flight
      const T vS = m == 0 ? hS : v[m - 1];
      T vN = m + 1 < C ? v[m + 1] : hN;
      if (m + 1 == rows) vN = hN;                            // (wave-uniform: only the wave with the grid's top row has rows < C)
      const T vW = cols_shift<true, PERX>(v[m]), vE = cols_shift<false, PERX>(v[m]);
      T acc = 0;                                             // summation order of calcZ_v4 (:81-90): S, W, C, E, N
      acc = fma(cS[m], vS, acc);
      acc = fma(cW[m], vW, acc);
      acc = fma(cC[m], v[m], acc);
      acc = fma(cE[m], vE, acc);
      acc = fma(cN[m], vN, acc);
      z[m] = acc;
      __builtin_amdgcn_sched_barrier(0);                     // (one row at a time: the shifted copies of ALL rows at once do not fit the registers)
    }
  };
  // my first and last rows of z' for the waves below and above (read behind the next block sum's barrier)
  auto publish_z_edges = [&]() __attribute__((always_inline)) {
    if (rows > 0) {
      zhalo[par][0][wave][lane] = z[0];
#pragma unroll
      for (int m = 0; m < C; ++m) if (m == rows - 1) zhalo[par][1][wave][lane] = z[m];
    }
  };
  CgState st = {0, 0, 0, 0};
  T pz = 1, vs = 0, rz_next = 0, sumr = 0;                  // (cg_init: SC_PZ = 1, SC_VS = 0)
  T lr = 0;
  int lc = 0;                                               // this thread's part of sum r_k and of #{|r_k| >= accuracy}
#pragma unroll
  for (int m = 0; m < C; ++m) if (m < rows && col) { lr += r[m]; lc += (absval(r[m]) < accuracy) ? 0 : 1; }
  const T ncells = (T)n;
#ifdef PISO_TINY_DIAG
  long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = clock64();
#define TINY_TICK(q) { __builtin_amdgcn_sched_barrier(0); const long long tn = clock64(); tph[q] += tn - tlast; tlast = tn; __builtin_amdgcn_sched_barrier(0); }
#else
#define TINY_TICK(q)
#endif
  T beta_next = 0;
  int to_reset = reset > 0 ? reset - 1 : -1;                // iterations until the next restart (no integer division in the loop)
  int to_test = 0;                                          // k % 5
  for (int k = 0; k < total && !st.done; ++k) {
    const bool is_reset = to_reset == 0;
    to_reset = is_reset ? reset - 1 : to_reset - 1;
    const bool test_now = k > 0 && to_test == 0;            // the stopping test of iteration k
    const bool count_now = to_test == 4;                    // ... needs #{|r_k| >= accuracy}, counted by U(k - 1)
    to_test = to_test == 4 ? 0 : to_test + 1;
    const T beta = is_reset ? (T)0 : beta_next;
    bool tested = false;
    if (is_reset) {
      // the test of iteration k belongs in front of the restart (cg_k1 MODE_RESET with do_check): #{|r_k| >= accuracy} travels with sum x
      T sx[8] = {0, (T)lc, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int m = 0; m < C; ++m) { p[m] = xmine[m * 64]; sx[0] += p[m]; }     // (p = r + 0 p follows)
      {   // the rows of x next to mine through LDS (a barrier of its own: once per reset)
        if (rows > 0) {
          halo[0][wave][lane] = p[0];
#pragma unroll
          for (int m = 0; m < C; ++m) if (m == rows - 1) halo[1][wave][lane] = p[m];
        }
        __syncthreads();
        const T xS = hasS ? halo[1][srcS][lane] : (T)0, xN = hasN ? halo[0][srcN][lane] : (T)0;
        stencil(p, xS, xN);
      }
      publish_z_edges();                                    // (L x of the ring rows: their restarted residual)
      const int parx = par;
      cols_block_sum8<T>(sx, red2[par]); par ^= 1;
      if (test_now) {
        const int exceeded = sx[1] > 0;
        if (st.flag && !exceeded) { st.done = 1; st.iterations = k; }
        else st.flag = 1;
      }
      if (st.done) break;
      tested = true;
      st.flag = 0;                                          // initVariablesWithGuess clears the device flag
      // r = b - (L x + c sum x), then the common path with beta = 0: p = r                (:260-274)
      const T vsx = sc_c * sx[0];
      lr = 0; lc = 0;
#pragma unroll
      for (int m = 0; m < C; ++m) {
        if (m < rows && col) {
          r[m] = bmine[m * 64] - (z[m] + vsx);
          lr += r[m]; lc += (absval(r[m]) < accuracy) ? 0 : 1;
        }
      }
      rS = (hasS && col) ? bbuf[jS * 64 + lane] - (zhalo[parx][1][srcS < 0 ? 0 : srcS][lane] + vsx) : (T)0;
      rN = (hasN && col) ? bbuf[jN * 64 + lane] - (zhalo[parx][0][srcN < 0 ? 0 : srcN][lane] + vsx) : (T)0;
    }
#pragma unroll
    for (int m = 0; m < C; ++m) p[m] = fma(beta, p[m], r[m]);       // p = r + beta p (k = 0, resets: beta = 0; phantoms stay 0)
    pS = is_reset ? rS : fma(beta, pS, rS);                 // (a reset: p = r - the copies held x's neighbours never, but 0 x inf is not 0)
    pN = is_reset ? rN : fma(beta, pN, rN);
    // ---- D(k): z' = L p and the sums
    TINY_TICK(0)                                            // loop top, p update
    stencil(p, pS, pN);
    publish_z_edges();
    TINY_TICK(1)                                            // stencil
    T sD[8] = {0, 0, 0, 0, 0, 0, lr, (T)lc};
#pragma unroll
    for (int m = 0; m < C; ++m) {
      sD[0] += p[m];
      sD[1] = fma(p[m], r[m], sD[1]);
      sD[2] = fma(p[m], z[m], sD[2]);
      sD[3] = fma(r[m], z[m], sD[3]);
      sD[4] = fma(z[m], z[m], sD[4]);
      sD[5] += z[m];
    }
    TINY_TICK(2)                                            // partial sums
    const int parz = par;
    cols_block_sum8<T>(sD, red2[par]); par ^= 1;
    const T zS = hasS ? zhalo[parz][1][srcS][lane] : (T)0, zN = hasN ? zhalo[parz][0][srcN][lane] : (T)0;   // (consumed behind alpha)
    TINY_TICK(3)                                            // block reduction
    // ---- the stopping test of iteration k (:312-335), behind the reduction but before anything moves
    if (!tested && test_now) {
      const int exceeded = sD[7] > 0;
      if (st.flag && !exceeded) { st.done = 1; st.iterations = k; }
      else st.flag = 1;
    }
    if (st.done) break;
    // ---- alpha (:301-302) and what beta of the next iteration needs
    sumr = sD[6];
    vs = sc_c * sD[0];
    pz = sD[2] + vs * sD[0];
    const T alpha = (absval(pz) > 0) ? sD[1] / pz : (T)0;
    rz_next = sD[3] - alpha * (sD[4] + vs * sD[5]);
    sumr = sumr - alpha * (sD[5] + ncells * vs);
    beta_next = -(rz_next + vs * sumr) / pz;                // beta of iteration k + 1: unguarded, as coded (:351-352); same divisor as alpha
    TINY_TICK(4)                                            // test, alpha, beta
    // ---- U(k): x += alpha p; r -= alpha (z' + c sum p) on my cells and on the ring
    lr = 0; lc = 0;
    rS = fma(-alpha, zS + ((hasS && col) ? vs : (T)0), rS);
    rN = fma(-alpha, zN + ((hasN && col) ? vs : (T)0), rN);
    if (count_now) {             // (the count is only read by the test of iteration k + 1)
#pragma unroll
      for (int m = 0; m < C; ++m) {
        xmine[m * 64] = fma(alpha, p[m], xmine[m * 64]);
        if (m < rows && col) {
          r[m] = fma(-alpha, z[m] + vs, r[m]);
          lr += r[m];
          lc += (absval(r[m]) < accuracy) ? 0 : 1;          // NaN counts as exceeding
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < C; ++m) {
        xmine[m * 64] = fma(alpha, p[m], xmine[m * 64]);
        if (m < rows && col) {
          r[m] = fma(-alpha, z[m] + vs, r[m]);
          lr += r[m];
        }
      }
    }
    TINY_TICK(5)                                            // U
  }
#ifdef PISO_TINY_DIAG
  if (lane == 0 && (wave == 0 || wave == 7))
    printf("cg_tiny_cols wave %d: clocks per phase: top %lld, halo + stencil %lld, sums %lld, reduction %lld, scalars %lld, U %lld\n", wave, tph[0], tph[1],
           tph[2], tph[3], tph[4], tph[5]);
#endif
#pragma unroll
  for (int m = 0; m < C; ++m) if (m < rows && col) x_out[(j0 + m) * nx + lane] = xmine[m * 64];
  if (threadIdx.x == 0) { *state_out = st; if (iterations_dev) *iterations_dev = st.done ? st.iterations : total; }
}

template <typename T, bool PERX>
__global__ __launch_bounds__(kTinyThreads) void cg_tiny_cols(const T* __restrict__ L, const T* __restrict__ b, T* __restrict__ x_out, int nx, int ny,
                                                              int per_y, float accuracy_f, int total, int reset, int rank_deficient,
                                                              CgState* state_out, int* iterations_dev) {
  // `red2` and `zhalo` are double-buffered by the parity of the block sums: ONE barrier per iteration sits between a buffer's writes
  // and its reads, and whoever writes the same parity again has passed the NEXT barrier, which every wave reaches only after its
  // reads of this one.  `halo` (the rows of x at a residual reset) has a barrier of its own and is used once per reset.
  __shared__ T halo[2][kTinyThreads / 64][64];              // [0]: a wave's first row, [1]: its last row
  __shared__ T zhalo[2][2][kTinyThreads / 64][64];          // [parity][0: first row / 1: last row][wave][lane]: z' of the waves' edge rows
  __shared__ double red[2][64];
  __shared__ T bbuf[kColsMaxNy * 64], xbuf[kColsMaxNy * 64];
  const int wave = threadIdx.x >> 6;
  if (PISO_TINY_COLS_FULL && nx == 64 && (wave + 1) * kColsRows <= ny)
    cols_solve<T, PERX, true>(L, b, x_out, nx, ny, per_y, accuracy_f, total, reset, rank_deficient, state_out, iterations_dev, halo, zhalo, red, bbuf, xbuf);
  else
    cols_solve<T, PERX, false>(L, b, x_out, nx, ny, per_y, accuracy_f, total, reset, rank_deficient, state_out, iterations_dev, halo, zhalo, red, bbuf, xbuf);
}


}  // namespace piso
