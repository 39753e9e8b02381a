// Pressure CG of a TINY grid (at most kTinyMaxCells = 4 608 cells: the lid-driven cavity of BASELINE.json's config 1, 64 x 65) inside ONE
// workgroup: no grid-wide exchange, no launch per iteration - the two reductions of a CG iteration are block reductions between
// two __syncthreads, and the residual resets of the reference's default residual_reset = 10 happen in the same launch.
//
// Why: on such a grid an iteration of the chip-wide paths is nothing but latency - two dependent launches (~13 us) or one grid
// exchange (~3.6 us even with all workgroups on one XCD) for 4 160 cells of arithmetic; here it is ~1 us.
//
// Same iteration and the same control flow as cg_k1 / cg_k2 driven by cg_run (cg_kernels.h, cg.hip), i.e. as
// pressure_solve_op.cu.cc:257-357: x0 = 0, r0 = b, the rank-1 shift c sum(p) with c = 0.1 mean|diag| (:161-190, :277-286), the
// stopping test of iteration k - 1 evaluated at the top of iteration k for k % 5 == 0 with the device-flag semantics (:312-335),
// beta unguarded (:351-352), alpha guarded (:301-302), restart r = b - (L x + c sum x), p = r when (k + 1) % reset == 0 (:260-274).
// The matrix is taken as given ([N][5] = -y, -x, diag, +x, +y, any values): the coefficients of a thread's cells live in its
// registers, r and x too, the direction in LDS (the stencil reads its neighbours there).  Summation order inside a cell as calcZ_v4
// (:81-90): S, W, C, E, N with explicit fma.
#pragma once
#include "cg_kernels.h"
#include "cg_persist.h"

namespace piso {

constexpr int kTinyThreads = 768;             // 12 waves = 3 per SIMD: 168 VGPRs per lane hold coefficients, r, x, p, z of 6 cells
constexpr int kTinyCellsPerThread = 6;
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
  // a fixed TREE over the waves' sums (the loop runs on the latency of dependent fp64 additions: depth 4 instead of 12)
  constexpr int NW = kTinyThreads / 64;
  static_assert(NW == 12, "the tree below adds twelve wave sums");
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const T* w = smem + q * 16;
    const T a0 = w[0] + w[1], a1 = w[2] + w[3], a2 = w[4] + w[5], a3 = w[6] + w[7], a4 = w[8] + w[9], a5 = w[10] + w[11];
    v[q] = ((a0 + a1) + (a2 + a3)) + (a4 + a5);
  }
}

// out: CgState {flag, done, iterations}; total = iterations to run at most; reset <= 0: never restart (fixed-work mode)
template <typename T>
__global__ __launch_bounds__(kTinyThreads) void cg_tiny(const T* __restrict__ L, const T* __restrict__ b, T* __restrict__ x_out, int nx, int ny,
                                                         int per_x, int per_y, float accuracy_f, int total, int reset, int rank_deficient,
                                                         CgState* state_out) {
  constexpr int C = kTinyCellsPerThread;
  __shared__ T pbuf[kTinyMaxCells + 1];                     // the direction (slot n: the zero every missing neighbour reads)
  __shared__ T rbuf[kTinyMaxCells], xbuf[kTinyMaxCells];    // r and x of a thread's own cells (168 VGPRs per lane hold the coefficients,
                                                            // p and z' of six cells; r and x on top of them spill)
  __shared__ T bbuf[kTinyMaxCells];                         // the right-hand side (read again at every residual reset)
  __shared__ T smem[3 * 16];                                 // (block sums: up to 16 waves)
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
  T pz = 1, vs = 0;                                         // (cg_init: SC_PZ = 1, SC_VS = 0)
  T tB[3] = {0, 0, 0};                                      // r.z', sum r, #cells with !(|r| < accuracy) of the previous iteration
  for (int k = 0; k < total && !st.done; ++k) {
    const bool is_reset = reset > 0 && ((k + 1) % reset == 0);
    // ---- top of iteration k: the stopping test of iteration k - 1 and beta (cg_k1 with do_check)
    T beta = 0;
    if (k > 0) {
      if ((k % 5) == 0) {
        const int exceeded = tB[2] > 0;
        if (st.flag && !exceeded) { st.done = 1; st.iterations = k; }
        else st.flag = 1;
      }
      if (st.done) break;
      if (!is_reset) beta = -(tB[0] + vs * tB[1]) / pz;     // unguarded, as coded (:351-352)
    }
    T z[C];
    if (is_reset) {
      st.flag = 0;                                          // initVariablesWithGuess clears the device flag
      // r = b - (L x + c sum x), then the common path with beta = 0: p = r                (:260-274)
      T sx[1] = {0};
#pragma unroll
      for (int m = 0; m < C; ++m) { pm[m] = 0; if (own[m]) { const T xv = xbuf[t + m * kTinyThreads]; pm[m] = xv; pbuf[t + m * kTinyThreads] = xv; sx[0] += xv; } }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < C; ++m) z[m] = stencil(m);
      tiny_block_sum<T, 1>(sx, smem);                       // (its barriers also end the stencil's reads of x)
      const T vsx = sc_c * sx[0];
#pragma unroll
      for (int m = 0; m < C; ++m) {
        if (own[m]) { const int i = t + m * kTinyThreads; rbuf[i] = bbuf[i] - (z[m] + vsx); }
        pm[m] = 0;                                          // (p = r + 0 p below)
      }
    }
    // (no barrier here: the two block sums of the previous iteration separate its stencil reads from these writes)
#pragma unroll
    for (int m = 0; m < C; ++m) {
      if (own[m]) { const int i = t + m * kTinyThreads; pm[m] = fma(beta, pm[m], rbuf[i]); pbuf[i] = pm[m]; }   // p = r + beta p (k = 0, resets: p = r)
    }
    __syncthreads();
    // ---- K1: z' = L p; sum p, p.r, p.z'
    T sA[3] = {0, 0, 0};
#pragma unroll
    for (int m = 0; m < C; ++m) {
      z[m] = stencil(m);
      sA[0] += pm[m];
      if (own[m]) sA[1] = fma(pm[m], rbuf[t + m * kTinyThreads], sA[1]);
      sA[2] = fma(pm[m], z[m], sA[2]);
    }
    tiny_block_sum<T, 3>(sA, smem);
    // ---- K2: alpha; x += alpha p; r -= alpha (z' + c sum p); r.z', sum r, #{|r| >= accuracy}
    vs = sc_c * sA[0];
    pz = sA[2] + vs * sA[0];
    const T alpha = (absval(pz) > 0) ? sA[1] / pz : (T)0;   // (:301-302)
    tB[0] = tB[1] = tB[2] = 0;
#pragma unroll
    for (int m = 0; m < C; ++m) {
      if (own[m]) {
        const int i = t + m * kTinyThreads;
        xbuf[i] = fma(alpha, pm[m], xbuf[i]);
        const T rn = fma(-alpha, z[m] + vs, rbuf[i]);
        rbuf[i] = rn;
        tB[0] = fma(rn, z[m], tB[0]);
        tB[1] += rn;
        tB[2] += (absval(rn) < accuracy) ? (T)0 : (T)1;    // NaN counts as exceeding
      }
    }
    tiny_block_sum<T, 3>(tB, smem);
  }
#pragma unroll
  for (int m = 0; m < C; ++m) if (own[m]) x_out[t + m * kTinyThreads] = xbuf[t + m * kTinyThreads];
  if (t == 0) *state_out = st;
}

}  // namespace piso
