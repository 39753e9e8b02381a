// Pressure CG of a TINY grid (at most kTinyMaxCells = 4 608 cells: the lid-driven cavity of BASELINE.json's config 1, 64 x 65) inside ONE
// workgroup: no grid-wide exchange, no launch per iteration - the ONE reduction of a CG iteration (merged as in cg_persist1.h) is
// a block reduction, and the residual resets of the reference's default residual_reset = 10 happen in the same launch.
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
                                                         CgState* state_out) {
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
  if (t == 0) *state_out = st;
}

}  // namespace piso
