// Persistent CG segment kernel: many NORMAL iterations in ONE launch, for grids whose per-wave working set fits on chip.
//
// One workgroup of 512 threads per CU; every wave owns two fixed regions of 128 columns x R rows (2 x 16 cells per lane at
// R = 8, i.e. up to 4096 regions = 2048^2 cells on 256 CUs).  Across the iterations of a segment
//   * the residual r and z' = L p of the region stay in REGISTERS, the solution x in LDS (128 KB per workgroup);
//   * only the search direction p (ping-pong, read with its halo and written once), the 4 float off-diagonals and the
//     PERIMETER of r (edge rows / columns that neighbouring regions need to rebuild p on their halo) touch HBM:
//     ~4.7 words per cell and iteration instead of 11;
//   * the two global reductions of an iteration are two grid barriers (monotonic counter, agent-scope release / acquire)
//     that also carry the per-workgroup partial sums, reduced by everyone in a fixed order (deterministic).
// The arithmetic, its order and the stopping logic are those of cg_k1 / cg_k2 (same helper code paths); a segment starts
// from and ends in the global-memory state of the two-kernel path, so resets, the first iteration and grids that do not
// fit simply use cg_k1 / cg_k2.
#pragma once
#include "cg_kernels.h"

namespace piso {

constexpr int kPersistThreads = 512;            // 8 waves per CU = 2 per SIMD -> 256 VGPRs per lane: state in registers without spills
constexpr int kPersistWaves = kPersistThreads / 64;
constexpr int kPersistRegions = 2;              // regions per wave
constexpr int kPersistRegionsPerWg = kPersistWaves * kPersistRegions;

struct PersistCtl {
  unsigned* bar;        // monotonic arrival counter (zeroed before every launch)
  int* err;             // set to 1 if a spin gave up
  int nreg, ntx;        // regions (= waves with work), strips per row
  unsigned long long* timing;   // diagnostics (PISO_CG_PERSIST_TIMING): [4][grid] 100 MHz ticks in phase A / barrier A / phase B / barrier B
};

// barrier + exchange of 3 partial sums per workgroup; returns the totals (fixed summation order) in every thread
template <typename T>
__device__ __forceinline__ bool grid_exchange(const PersistCtl& c, T* __restrict__ gparts, T (&v)[3], unsigned target,
                                              T* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // workgroup partial: wave sums -> LDS -> thread 0
#pragma unroll
  for (int q = 0; q < 3; ++q) v[q] = wave_sum(v[q]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q) smem[q * kPersistWaves + wave] = v[q];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores before the barrier
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      T s = 0;
      for (int w = 0; w < kPersistWaves; ++w) s += smem[q * kPersistWaves + w];
      gparts[q * kMaxPartials + blockIdx.x] = s;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(c.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(c.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 24)) { *c.err = 1; ok = false; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  // every wave-0 lane sums a slice of the records, then the wave combines (same order in every workgroup)
  if (wave == 0) {
    T s[3] = {0, 0, 0};
    for (int b = lane; b < (int)gridDim.x; b += 64) {
#pragma unroll
      for (int q = 0; q < 3; ++q) s[q] += gparts[q * kMaxPartials + b];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) s[q] = wave_sum(s[q]);
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q) smem[3 * kPersistWaves + q] = s[q];
      smem[3 * kPersistWaves + 3] = ok ? (T)0 : (T)1;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 3; ++q) v[q] = smem[3 * kPersistWaves + q];
  const bool all_ok = smem[3 * kPersistWaves + 3] == (T)0;
  __syncthreads();
  return all_ok;
}

// ---- buffer addressing: a 128-bit descriptor per array in SGPRs, one per-lane byte offset in a VGPR, the row offset in an
// SGPR.  Keeps the address arithmetic of 32 rows x 9 arrays out of the vector registers (which hold the solver state).
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
template <typename S, int V>
__device__ __forceinline__ Vec<S, V> bld(rsrc_t r, unsigned voff, unsigned soff) {
  Vec<S, V> o;
  constexpr int B = sizeof(S) * V;
  static_assert(B == 16 || B == 8, "16- or 8-byte lane accesses");
  if constexpr (B == 16) {
    const auto t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    __builtin_memcpy(&o, &t, 16);
  } else {
    const auto t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    __builtin_memcpy(&o, &t, 8);
  }
  return o;
}
template <typename S>
__device__ __forceinline__ S bld1(rsrc_t r, unsigned voff, unsigned soff) {
  S o;
  if constexpr (sizeof(S) == 8) {
    const auto t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    __builtin_memcpy(&o, &t, 8);
  } else {
    const auto t = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
    __builtin_memcpy(&o, &t, 4);
  }
  return o;
}
template <typename S, int V>
__device__ __forceinline__ void bst(rsrc_t r, unsigned voff, unsigned soff, const Vec<S, V>& v) {
  static_assert(sizeof(S) * V == 16, "16-byte lane stores");
  __attribute__((ext_vector_type(4))) unsigned int t;
  __builtin_memcpy(&t, &v, 16);
  __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, soff, 0);
}
template <typename S>
__device__ __forceinline__ void bst1(rsrc_t r, unsigned voff, unsigned soff, S v) {
  if constexpr (sizeof(S) == 8) {
    __attribute__((ext_vector_type(2))) unsigned int t;
    __builtin_memcpy(&t, &v, 8);
    __builtin_amdgcn_raw_buffer_store_b64(t, r, voff, soff, 0);
  } else {
    unsigned int t;
    __builtin_memcpy(&t, &v, 4);
    __builtin_amdgcn_raw_buffer_store_b32(t, r, voff, soff, 0);
  }
}

// Host guarantees: nx % (64 V) == 0 (every lane of a strip has cells) and ny % R == 0 (every region has R rows).
template <typename T, typename CT, int R, bool RECON>
__global__ __launch_bounds__(kPersistThreads) void cg_persist(CgArgs<T> a, PersistCtl c, int k_begin, int k_end, int sv) {
  constexpr int V = 16 / sizeof(T);                        // 16-byte lane accesses
  constexpr int NQ = kPersistRegions;
  __shared__ T xs[kPersistRegionsPerWg * R * 64 * V];      // the solution of my regions (128 KB at R = 8, fp64)
  __shared__ T smem[4 * kPersistWaves];
  const int nx = a.nx, ny = a.ny;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: scalar branches, SGPRs
  int wg = blockIdx.x;                                     // XCD-contiguous bands (block b is observed on XCD b % 8)
  if (gridDim.x % kXcds == 0) wg = (blockIdx.x % kXcds) * (gridDim.x / kXcds) + blockIdx.x / kXcds;
  int j0[NQ], tx0[NQ];
  bool has[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int reg = (wg * kPersistWaves + wave) * NQ + q;
    has[q] = reg < c.nreg;
    const int ty = has[q] ? reg / c.ntx : 0;
    tx0[q] = has[q] ? reg - ty * c.ntx : 0;
    j0[q] = ty * R;
  }
  const unsigned nbytesT = (unsigned)((size_t)nx * ny * sizeof(T)), nbytesC = (unsigned)((size_t)nx * ny * sizeof(CT));
  const unsigned rowT = (unsigned)(nx * sizeof(T)), rowC = (unsigned)(nx * sizeof(CT));
  const rsrc_t Rr = make_rsrc(a.r, nbytesT), Rx = make_rsrc(a.x, nbytesT);
  const rsrc_t RoS = make_rsrc(a.oS, nbytesC), RoW = make_rsrc(a.oW, nbytesC), RoE = make_rsrc(a.oE, nbytesC), RoN = make_rsrc(a.oN, nbytesC);
  const rsrc_t RcC = make_rsrc(a.cC, nbytesT);
  const rsrc_t Rp0 = make_rsrc(a.p[0], nbytesT), Rp1 = make_rsrc(a.p[1], nbytesT);
  auto row_wrap = [&](int j, bool& valid) __attribute__((always_inline)) -> int {   // scalar: rows outside wrap or vanish
    valid = true;
    if (j < 0) { if (!a.per_y) valid = false; return ny - 1; }
    if (j >= ny) { if (!a.per_y) valid = false; return 0; }
    return j;
  };

  // ---- load the state of the two-kernel path: r of my regions into registers, x into LDS
  Vec<T, V> rr[NQ][R], zz[NQ][R];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const unsigned vT = (unsigned)((tx0[q] * 64 + lane) * V * sizeof(T));
    T* xl = xs + (size_t)(wave * NQ + q) * R * 64 * V + lane * V;
#pragma unroll
    for (int jj = 0; jj < R; ++jj) {
#pragma unroll
      for (int e = 0; e < V; ++e) { rr[q][jj].v[e] = 0; zz[q][jj].v[e] = 0; }
      if (has[q]) {
        rr[q][jj] = bld<T, V>(Rr, vT, (unsigned)(j0[q] + jj) * rowT);
        stv<T, V>(xl + jj * 64 * V, bld<T, V>(Rx, vT, (unsigned)(j0[q] + jj) * rowT));
      }
    }
  }
  CgState st = a.state[sv & 1];
  T pz = a.scal[SC_PZ], vs = a.scal[SC_VS], alpha = a.scal[SC_ALPHA];
  const T sc_c = a.scal[SC_C];
  // totals of the previous K2 (or previous segment): r.z', sum r, #cells with |r| >= accuracy
  T tB[3];
  {
    T s[3] = {0, 0, 0};
    if (wave == 0) {
      for (int b = lane; b < a.nB; b += 64) {
#pragma unroll
        for (int q = 0; q < 3; ++q) s[q] += a.partsB[q * kMaxPartials + b];
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) s[q] = wave_sum(s[q]);
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) smem[q] = s[q];
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 3; ++q) tB[q] = smem[q];
    __syncthreads();
  }

  unsigned epoch = 0;
  bool healthy = true;
  int k = k_begin;
  unsigned long long tacc[4] = {0, 0, 0, 0}, tlast = c.timing ? wall_clock64() : 0;
  auto tick = [&](int slot) __attribute__((always_inline)) {
    if (c.timing) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t = wall_clock64(); tacc[slot] += t - tlast; tlast = t; }
  };
  for (; k < k_end && healthy; ++k) {
    // ---- start of iteration k: stopping test of iteration k-1 (pressure_solve_op.cu.cc:312-335), beta (:351-352)
    if (k > 0 && (k % 5) == 0) {
      const int exceeded = tB[2] > 0;
      if (st.flag && !exceeded) { st.done = 1; st.iterations = k; }
      else st.flag = 1;
    }
    const rsrc_t Rpin = (k & 1) ? Rp1 : Rp0, Rpout = (k & 1) ? Rp0 : Rp1;
    if (st.done) {                                          // add the last direction to x and leave
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (has[q]) {
          const unsigned vT = (unsigned)((tx0[q] * 64 + lane) * V * sizeof(T));
          T* xl = xs + (size_t)(wave * NQ + q) * R * 64 * V + lane * V;
#pragma unroll
          for (int jj = 0; jj < R; ++jj) {
            Vec<T, V> xv = ldv<T, V>(xl + jj * 64 * V);
            const Vec<T, V> pq = bld<T, V>(Rpin, vT, (unsigned)(j0[q] + jj) * rowT);
#pragma unroll
            for (int e = 0; e < V; ++e) xv.v[e] = fma(alpha, pq.v[e], xv.v[e]);
            stv<T, V>(xl + jj * 64 * V, xv);
          }
        }
      break;
    }
    const T beta = -(tB[0] + vs * tB[1]) / pz;
    const T alpha_prev = alpha;

    // ---- phase A: x += alpha_prev p_old ; p_new = r + beta p_old (own rows from registers, halo from HBM) ; z' = L p_new
    T sA[3] = {0, 0, 0};                                   // sum p, p.r, p.z'
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (!has[q]) continue;
      const int cq = (tx0[q] * 64 + lane) * V;
      const unsigned vT = (unsigned)(cq * sizeof(T)), vC = (unsigned)(cq * sizeof(CT));
      T* xl = xs + (size_t)(wave * NQ + q) * R * 64 * V + lane * V;
      // p_new of one cell of another region's column (strip-edge neighbours), row j of MY region
      auto edge_val = [&](int j, int cc) __attribute__((always_inline)) -> T {
        const unsigned vo = (unsigned)(cc * sizeof(T)), so = (unsigned)j * rowT;
        return fma(beta, bld1<T>(Rpin, vo, so), bld1<T>(Rr, vo, so));
      };
      auto halo_row = [&](int j) __attribute__((always_inline)) -> Vec<T, V> {   // p_new on a row owned by another region
        Vec<T, V> o;
#pragma unroll
        for (int e = 0; e < V; ++e) o.v[e] = 0;
        bool valid;
        const int jw = row_wrap(j, valid);
        if (!valid) return o;
        o = bld<T, V>(Rr, vT, (unsigned)jw * rowT);
        const Vec<T, V> pq = bld<T, V>(Rpin, vT, (unsigned)jw * rowT);
#pragma unroll
        for (int e = 0; e < V; ++e) o.v[e] = fma(beta, pq.v[e], o.v[e]);
        return o;
      };
      // own row jj of the new direction; also performs x <- x + alpha_prev p_old (the axpy of iteration k-1, :303)
      auto own_row = [&](int jj, const Vec<T, V>& rreg) __attribute__((always_inline)) -> Vec<T, V> {
        Vec<T, V> o;
        const Vec<T, V> pq = bld<T, V>(Rpin, vT, (unsigned)(j0[q] + jj) * rowT);
        Vec<T, V> xv = ldv<T, V>(xl + jj * 64 * V);
#pragma unroll
        for (int e = 0; e < V; ++e) { o.v[e] = fma(beta, pq.v[e], rreg.v[e]); xv.v[e] = fma(alpha_prev, pq.v[e], xv.v[e]); }
        stv<T, V>(xl + jj * 64 * V, xv);
        return o;
      };
      Vec<T, V> behind = halo_row(j0[q] - 1);
      Vec<T, V> cur = own_row(0, rr[q][0]);
#pragma unroll
      for (int jj = 0; jj < R; ++jj) {
        const int j = j0[q] + jj;
        const unsigned sT = (unsigned)j * rowT, sC = (unsigned)j * rowC;
        Vec<T, V> ahead;
        if (jj + 1 < R) ahead = own_row(jj + 1 < R ? jj + 1 : jj, rr[q][jj + 1 < R ? jj + 1 : jj]);
        else ahead = halo_row(j0[q] + R);
        T left = __shfl_up(cur.v[V - 1], 1, kWave);
        T right = __shfl_down(cur.v[0], 1, kWave);
        if (lane == 0) {
          const int cc = cq - 1;
          left = (cc >= 0) ? edge_val(j, cc) : (a.per_x ? edge_val(j, nx - 1) : (T)0);
        }
        if (lane == 63) {
          const int cc = cq + V;
          right = (cc < nx) ? edge_val(j, cc) : (a.per_x ? edge_val(j, 0) : (T)0);
        }
        const Vec<CT, V> kS = bld<CT, V>(RoS, vC, sC), kW = bld<CT, V>(RoW, vC, sC), kE = bld<CT, V>(RoE, vC, sC),
                         kN = bld<CT, V>(RoN, vC, sC);
        Vec<T, V> kC;
        if constexpr (RECON) {
#pragma unroll
          for (int e = 0; e < V; ++e) {
            T d = 0;
            d -= (T)kS.v[e]; d -= (T)kN.v[e]; d -= (T)kW.v[e]; d -= (T)kE.v[e];
            kC.v[e] = d;
          }
        } else {
          kC = bld<T, V>(RcC, vT, sT);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const T pw = (e == 0) ? left : cur.v[e > 0 ? e - 1 : 0];
          const T pe = (e == V - 1) ? right : cur.v[e < V - 1 ? e + 1 : 0];
          T tmp = 0;                                        // summation order of calcZ_v4 (:81-90)
          tmp = fma((T)kS.v[e], behind.v[e], tmp);
          tmp = fma((T)kW.v[e], pw, tmp);
          tmp = fma(kC.v[e], cur.v[e], tmp);
          tmp = fma((T)kE.v[e], pe, tmp);
          tmp = fma((T)kN.v[e], ahead.v[e], tmp);
          zz[q][jj].v[e] = tmp;
          sA[0] += cur.v[e];
          sA[1] = fma(cur.v[e], rr[q][jj].v[e], sA[1]);
          sA[2] = fma(cur.v[e], tmp, sA[2]);
        }
        bst<T, V>(Rpout, vT, sT, cur);
        behind = cur;
        cur = ahead;
        if (jj & 1) __builtin_amdgcn_sched_barrier(0);       // at most two rows of loads in flight per region: bounds the VGPR pressure
      }
    }
    ++epoch;
    tick(0);
    healthy = grid_exchange<T>(c, a.partsA, sA, epoch * gridDim.x, smem);
    tick(1);
    if (!healthy) break;
    // ---- alpha (:301-302), then phase B: r -= alpha (z' + vs), partial sums, publish the perimeter of r
    vs = sc_c * sA[0];
    pz = sA[2] + vs * sA[0];
    alpha = (absval(pz) > 0) ? sA[1] / pz : (T)0;
    T sB[3] = {0, 0, 0};
    const T accuracy = (T)a.accuracy;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (!has[q]) continue;
      const unsigned vT = (unsigned)((tx0[q] * 64 + lane) * V * sizeof(T));
#pragma unroll
      for (int jj = 0; jj < R; ++jj) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const T rn = fma(-alpha, zz[q][jj].v[e] + vs, rr[q][jj].v[e]);
          rr[q][jj].v[e] = rn;
          sB[0] = fma(rn, zz[q][jj].v[e], sB[0]);
          sB[1] += rn;
          sB[2] += (absval(rn) < accuracy) ? (T)0 : (T)1;
        }
        const unsigned sT = (unsigned)(j0[q] + jj) * rowT;
        if (jj == 0 || jj == R - 1) {
          bst<T, V>(Rr, vT, sT, rr[q][jj]);                   // edge rows: whole row
        } else {
          if (lane == 0) bst1<T>(Rr, vT, sT, rr[q][jj].v[0]);                                   // edge columns
          if (lane == 63) bst1<T>(Rr, vT + (unsigned)((V - 1) * sizeof(T)), sT, rr[q][jj].v[V - 1]);
        }
      }
    }
    ++epoch;
    tick(2);
    healthy = grid_exchange<T>(c, a.partsB, sB, epoch * gridDim.x, smem);
    tick(3);
#pragma unroll
    for (int q = 0; q < 3; ++q) tB[q] = sB[q];
  }
  if (c.timing && threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) c.timing[q * gridDim.x + blockIdx.x] += tacc[q];
  }

  // ---- back to the global-memory state of the two-kernel path
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (has[q]) {
      const unsigned vT = (unsigned)((tx0[q] * 64 + lane) * V * sizeof(T));
      T* xl = xs + (size_t)(wave * NQ + q) * R * 64 * V + lane * V;
#pragma unroll
      for (int jj = 0; jj < R; ++jj) {
        const unsigned sT = (unsigned)(j0[q] + jj) * rowT;
        bst<T, V>(Rr, vT, sT, rr[q][jj]);
        bst<T, V>(Rx, vT, sT, ldv<T, V>(xl + jj * 64 * V));
      }
    }
  if (blockIdx.x == 0) {
    // the next launch (cg_k1 with do_check, or another segment) finds the last K2-totals in record 0 of partsB
    for (int b = threadIdx.x; b < a.nB; b += kPersistThreads) {
#pragma unroll
      for (int q = 0; q < 3; ++q) a.partsB[q * kMaxPartials + b] = (b == 0) ? tB[q] : (T)0;
    }
    if (threadIdx.x == 0) {
      a.scal[SC_PZ] = pz; a.scal[SC_VS] = vs; a.scal[SC_ALPHA] = alpha;
      a.state[0] = st; a.state[1] = st;
      if (!healthy) *c.err = 1;
    }
  }
}

}  // namespace piso
