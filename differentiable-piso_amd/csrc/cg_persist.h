// Persistent CG segment kernel: many NORMAL iterations in ONE launch, for grids whose solver state fits on chip
// (DESIGN.md 3.1 has the measurements behind every choice below).
//
// One workgroup of 512 threads per CU (2 waves per SIMD, 256 VGPRs per lane); every wave owns one region of 16 rows x 128
// columns (or two regions of 2 / 4 / 8 rows on smaller grids): 2048^2 cells = 2048 regions = 256 CUs x 8 waves.
// Across the iterations of a segment
//   * the residual r and the search direction p of the region stay in REGISTERS, the solution x in LDS (128 KB per workgroup);
//     z' = L p is never stored: phase A computes it for the dot products, phase B computes it again (same registers, same
//     instruction sequence, bitwise the same values) for the update of r;
//   * HBM sees only the float coefficient rows (S and W when the matrix is symmetric, else all four; streamed once per phase
//     through a circular 4-row software pipeline) and the PERIMETERS of r and p (first / last row and end columns of a region)
//     that neighbouring regions need to rebuild p_new = r + beta p on their halo - written write-through and read at agent
//     scope (sc1), the scope at which the 8 XCD-private L2s are coherent;
//   * the two global reductions of an iteration are two grid-wide EXCHANGES without atomics or fences: every workgroup
//     publishes three partial sums as tagged 8-byte words, wave 0 polls all records and adds them in a fixed order, so every
//     workgroup holds bitwise the same totals (grid_exchange below).
// The arithmetic per cell, its order and the stopping logic are those of cg_k1 / cg_k2; a segment starts from and ends in
// the global-memory state of the two-kernel path, so resets, the first iteration and grids that do not fit use cg_k1 / cg_k2.
#pragma once
#include "cg_kernels.h"

// scheduling fences of the two row loops (measured: with / without them the iteration time is the same; they keep the
// register allocation of the unrolled loops predictable)
#define PISO_SB_A1 __builtin_amdgcn_sched_barrier(0)
#define PISO_SB_A2 __builtin_amdgcn_sched_barrier(0)
#define PISO_SB_B1 __builtin_amdgcn_sched_barrier(0)
#define PISO_SB_B2 __builtin_amdgcn_sched_barrier(0)

namespace piso {

constexpr int kPersistThreads = 512;            // 8 waves per CU = 2 per SIMD -> 256 VGPRs per lane: state in registers without spills
constexpr int kPersistWaves = kPersistThreads / 64;
// region shapes (rows R x regions per wave NQ): 2 x 2, 4 x 2, 8 x 2 and 16 x 1 - at most 16 rows of 128 columns per wave

struct PersistCtl {
  unsigned long long* rec;   // exchange records: [2 (parity)][kPersistMaxGrid][8] 8-byte words, zeroed before every launch
  int* err;             // set to 1 if a spin gave up
  int nreg, ntx;        // regions (= waves with work), strips per row
  unsigned epoch0;      // tags of this launch's exchanges are epoch0 + 1, epoch0 + 2, ...: no record of an earlier launch can match
  unsigned long long* timing;   // diagnostics (-DPISO_PERSIST_DIAG + PISO_CG_PERSIST_TIMING): [5][grid] 100 MHz ticks per phase / exchange
  // XCD-local mode of cg_persist1 (small grids: all participating workgroups on ONE XCD, exchanges through that XCD's L2):
  int* xcd;             // [0..7] arrivals per XCD, [8] 1 + the XCD that runs the solve (0: not decided yet); zeroed before every launch
  int local_n;          // workgroups that take part (the launch has 8 x local_n: some XCD is dealt at least local_n of them)
};
#ifdef PISO_PERSIST_DIAG
constexpr bool kPersistDiag = true;     // per-phase clocks of wave 0 (PISO_CG_PERSIST_TIMING=1); costs a few registers
#else
constexpr bool kPersistDiag = false;
#endif
constexpr int kPersistMaxGrid = 256;   // workgroups (one per CU); the exchange keeps kPersistMaxGrid / 64 records per lane in registers
constexpr int kPersistMaxDepth = 4;     // coefficient rows in flight per wave: deeper spills registers, and spills cost more than latency (measured 3..16)

__device__ __forceinline__ double read_lane_c(double v, int src) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// a wave-uniform value moved to scalar registers (the VALU results of the reductions / divisions would otherwise occupy
// vector registers for the whole iteration; every VALU instruction can read one scalar operand directly)
template <typename S>
__device__ __forceinline__ S uniform(S v) {
  if constexpr (sizeof(S) == 8) {
    const unsigned long long b = (unsigned long long)__double_as_longlong((double)v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return (S)__longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  } else {
    return (S)__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)v)));
  }
}
// Wave-wide sum on the DPP network (no LDS traffic, ~4x shorter dependent chain than the ds_bpermute butterfly of wave_sum):
// quad swaps, half-row and row mirrors leave every lane with the sum of its row of 16; the four row sums are read into scalar
// registers and added in a fixed order.  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)b, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum_uniform(double v) {
  v += dpp_move<0xB1>(v);                                  // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);                                  // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);                                 // row_half_mirror
  v += dpp_move<0x140>(v);                                 // row_mirror
  return ((read_lane_c(v, 0) + read_lane_c(v, 16)) + read_lane_c(v, 32)) + read_lane_c(v, 48);
}

// Grid-wide exchange of 3 partial sums per workgroup that doubles as the grid barrier (measured 4.4 us for 256 workgroups
// against 11.3 us for "atomic counter + fence + read the partials", scripts/barrier_bench.hip).
//   * every workgroup publishes one 64-byte record: each double travels as two 8-byte words {32 payload bits | 32-bit epoch},
//     written and read with relaxed agent-scope atomics (single-copy atomic, coherent across the 8 XCDs' L2s);
//   * wave 0 of every workgroup polls all records until they carry the current epoch and adds them in a fixed order, so every
//     workgroup obtains bitwise the same totals - no counter, no fence, one memory round trip;
//   * records alternate between two arrays (epoch parity): a fast workgroup may publish epoch e+1 while a slow one still
//     reads epoch e, and nobody can reach e+2 before everybody has published e+1.
// DATA written before the exchange (p, the perimeter of r) is stored write-through at agent scope (sc1) and drained
// (s_waitcnt vmcnt(0)) by every wave before the workgroup publishes; readers load it at agent scope as well.
// KEEP = number of vector-memory LOADS this wave issued after its last store and may leave in flight (prefetch for the next
// phase; vmcnt retires in issue order, so "at most KEEP outstanding" means every store has completed).
// `between(rec)` runs in every wave while the exchange is in flight (wave 0: right after it has published the workgroup's
// record, before it starts polling): a record that carries the epoch also says "this workgroup's perimeter stores have
// completed", which phase A uses to fetch its halos from the neighbouring workgroups before the global sums are known.
struct NoBetween { __device__ __forceinline__ void operator()(const unsigned long long*) const {} };
template <typename T, int KEEP, bool HYBRID, typename F = NoBetween>
__device__ __forceinline__ bool grid_exchange(const PersistCtl& c, T (&v)[3], unsigned epoch, T* smem, F between = F()) {
  typedef unsigned long long u64;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  T* sm = smem + (epoch & 1) * 32;                          // parity double buffer: two __syncthreads per exchange
#pragma unroll
  for (int q = 0; q < 3; ++q) v[q] = (T)wave_sum_uniform((double)v[q]);
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q) sm[q * kPersistWaves + wave] = v[q];
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");   // this wave's write-through stores have completed
  __syncthreads();
  if (wave == 0) {
    u64* rec = c.rec + (size_t)(epoch & 1) * kPersistMaxGrid * 8;
    {
      // workgroup partial (every lane, broadcast LDS reads), then lanes 0..5 publish the six tagged words with ONE store
      T s[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        s[q] = 0;
        for (int w = 0; w < kPersistWaves; ++w) s[q] += sm[q * kPersistWaves + w];
      }
      const int vq = lane >> 1;
      const u64 bits = (u64)__double_as_longlong((double)(vq == 0 ? s[0] : (vq == 1 ? s[1] : s[2])));
      const u64 word = (lane & 1) ? ((bits & 0xffffffff00000000ull) | epoch) : (((bits & 0xffffffffull) << 32) | epoch);
      if (lane < 6) __hip_atomic_store(rec + (size_t)blockIdx.x * 8 + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    between(rec);
    // Lane l owns records l, l + 64, ...  Arrivals cluster: once a lane's first record is there, the others almost always
    // are too.  HYBRID polls the first record, then reads the remaining ones in one go and re-polls only stragglers (two
    // memory round trips instead of up to four) - 36 more registers, which the kernels with 16 rows of state per wave
    // do not have (a spill costs more than a round trip); those poll record after record.  Reading everything from the
    // start, or polling from several waves, multiplies the polling traffic that slows the publishing stores down
    // (scripts/barrier_bench.hip, variants 1 / 3 / 8 / 15 / 19).
    double tot[3] = {0, 0, 0};
    bool good = true;
    unsigned spins = 0;
    if constexpr (HYBRID) {
      constexpr int NR = kPersistMaxGrid / 64;
      u64 w[NR][6];
      bool ok[NR];
#pragma unroll
      for (int m = 0; m < NR; ++m) {
        ok[m] = (m * 64 + lane) >= (int)gridDim.x;
#pragma unroll
        for (int q = 0; q < 6; ++q) w[m][q] = 0;
      }
      while (true) {
        if (!ok[0]) {
#pragma unroll
          for (int q = 0; q < 6; ++q) w[0][q] = __hip_atomic_load(rec + (size_t)lane * 8 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bool o = true;
#pragma unroll
          for (int q = 0; q < 6; ++q) o = o && ((unsigned)(w[0][q] & 0xffffffffull) == epoch);
          ok[0] = o;
        }
        if (__all(ok[0])) break;
        if (++spins > (1u << 22)) { good = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      while (good) {
        bool all = true;
#pragma unroll
        for (int m = 1; m < NR; ++m)
          if (!ok[m]) {
#pragma unroll
            for (int q = 0; q < 6; ++q)
              w[m][q] = __hip_atomic_load(rec + (size_t)(m * 64 + lane) * 8 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
        for (int m = 1; m < NR; ++m) {
          if (!ok[m]) {
            bool o = true;
#pragma unroll
            for (int q = 0; q < 6; ++q) o = o && ((unsigned)(w[m][q] & 0xffffffffull) == epoch);
            ok[m] = o;
          }
          all = all && ok[m];
        }
        if (__all(all)) break;
        if (++spins > (1u << 22)) { good = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int m = 0; m < NR; ++m) {
        const bool active = (m * 64 + lane) < (int)gridDim.x;
#pragma unroll
        for (int q = 0; q < 3; ++q)
          tot[q] += active ? __longlong_as_double((long long)((w[m][2 * q] >> 32) | (w[m][2 * q + 1] & 0xffffffff00000000ull))) : 0.0;
      }
    } else {
      // sequential rounds: a lane holds one record at a time
      for (int m = 0; m < ((int)gridDim.x + 63) / 64; ++m) {
        const int b = m * 64 + lane;
        const bool active = b < (int)gridDim.x;
        u64 w[6] = {0, 0, 0, 0, 0, 0};
        bool ok = !active;
        while (true) {
          if (!ok) {
#pragma unroll
            for (int q = 0; q < 6; ++q) w[q] = __hip_atomic_load(rec + (size_t)b * 8 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = true;
#pragma unroll
            for (int q = 0; q < 6; ++q) ok = ok && ((unsigned)(w[q] & 0xffffffffull) == epoch);
          }
          if (__all(ok)) break;
          if (++spins > (1u << 22)) { good = false; break; }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q)
          tot[q] += active ? __longlong_as_double((long long)((w[2 * q] >> 32) | (w[2 * q + 1] & 0xffffffff00000000ull))) : 0.0;
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) tot[q] = wave_sum_uniform(tot[q]);
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q) sm[24 + q] = (T)tot[q];
      sm[27] = good ? (T)0 : (T)1;
      if (!good) *c.err = 1;
    }
  } else {
    between(c.rec + (size_t)(epoch & 1) * kPersistMaxGrid * 8);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 3; ++q) v[q] = uniform(sm[24 + q]);
  return uniform(sm[27]) == (T)0;
}

// ---- buffer addressing: a 128-bit descriptor per array in SGPRs, one per-lane byte offset in a VGPR, the row offset in an
// SGPR.  Keeps the address arithmetic of 32 rows x 9 arrays out of the vector registers (which hold the solver state).
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
// cache policy of a buffer access (gfx940+ encoding of the intrinsics' aux operand): kAgent = sc1, the scope at which the
// L2s of the 8 XCDs are coherent - used for everything one workgroup writes and another reads inside a launch.
constexpr int kPlain = 0, kAgent = 16;
template <typename S, int V, int AUX = kPlain>
__device__ __forceinline__ Vec<S, V> bld(rsrc_t r, unsigned voff, unsigned soff) {
  Vec<S, V> o;
  constexpr int B = sizeof(S) * V;
  static_assert(B == 16 || B == 8, "16- or 8-byte lane accesses");
  if constexpr (B == 16) {
    const auto t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
    __builtin_memcpy(&o, &t, 16);
  } else {
    const auto t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
    __builtin_memcpy(&o, &t, 8);
  }
  return o;
}
template <typename S, int AUX = kPlain>
__device__ __forceinline__ S bld1(rsrc_t r, unsigned voff, unsigned soff) {
  S o;
  if constexpr (sizeof(S) == 8) {
    const auto t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
    __builtin_memcpy(&o, &t, 8);
  } else {
    const auto t = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX);
    __builtin_memcpy(&o, &t, 4);
  }
  return o;
}
// 16-byte stores carry the row offset in the VECTOR offset, not in an SGPR.  A buffer store of more than 8 bytes reads its data
// registers for some cycles after it has issued; a VALU instruction that overwrites one of them right behind it replaces what
// the LAST lanes store (seen: the low dword of lanes 12-15 of every row of 16, on the second wave of a SIMD, when the memory
// pipeline was busy).  The compiler pads that hazard for stores WITHOUT a register soffset only (it assumes the form with an
// SGPR soffset is free of it - not so on gfx950: `buffer_store_dwordx4 v[150:153], v1, s[20:23], s0 offen` followed directly
// by `v_mov_b32 v150, v1` published perimeters with the low dword of a lane offset in them).  voff must be a real offset.
template <typename S, int V, int AUX = kPlain>
__device__ __forceinline__ void bst(rsrc_t r, unsigned voff, unsigned soff, const Vec<S, V>& v) {
  static_assert(sizeof(S) * V == 16, "16-byte lane stores");
  __attribute__((ext_vector_type(4))) unsigned int t;
  __builtin_memcpy(&t, &v, 16);
  __builtin_amdgcn_raw_buffer_store_b128(t, r, voff + soff, 0, AUX);
}
template <typename S, int AUX = kPlain>
__device__ __forceinline__ void bst1(rsrc_t r, unsigned voff, unsigned soff, S v) {
  if constexpr (sizeof(S) == 8) {
    __attribute__((ext_vector_type(2))) unsigned int t;
    __builtin_memcpy(&t, &v, 8);
    __builtin_amdgcn_raw_buffer_store_b64(t, r, voff, soff, AUX);
  } else {
    unsigned int t;
    __builtin_memcpy(&t, &v, 4);
    __builtin_amdgcn_raw_buffer_store_b32(t, r, voff, soff, AUX);
  }
}

// value of lane `src` (compile-time constant after unrolling) as a wave-uniform scalar
template <typename S>
__device__ __forceinline__ S read_lane(S v, int src) {
  if constexpr (sizeof(S) == 8) {
    const unsigned long long b = (unsigned long long)__double_as_longlong((double)v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
    return (S)__longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  } else {
    return (S)__int_as_float(__builtin_amdgcn_readlane(__float_as_int((float)v), src));
  }
}
// lane l receives `v` of lane l - 1 (UP) or l + 1 (!UP) through the DPP wavefront shifts; the lane without a source keeps `edge`
template <bool UP, typename S>
__device__ __forceinline__ S shift_lane(S v, S edge) {
  constexpr int ctrl = UP ? 0x138 /* wave_shr:1 */ : 0x130 /* wave_shl:1 */;
  if constexpr (sizeof(S) == 8) {
    const unsigned long long b = (unsigned long long)__double_as_longlong((double)v), e = (unsigned long long)__double_as_longlong((double)edge);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)e, (int)(unsigned)b, ctrl, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(e >> 32), (int)(unsigned)(b >> 32), ctrl, 0xf, 0xf, false);
    return (S)__longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  } else {
    return (S)__int_as_float(__builtin_amdgcn_update_dpp(__float_as_int((float)edge), __float_as_int((float)v), ctrl, 0xf, 0xf, false));
  }
}

// Host guarantees: nx % (64 V) == 0 (every lane of a strip has cells) and ny % R == 0 (every region has R rows).
// SYM: the matrix is symmetric (verified bit for bit by cg_setup_coeffs): N of a cell is S of the cell above, E is W of the cell
// to the right - only the S and W arrays are streamed (8 instead of 16 coefficient bytes per cell).
template <typename T, typename CT, int R, int NQ, bool RECON, bool SYM>
__global__ __launch_bounds__(kPersistThreads) void cg_persist(CgArgs<T> a, PersistCtl c, int k_begin, int k_end, int sv, int pend) {
  constexpr int V = 16 / sizeof(T);                        // 16-byte lane accesses
  static_assert(R * NQ <= 16 && 2 * R <= 64, "at most 16 rows per wave; the edge columns of a region fit one wave-wide load");
  __shared__ T xs[kPersistWaves * NQ * R * 64 * V];        // the solution of my regions (128 KB at 16 rows per wave, fp64)
  __shared__ T smem[64];
  __shared__ int nbr_s[kPersistWaves * 8];                   // per wave: record slots of the (up to 4 NQ) neighbouring workgroups
  __shared__ int bad_s;                                      // a neighbour poll gave up (read after the exchange's last barrier)
  // p_new on the rows below / above my regions (rebuilt from the neighbours' perimeters in phase A, used by the first / last
  // row of both phases): parked in LDS, two reads per phase, instead of 8 registers held through both row loops
  constexpr bool kParkHalos = (NQ * R < 16) || NQ == 1;     // (two regions of 8 rows: x already fills the LDS)
  __shared__ T halo_s[kParkHalos ? kPersistWaves * NQ * 2 * 64 * V : 1];
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

  // ---- load the state of the two-kernel path: r and the search direction p of my regions into registers, x into LDS
  const T alpha0 = pend ? uniform(a.scal[SC_ALPHA]) : (T)0;   // pend: x still lacks alpha p of the iteration before k_begin
  Vec<T, V> rr[NQ][R], pp[NQ][R];
  unsigned vT[NQ];
  {
    const rsrc_t Rp = (k_begin & 1) ? Rp1 : Rp0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int cq = (tx0[q] * 64 + lane) * V;
      vT[q] = (unsigned)(cq * sizeof(T));
      T* xl = xs + (size_t)(wave * NQ + q) * R * 64 * V + lane * V;
#pragma unroll
      for (int jj = 0; jj < R; ++jj) {
#pragma unroll
        for (int e = 0; e < V; ++e) { rr[q][jj].v[e] = 0; pp[q][jj].v[e] = 0; }
        if (has[q]) {
          rr[q][jj] = bld<T, V>(Rr, vT[q], (unsigned)(j0[q] + jj) * rowT);
          pp[q][jj] = bld<T, V>(Rp, vT[q], (unsigned)(j0[q] + jj) * rowT);
          // (the two-kernel path defers x += alpha p of its last iteration to the next K1: applied here, on entry)
          Vec<T, V> xv = bld<T, V>(Rx, vT[q], (unsigned)(j0[q] + jj) * rowT);
#pragma unroll
          for (int e = 0; e < V; ++e) xv.v[e] = fma(alpha0, pp[q][jj].v[e], xv.v[e]);
          stv<T, V>(xl + jj * 64 * V, xv);
        }
      }
    }
  }
  CgState st = a.state[sv & 1];
  T pz = uniform(a.scal[SC_PZ]), vs = uniform(a.scal[SC_VS]), alpha = uniform(a.scal[SC_ALPHA]);
  const T sc_c = uniform(a.scal[SC_C]);
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
    for (int q = 0; q < 3; ++q) tB[q] = uniform(smem[q]);
    __syncthreads();
  }

  // ---- coefficient pipeline: both phases of an iteration stream the coefficient rows of my regions in the same order;
  // the loads of row t + D are issued when row t has been consumed, CIRCULARLY (the last D steps of a phase issue rows
  // 0 .. D-1 for the next phase, which then travel while the grid exchange leaves the memory system idle).
  constexpr int coef_regs = ((SYM ? 2 : 4) * (int)sizeof(CT) * V + (RECON ? 0 : (int)sizeof(T) * V)) / 4;   // VGPRs of a row in flight
  constexpr int NT = NQ * R;
  constexpr int budget = (NQ == 1 || NT < 16) ? 16 : 8;    // VGPRs for rows in flight (two regions keep twice the halo state)
  constexpr int Dw = budget / coef_regs < 2 ? 2 : (budget / coef_regs > kPersistMaxDepth ? kPersistMaxDepth : budget / coef_regs);
  constexpr int D = (NT >= Dw) ? Dw : NT;
  constexpr bool kFetchEarly = (NQ == 1) || NT <= 8;                  // halos fetched during exchange B (20 registers per region)
  constexpr bool kHybridPoll = NT <= 8;                                // (see grid_exchange: a question of registers)
  constexpr int kBaseLoads = (SYM ? 2 : 4) + (RECON ? 0 : 1);          // vector loads every row issues (some rows one or two more)
  Vec<CT, V> cS[NT], cW[NT], cE[NT], cN[NT], cSh[NQ];
  Vec<T, V> cD[NT];
  CT eW[NQ];
  // (the per-lane byte offset into a coefficient row is recomputed from vT at every use - one shift - instead of living in a
  // register for the whole kernel: the empty asm keeps the optimiser from hoisting it back into one)
  auto coef_offset = [&](int q) __attribute__((always_inline)) -> unsigned {
    unsigned o = vT[q];
    asm volatile("" : "+v"(o));
    return (unsigned)((unsigned long long)o * sizeof(CT) / sizeof(T));
  };
  auto issue_coef = [&](int t) __attribute__((always_inline)) {
    const int q = t / R, jj = t - q * R;
    const unsigned vCq = coef_offset(q);
    const unsigned sT = (unsigned)(j0[q] + jj) * rowT, sC = (unsigned)(j0[q] + jj) * rowC;
    cS[t] = bld<CT, V>(RoS, vCq, sC); cW[t] = bld<CT, V>(RoW, vCq, sC);
    if constexpr (!SYM) { cE[t] = bld<CT, V>(RoE, vCq, sC); cN[t] = bld<CT, V>(RoN, vCq, sC); }
    if constexpr (!RECON) cD[t] = bld<T, V>(RcC, vT[q], sT);
  };
  // SYM: W of the first column of the strip to the right (E of my last column; lane R + jj: row jj) and S of the row above the
  // region (N of my last row): constants of the launch, loaded once (not with rows 0 / R-1 of every pass)
  if constexpr (SYM) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int side = lane / R, er = lane - side * R;
      int cc = (tx0[q] + 1) * 64 * V;
      if (cc >= nx) cc = a.per_x ? 0 : -1;
      const unsigned vo = (has[q] && side == 1 && cc >= 0) ? (unsigned)(j0[q] + er) * rowC + (unsigned)(cc * sizeof(CT)) : 0xffffffffu;
      eW[q] = bld1<CT>(RoW, vo, 0);
      bool valid;
      const int jw = row_wrap(j0[q] + R, valid);
      cSh[q] = bld<CT, V>(RoS, (has[q] && valid) ? coef_offset(q) : 0xffffffffu, (unsigned)jw * rowC);
    }
  }
  // p_new on the cells around a region, rebuilt from what the neighbours published (perimeters of r and of the old p):
  // pnb / pna = the rows below / above, edge = the two columns next to the strip (lane l < R: left neighbour of row l,
  // lane R + l: right neighbour).  Kept from phase A to phase B.
  T edge[NQ];
  Vec<T, V> pnb[NQ], pna[NQ];                               // (registers when the LDS has no room: !kParkHalos)
  // z' = L p of row t of my regions: summation order of calcZ_v4 (pressure_solve_op.cu.cc:81-90).  Phase A and phase B
  // both call this on the same registers, so they see bitwise the same z'.
  auto zrow = [&](int t) __attribute__((always_inline)) -> Vec<T, V> {
    const int q = t / R, jj = t - q * R;
    T* hs = halo_s + (kParkHalos ? (size_t)((wave * NQ + q) * 2) * 64 * V + lane * V : 0);
    const Vec<T, V> behind = (jj > 0) ? pp[q][jj > 0 ? jj - 1 : 0] : (kParkHalos ? ldv<T, V>(hs) : pnb[q]);
    const Vec<T, V> cur = pp[q][jj];
    const Vec<T, V> ahead = (jj + 1 < R) ? pp[q][jj + 1 < R ? jj + 1 : jj] : (kParkHalos ? ldv<T, V>(hs + (kParkHalos ? 64 * V : 0)) : pna[q]);
    const T left = shift_lane<true, T>(cur.v[V - 1], read_lane<T>(edge[q], jj));
    const T right = shift_lane<false, T>(cur.v[0], read_lane<T>(edge[q], R + jj));
    Vec<CT, V> kN, kE;
    if constexpr (SYM) {
      kN = (jj + 1 < R) ? cS[t + 1 < NT ? t + 1 : t] : cSh[q];
#pragma unroll
      for (int e = 0; e + 1 < V; ++e) kE.v[e] = cW[t].v[e + 1];
      kE.v[V - 1] = shift_lane<false, CT>(cW[t].v[0], read_lane<CT>(eW[q], R + jj));
    } else {
      kN = cN[t]; kE = cE[t];
    }
    Vec<T, V> kC, z;
    if constexpr (RECON) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        T d = 0;
        d -= (T)cS[t].v[e]; d -= (T)kN.v[e]; d -= (T)cW[t].v[e]; d -= (T)kE.v[e];
        kC.v[e] = d;
      }
    } else {
      kC = cD[t];
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const T pw = (e == 0) ? left : cur.v[e > 0 ? e - 1 : 0];
      const T pe = (e == V - 1) ? right : cur.v[e < V - 1 ? e + 1 : 0];
      T tmp = 0;
      tmp = fma((T)cS[t].v[e], behind.v[e], tmp);
      tmp = fma((T)cW[t].v[e], pw, tmp);
      tmp = fma(kC.v[e], cur.v[e], tmp);
      tmp = fma((T)kE.v[e], pe, tmp);
      tmp = fma((T)kN.v[e], ahead.v[e], tmp);
      z.v[e] = tmp;
    }
    return z;
  };
  // perimeter of row jj of region q (what neighbouring regions read): the whole first / last row, else the two end cells
  auto publish = [&](rsrc_t Rd, int q, int jj, const Vec<T, V>& val) __attribute__((always_inline)) {
    const unsigned sT = (unsigned)(j0[q] + jj) * rowT;
    if (jj == 0 || jj == R - 1) {
      bst<T, V, kAgent>(Rd, vT[q], sT, val);
    } else {
      if (lane == 0) bst1<T, kAgent>(Rd, vT[q], sT, val.v[0]);
      if (lane == 63) bst1<T, kAgent>(Rd, vT[q] + (unsigned)((V - 1) * sizeof(T)), sT, val.v[V - 1]);
    }
  };
  // ---- halos: what the neighbours published (perimeters of r and of the direction): the columns next to the strip (all R
  // rows with ONE pair of loads: lane l < R the left neighbour of row l, lane R + l the right one; lanes without a cell and
  // walls read out of range -> 0) and the rows below / above the region.  Issued either at the top of phase A or - normally -
  // by fetch_halos() in the middle of the previous exchange B, as soon as the NEIGHBOURING workgroups have published.
  T eP[NQ], eR[NQ];
  Vec<T, V> hbR[NQ], hbP[NQ], haR[NQ], haP[NQ];
  bool have_halos = false;
  auto issue_halos = [&](rsrc_t Rp) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int side = lane / R, er = lane - side * R;
      int cc = (side == 0) ? tx0[q] * 64 * V - 1 : (tx0[q] + 1) * 64 * V;
      if (cc < 0) cc = a.per_x ? nx - 1 : -1;
      else if (cc >= nx) cc = a.per_x ? 0 : -1;
      const unsigned vo = (side < 2 && cc >= 0) ? (unsigned)(j0[q] + er) * rowT + (unsigned)(cc * sizeof(T)) : 0xffffffffu;
      eP[q] = bld1<T, kAgent>(Rp, vo, 0);
      eR[q] = bld1<T, kAgent>(Rr, vo, 0);
      bool vb, va;
      const int jb = row_wrap(j0[q] - 1, vb), ja = row_wrap(j0[q] + R, va);
      const unsigned hb = vb ? vT[q] : 0xffffffffu, ha = va ? vT[q] : 0xffffffffu;   // beyond a wall: out of range -> 0
      hbR[q] = bld<T, V, kAgent>(Rr, hb, (unsigned)jb * rowT);
      hbP[q] = bld<T, V, kAgent>(Rp, hb, (unsigned)jb * rowT);
      haR[q] = bld<T, V, kAgent>(Rr, ha, (unsigned)ja * rowT);
      haP[q] = bld<T, V, kAgent>(Rp, ha, (unsigned)ja * rowT);
    }
  };
  // lane l < 4 NQ: the workgroup (record slot) that owns the region below / above / left / right of my region l / 4, or -1
  // (no neighbour there, or my own workgroup - whose stores are complete once the exchange's first barrier has passed)
  int nbr_slot = -1;
  if (has[0] && lane < 4 * NQ) {
    const int q = lane >> 2, dir = lane & 3;
    const int nty = ny / R;
    int ty = 0, tx = 0;
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) if (q == qq) { ty = j0[qq] / R; tx = tx0[qq]; }   // (wave-uniform arrays, selected per lane)
    int y = ty + (dir == 0 ? -1 : (dir == 1 ? 1 : 0)), x = tx + (dir == 2 ? -1 : (dir == 3 ? 1 : 0));
    bool exists = true;
    if (y < 0) { exists = a.per_y; y = nty - 1; }
    if (y >= nty) { exists = a.per_y; y = 0; }
    if (x < 0) { exists = exists && a.per_x; x = c.ntx - 1; }
    if (x >= c.ntx) { exists = exists && a.per_x; x = 0; }
    if (exists) {
      const int lw = (y * c.ntx + x) / (kPersistWaves * NQ);           // logical workgroup of that region
      int b = lw;                                                       // inverse of the XCD permutation above
      if (gridDim.x % kXcds == 0) { const int per = gridDim.x / kXcds; b = (lw % per) * kXcds + lw / per; }
      if (b != (int)blockIdx.x) nbr_slot = b;
    }
  }
  if (lane < 8) nbr_s[wave * 8 + lane] = nbr_slot;           // (parked in LDS: read once per iteration, not worth a register)
  if (threadIdx.x == 0) bad_s = 0;
  unsigned fetch_epoch = 0;
  rsrc_t fetch_rp = Rp0;
  auto fetch_halos = [&](const unsigned long long* rec) __attribute__((always_inline)) {
    if (!has[0]) return;
    unsigned spins = 0;
    const int slot = nbr_s[wave * 8 + (lane & 7)];
    const bool mine = lane < 4 * NQ && slot >= 0;
    while (true) {
      bool ok = true;
      if (mine)
        ok = (unsigned)(__hip_atomic_load(rec + (size_t)slot * 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffffffull) == fetch_epoch;
      if (__all(ok)) break;
      if (++spins > (1u << 22)) {                           // same bound as the global exchange; a give-up FAILS the launch:
        if (lane == 0) { bad_s = 1; *c.err = 1; }           // the halos below would be stale (the host restarts on cg_k1 / cg_k2)
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    issue_halos(fetch_rp);
  };

  if (has[0]) {
#pragma unroll
    for (int t = 0; t < D; ++t) issue_coef(t);
  }

  unsigned epoch = c.epoch0;
  bool healthy = true;
  int k = k_begin;
  unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tlast = (kPersistDiag && c.timing) ? wall_clock64() : 0;
  auto tick = [&](int slot) __attribute__((always_inline)) {
    if (kPersistDiag && c.timing) { const unsigned long long t = wall_clock64(); tacc[slot] += t - tlast; tlast = t; }   // (scalar registers only)
  };
  for (; k < k_end && healthy; ++k) {
    // ---- start of iteration k: stopping test of iteration k-1 (pressure_solve_op.cu.cc:312-335), beta (:351-352)
    if (!st.done && k > 0 && (k % 5) == 0) {               // (!done: a launch queued behind a converged one changes nothing)
      const int exceeded = tB[2] > 0;
      if (st.flag && !exceeded) { st.done = 1; st.iterations = k; }
      else st.flag = 1;
    }
    const rsrc_t Rpin = (k & 1) ? Rp1 : Rp0, Rpout = (k & 1) ? Rp0 : Rp1;
    if (st.done) break;                                    // (x already holds every direction: phase B adds alpha p at once)
    const T beta = uniform(-(tB[0] + vs * tB[1]) / pz);

    // ---- phase A: p_new = r + beta p_old (registers) ; z' = L p_new ; sums p, p.r, p.z'
    T sA[3] = {0, 0, 0};
    if (has[0]) {                                          // the host makes nreg a multiple of NQ: a wave owns NQ regions or none
      if (!have_halos) issue_halos(Rpin);                  // (first iteration of a launch; later ones were fetched in exchange B)
      // meanwhile, on chip: the new direction; its perimeter goes out for iteration k+1
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int jj = 0; jj < R; ++jj) {
#pragma unroll
          for (int e = 0; e < V; ++e) pp[q][jj].v[e] = fma(beta, pp[q][jj].v[e], rr[q][jj].v[e]);
          publish(Rpout, q, jj, pp[q][jj]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        edge[q] = fma(beta, eP[q], eR[q]);
        Vec<T, V> below, above;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          below.v[e] = fma(beta, hbP[q].v[e], hbR[q].v[e]);
          above.v[e] = fma(beta, haP[q].v[e], haR[q].v[e]);
        }
        if constexpr (kParkHalos) {
          T* hs = halo_s + (size_t)((wave * NQ + q) * 2) * 64 * V + lane * V;
          stv<T, V>(hs, below);
          stv<T, V>(hs + 64 * V, above);
        } else {
          pnb[q] = below; pna[q] = above;
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int q = t / R, jj = t - q * R;
        const Vec<T, V> z = zrow(t);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          sA[0] += pp[q][jj].v[e];
          sA[1] = fma(pp[q][jj].v[e], rr[q][jj].v[e], sA[1]);
          sA[2] = fma(pp[q][jj].v[e], z.v[e], sA[2]);
        }
        PISO_SB_A1;
        if constexpr (D < NT) issue_coef(t + D < NT ? t + D : t + D - NT);   // wraps: rows 0 .. D-1 again, for phase B
        PISO_SB_A2;
      }
    }
    ++epoch;
    tick(0);
    // (every store of this phase was issued before NT rows of coefficient loads: at most D rows may stay in flight)
    healthy = grid_exchange<T, (D < NT) ? D * kBaseLoads : 0, kHybridPoll>(c, sA, epoch, smem);
    tick(1);
    if (!healthy) break;
    // ---- alpha (:301-302), then phase B: z' again, x += alpha p, r -= alpha (z' + vs), sums, publish the perimeter of r
    vs = uniform(sc_c * sA[0]);
    pz = uniform(sA[2] + vs * sA[0]);
    alpha = uniform((absval(pz) > 0) ? sA[1] / pz : (T)0);
    T sB[3] = {0, 0, 0};
    const T accuracy = uniform((T)a.accuracy);
    if (has[0]) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int q = t / R, jj = t - q * R;
        T* xl = xs + (size_t)(wave * NQ + q) * R * 64 * V + lane * V + jj * 64 * V;
        Vec<T, V> xv = ldv<T, V>(xl);                        // x += alpha p (:303); the LDS latency hides under the stencil
        const Vec<T, V> z = zrow(t);
#pragma unroll
        for (int e = 0; e < V; ++e) xv.v[e] = fma(alpha, pp[q][jj].v[e], xv.v[e]);
        stv<T, V>(xl, xv);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const T rn = fma(-alpha, z.v[e] + vs, rr[q][jj].v[e]);
          rr[q][jj].v[e] = rn;
          sB[0] = fma(rn, z.v[e], sB[0]);
          sB[1] += rn;
          sB[2] += (absval(rn) < accuracy) ? (T)0 : (T)1;
        }
        publish(Rr, q, jj, rr[q][jj]);
        PISO_SB_B1;
        if constexpr (D < NT) issue_coef(t + D < NT ? t + D : t + D - NT);   // wraps: rows 0 .. D-1 for phase A of the next iteration
        PISO_SB_B2;
      }
    }
    ++epoch;
    tick(2);
    // (the last row's perimeter store is followed by exactly one row of coefficient loads)
    if constexpr (kFetchEarly) {
      fetch_epoch = epoch;
      fetch_rp = Rpout;                                    // iteration k+1 reads the direction this iteration published
      healthy = grid_exchange<T, (D < NT) ? kBaseLoads : 0, kHybridPoll>(c, sB, epoch, smem, fetch_halos);
      healthy = healthy && (__builtin_amdgcn_readfirstlane(bad_s) == 0);
      have_halos = true;
    } else {
      healthy = grid_exchange<T, (D < NT) ? kBaseLoads : 0, kHybridPoll>(c, sB, epoch, smem);
    }
    tick(3);
#pragma unroll
    for (int q = 0; q < 3; ++q) tB[q] = uniform(sB[q]);
  }
  if (kPersistDiag && c.timing && threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 5; ++q) c.timing[q * gridDim.x + blockIdx.x] += tacc[q];
  }

  // ---- back to the global-memory state of the two-kernel path (iteration k reads its direction from p[k & 1])
  {
    const rsrc_t Rp = (k & 1) ? Rp1 : Rp0;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (has[q]) {
        T* xl = xs + (size_t)(wave * NQ + q) * R * 64 * V + lane * V;
#pragma unroll
        for (int jj = 0; jj < R; ++jj) {
          const unsigned sT = (unsigned)(j0[q] + jj) * rowT;
          bst<T, V>(Rr, vT[q], sT, rr[q][jj]);
          bst<T, V>(Rp, vT[q], sT, pp[q][jj]);
          bst<T, V>(Rx, vT[q], sT, ldv<T, V>(xl + jj * 64 * V));
        }
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
