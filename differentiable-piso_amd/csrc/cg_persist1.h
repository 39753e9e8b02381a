// Persistent CG segment kernel: many NORMAL iterations in ONE launch with ONE grid-wide exchange per iteration, for grids whose
// solver state fits on chip (DESIGN.md 3.1 has the measurements behind every choice below; the first persistent kernel of rounds
// 1-2 needed two exchanges).
//
// r, p in registers, x in LDS, float coefficient rows streamed through a circular software pipeline, one workgroup of 512 threads
// per CU; per-cell arithmetic as the two-kernel path (zrow: summation order of calcZ_v4, pressure_solve_op.cu.cc:81-90).  How the
// two dependent reductions of a CG iteration are obtained from one exchange:
//
//   D(k)  p_k = r_k + beta_k p_{k-1};  z' = L p_k;  local sums  S p, p.r, p.z', r.z', z'.z', S z'
//         the PERIMETER of z' is published (sc1 stores, ping-pong buffer k & 1)
//   E(k)  one exchange of 8 sums per workgroup: the six above and  S r_k, #{|r_k| >= accuracy}  left over from U(k-1)
//         -> alpha_k = p.r / (p.z' + vs S p)                                   (pressure_solve_op.cu.cc:301-302)
//            r_{k+1}.z'_k = r.z' - alpha (z'.z' + vs S z'),  S r_{k+1} = S r_k - alpha (S z' + N vs)    [r_{k+1} = r_k - alpha (z' + vs)]
//         -> beta_{k+1} = -(r_{k+1}.z'_k + vs S r_{k+1}) / pz                  (:351-352) without a second reduction
//         -> the stopping test of iteration k (every 5th, device-flag cadence of :312-335) on #{|r_k| >= accuracy}: it is
//            evaluated one exchange later than in cg_persist, BEFORE x and r move on, so a converged solve stops in exactly
//            the reference's state (x_k, iterations = k)
//   U(k)  z' = L p_k again (same registers, same instructions: bitwise the same);  x += alpha p_k;  r -= alpha (z' + vs);
//         local  S r_{k+1}, #{|r_{k+1}| >= accuracy}
//
// Every region keeps copies of r and p on the ring of cells around it and advances them with the SAME fma's as their
// owners (bitwise equal), using the published perimeter of z' - so one vector's perimeter crosses the fabric per iteration
// instead of two, and nothing has to be visible between D and U of one iteration: one exchange.
// Round 4: the exchange is a tree that follows the hardware (workgroup -> XCD leader through that XCD's L2 -> everybody: every wave
// polls the eight XCD records itself, no second barrier; grid_exchange8_hier), the end cells of a region's rows are published as
// one packed block per region (kPack: the L2 was full of single-cell cache lines), z' of U's first rows is computed while the
// exchange's records travel (kAhead), and whatever only the rare paths need is read again from the kernarg segment (karg).
// The one-step recurrences start from directly summed quantities in every iteration (no drift); they replace two dot
// products by algebraically equal expressions, so iterates differ from cg_k1 / cg_k2 at round-off level (like any two
// summation orders); converged answers and the stopping cadence are the reference's.  A segment starts from and ends in the
// global-memory state of the two-kernel path, exactly like cg_persist.
#pragma once
#include <cstddef>
#include <type_traits>

#include "cg_persist.h"
#include "peer.h"

namespace piso {

// SLAB = true: the kernel works on ONE y-slab of a grid that is cut over the GPUs of a node (cg_slab.hip).  What changes:
//   * the rows just below / above the slab belong to the neighbouring GPU: the edge regions publish their first / last row of z'
//     ALSO into that neighbour's mailbox (peer-mapped memory, system-scope stores over xGMI) and read the neighbour's row from
//     their own mailbox; r, p[] and x carry one halo row below (row -1) and above (row ny) as in the two-kernel slab path - the
//     ring copies start from them and are written back to them when the segment ends;
//   * the exchange's second level crosses the node: the XCD leaders store their records into every rank's mailbox, wave w of every
//     workgroup adds rank w's records, the rank totals meet in LDS (bitwise the same totals on every GPU, so every GPU takes the
//     same decisions; grid_exchange8_hier<..., XG>);
//   * N of the slab's last row comes from the N array (its S twin lives on the neighbour), sums of the previous K2 from a.gB.
struct NoSlab {};
// region shape of the persistent kernels for an nx x ny grid (V cells per lane, `cus` compute units): one region of 16 rows per
// wave has the smallest halo overhead and is taken when it keeps at least 3/4 of the waves busy (or when forced); else two
// regions of 2 / 4 rows per wave (two regions of 8 rows do not fit the registers: such shapes - ny a multiple of 8 but not of 16 on
// a grid too large for 4-row regions - iterate on the two-kernel path).  R = 0: the grid cannot be tiled (two-kernel iteration).
struct PersistShape { int R = 0, NQ = 0, nreg = 0, ntx = 0, grid = 0; };
inline PersistShape persist_shape(int nx, int ny, int V, int cus, int force_r) {
  PersistShape s;
  if (nx % (64 * V) != 0) return s;                         // every lane of a strip has cells
  const int ntx = nx / (64 * V);
  if (ny % 16 == 0 && (force_r <= 0 || force_r == 16)) {
    const long long nreg = (long long)ntx * (ny / 16);
    if (nreg <= (long long)cus * kPersistWaves && (force_r > 0 || 4 * nreg >= 3LL * cus * kPersistWaves)) {
      s.R = 16; s.NQ = 1; s.nreg = (int)nreg; s.ntx = ntx;
      s.grid = (int)((nreg + kPersistWaves - 1) / kPersistWaves);
    }
  }
  for (int R : {2, 4}) {
    if (s.R) break;
    if (force_r > 0 && force_r != R) continue;
    if (ny % R != 0) continue;                              // every region has R rows
    const long long nreg = (long long)ntx * (ny / R);
    if (nreg % 2 == 0 && nreg <= (long long)cus * kPersistWaves * 2) {   // a wave owns 2 regions or none
      s.R = R; s.NQ = 2; s.nreg = (int)nreg; s.ntx = ntx;
      s.grid = (int)((nreg + kPersistWaves * 2 - 1) / (kPersistWaves * 2));
    }
  }
  return s;
}
struct SlabCtl {
  PeerView pv;
  double ncells;           // cells of the GLOBAL grid
  char *rows_own, *rows_lo, *rows_hi;   // the row areas (PeerLayout::kRows) of my mailbox and of the lower / upper neighbour's
  unsigned hop_ticks;      // measurements only (option slab_hop_ticks): the XCD leaders' records leave this many 10 ns ticks late
};
// A kernel argument read AGAIN from the kernarg segment (scalar loads through a pointer the optimiser cannot see through).  The
// row loops of the persistent kernels are bound by VALU issue and short of scalar registers: whatever only the rare paths need -
// the mailbox addresses of the two edge waves of a slab, the pointers of the exit block - is fetched where it is used instead of
// living in SGPRs across the loop (a spilled SGPR comes back through v_readlane, a VALU slot; an s_load costs none).
// Persist1Kargs mirrors the argument list of cg_persist1 (arguments are laid out like the members of a struct); the slab kernel
// compares one reloaded field with the argument itself at entry and fails the launch if the layouts ever disagree.
template <typename T, typename SL>
struct Persist1Kargs { CgArgs<T> a; PersistCtl c; int k_begin, k_end, sv, pend; SL sl; };
template <typename F>
__device__ __forceinline__ F karg(unsigned off) {
  typedef __attribute__((address_space(4))) const char kchar;
  typedef __attribute__((address_space(4))) const unsigned kword;
  kchar* kp = (kchar*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  static_assert(sizeof(F) % 4 == 0, "whole dwords");
  constexpr int NW = (int)(sizeof(F) / 4);
  unsigned w[NW];
  kword* src = (kword*)(kp + off);
#pragma unroll
  for (int i = 0; i < NW; ++i) w[i] = src[i];              // (merged into s_load_dwordx2 / x4 / x8 / x16)
  F out;
  __builtin_memcpy(&out, w, sizeof(F));
  return out;
}

// ---- the three shape parameters an A/B build may still override (-DPISO_PERSIST1_<NAME>=n with scripts/build_variant.py); everything
// else that used to be a compile-time switch here is folded to the variant that shipped (the measurements that decided each one
// are in profiles/README.md, "persistent CG kernel: variants measured and dropped").
// Coefficient rows in flight per wave of the 16-row kernel on a symmetric matrix (kDeep); every other shape: 3 / 4.  With the
// back-and-forth row order: 3 rows 12.5 us, 4 rows 11.8 us, 5 rows 11.5 us per iteration (round 3), 6 rows spill.
#ifndef PISO_PERSIST1_DEPTH
#define PISO_PERSIST1_DEPTH 5
#endif
// Rows of z' = L p that the U pass finds precomputed (kAhead, <= DEPTH).  Round 6 A/B on one box (2048^2, us per iteration):
// DEPTH 5 / AHEAD 2 9.6 - 10.7 (shipped), 4 / 4 10.6 - 10.8, 5 / 3 10.2 - 11.0 (1 VGPR spilled), 5 / 4 11.1 - 11.4 (3), 5 / 5 11.8 - 12.7.
#ifndef PISO_PERSIST1_AHEAD
#define PISO_PERSIST1_AHEAD 2
#endif
// Small regions: U reuses D's z' where z' of all rows of a wave fits this many registers (kKeepZ).  64 would cover the 16-row
// kernel: 26 - 35 VGPRs spill at any DEPTH (round 6), i.e. the registers beside r and p are what z' has no room in.
#ifndef PISO_PERSIST1_KEEP_Z_REGS
#define PISO_PERSIST1_KEEP_Z_REGS 32
#endif
// s_sleep units (64 cycles) of the exchanges' polling (constants, not switches):
constexpr int kPollDelay = 24;                   // between publishing a record and the first polling pass (flat exchange)
constexpr int kPollDelay2 = 40;                  // tree, second level, behind the rows computed ahead (2048^2: 24 -> 9.12, 32 / 40 -> 8.94 us per iteration)
constexpr int kPollDelay2NoAhead = 8;            // ... where nothing is computed ahead (512^2 / 1024 x 256: 24 -> 8: 4.27 -> 4.15 us, 0: 4.22)
constexpr int kPollDelay2Xg = 8;                 // ... of the slab instance's node level (ring of one, 2048^2: 40 -> 8: 10.05 -> 9.67 us)
constexpr int kLocalDelay = 8;                   // XCD-local exchange with one working wave per SIMD
constexpr int kPollSleep = 1;                    // between two polling passes
constexpr int kX1Values = 8;                     // sums per exchange
constexpr int kX1RecWords = 16;                  // 8-byte words per record: 2 per sum {32 payload bits | 32-bit epoch}

// Grid-wide exchange of kX1Values partial sums per workgroup that doubles as the grid barrier (measured 4.4 us for 256 workgroups
// against 11.3 us for "atomic counter + fence + read the partials", scripts/barrier_bench.hip):
//   * every workgroup publishes one record: each double travels as two 8-byte words {32 payload bits | 32-bit epoch}, written and
//     read with relaxed agent-scope atomics (single-copy atomic, coherent across the 8 XCDs' L2s);
//   * the waves poll all records until they carry the current epoch and add them in a fixed order, so every workgroup obtains
//     bitwise the same totals - no counter, no fence, one memory round trip;
//   * records alternate between two arrays (epoch parity): a fast workgroup may publish epoch e+1 while a slow one still reads
//     epoch e, and nobody can reach e+2 before everybody has published e+1.
// DATA written before the exchange (the published perimeter rows) is stored write-through at agent scope (sc1) and drained
// (s_waitcnt vmcnt) by every wave before the workgroup publishes; readers load it at agent scope as well.  Measured: with one 128-byte record per LANE (64 cache
// lines per load instruction) the exchange is bound by the number of fabric transactions (11 us per exchange at 256
// workgroups).  Here the polling is COALESCED and spread over all 8 waves: lane l reads word l % 16 of record 4 i + l / 16, so
// one load instruction covers four whole records (512 contiguous bytes); wave w polls records 32 w .. 32 w + 31 with 8 loads
// per lane, one round trip once the records are there.  Lane pairs (2 q, 2 q + 1) hold the two halves of sum q.
constexpr int kX1Sm = 160;                        // LDS words per parity: [8 sums][8 waves] | [8 waves][8 sums] | 8 flags
struct NoPrefetch { __device__ __forceinline__ void operator()() const {} };

// ---- wave-level reductions of the exchange on as few VALU instructions as possible (the row loops around the exchange are
// bound by VALU issue, and every instruction of a 64-wide wave costs the same ~4.5 SIMD cycles whatever it does).
// 64-bit moves between lanes: DPP inside a row of 16 lanes (VALU, two instructions), the LDS crossbar (ds_bpermute: no VALU
// slot) across rows.
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_update(double old, double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v), o = (unsigned long long)__double_as_longlong(old);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)b, CTRL, 0xf, BANK, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(b >> 32), CTRL, 0xf, BANK, false);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double lanes_xor4(double v) {    // lane l <- lane l ^ 4: row_shl:4 into banks 0, 2 / row_shr:4 into banks 1, 3
  return dpp_update<0x114, 0xa>(dpp_update<0x104, 0x5>(v, v), v);
}
__device__ __forceinline__ double lanes_xor8(double v) { return dpp_move<0x128>(v); }                  // row_ror:8
// v + (v of lane l ^ 16) and v + (v of lane l ^ 32): gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd rows (the upper
// half) of one register with the even rows (the lower half) of another - two swaps of the value with itself leave "mine" and "the
// partner's" in two registers of EVERY lane, no trip through the LDS crossbar (ds_bpermute: ~100 cycles each in a dependent chain
// that every wave of the chip waits for).  Both lanes of a pair add the same two numbers (a + b, b + a: the same bits), as before.
template <int ROWS>
__device__ __forceinline__ double sum_xor_rows(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
  const auto r0 = ROWS == 16 ? __builtin_amdgcn_permlane16_swap(lo, lo, false, false) : __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto r1 = ROWS == 16 ? __builtin_amdgcn_permlane16_swap(hi, hi, false, false) : __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const double x = __longlong_as_double((long long)(((unsigned long long)r1[0] << 32) | r0[0]));
  const double y = __longlong_as_double((long long)(((unsigned long long)r1[1] << 32) | r0[1]));
  return x + y;
}
__device__ __forceinline__ double sum_xor16(double v) { return sum_xor_rows<16>(v); }
__device__ __forceinline__ double sum_xor32(double v) { return sum_xor_rows<32>(v); }
// Eight per-lane partial sums -> lane l holds the WAVE total of value l & 7.  Reduce-scatter butterfly over lane bits 0, 1, 2 (a
// lane keeps half of its values and receives the partner's contribution to them: 7 + 7 + ... instructions instead of three full
// butterflies of eight values), then plain butterflies of the ONE remaining value over bits 3 (DPP), 4 and 5 (LDS crossbar).
// ~56 VALU instructions; eight wave_sum_uniform calls are ~190.  Every step adds a lane's value and its partner's: both lanes of
// a pair compute a + b and b + a - the same bits.
__device__ __forceinline__ double wave_reduce_scatter8(const double (&v)[8]) {
  const int lane = threadIdx.x & 63;
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0;
  double a[4], b[2];
#pragma unroll
  for (int j = 0; j < 4; ++j)                                // a[j]: value 2 j + b0, summed over lane pairs
    a[j] = (b0 ? v[2 * j + 1] : v[2 * j]) + dpp_move<0xB1>(b0 ? v[2 * j] : v[2 * j + 1]);      // quad_perm [1,0,3,2]
#pragma unroll
  for (int m = 0; m < 2; ++m)                                // b[m]: value 4 m + 2 b1 + b0, summed over quads
    b[m] = (b1 ? a[2 * m + 1] : a[2 * m]) + dpp_move<0x4E>(b1 ? a[2 * m] : a[2 * m + 1]);      // quad_perm [2,3,0,1]
  double c = (b2 ? b[1] : b[0]) + lanes_xor4(b2 ? b[0] : b[1]);                                 // value l & 7, summed over 8 lanes
  c += lanes_xor8(c);
  c = sum_xor16(c);
  c = sum_xor32(c);
  return c;
}
// slot / nslots: this workgroup's record and the number of records in play (blockIdx.x / gridDim.x, or the rank / size of the
// XCD-local group).  LOCAL: every participant runs on the same XCD - records are stored without sc1 (they stay in that XCD's L2)
// and polled with sc1 loads (L1 bypassed, L2-served): an L2 round trip instead of two trips through the fabric.
template <typename T, bool LOCAL = false, typename F = NoPrefetch>
__device__ __forceinline__ bool grid_exchange8(const PersistCtl& c, T (&v)[kX1Values], unsigned epoch, T* smem, int slot, int nslots,
                                               F after_drain = F(), unsigned long long* tsub = nullptr) {
  // tsub (diagnostic builds): clocks of [0] reduction + drain of this wave's stores, [1] first barrier, [2] publish + polling,
  // [3] wave sums + second barrier
  unsigned long long t0 = (kPersistDiag && tsub) ? wall_clock64() : 0;
  auto tsplit = [&](int q) __attribute__((always_inline)) {
    if (kPersistDiag && tsub) { const unsigned long long t = wall_clock64(); tsub[q] += t - t0; t0 = t; }
  };
  typedef unsigned long long u64;
  constexpr int NV = kX1Values;
  static_assert(kPersistMaxGrid == kPersistWaves * 32 && (kPersistWaves & (kPersistWaves - 1)) == 0, "every wave polls 32 records");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  T* sm = smem + (epoch & 1) * kX1Sm;                       // parity double buffer: two __syncthreads per exchange
  {
    double vd[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) vd[q] = (double)v[q];
    const double mine = wave_reduce_scatter8(vd);              // lane l: value l & 7, summed over this wave
    if (lane < NV) sm[lane * kPersistWaves + wave] = (T)mine;
  }
  // EVERY vector-memory operation of this wave has completed - in particular its write-through perimeter stores - before the
  // workgroup's record says so.  (A counted wait that lets the two prefetch loads issued behind the last store stay in flight
  // would save ~0.4 us; it relies on loads and stores retiring in one order, which is not promised.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  after_drain();
  tsplit(0);
  __syncthreads();                                            // every wave of the workgroup has drained its stores
  tsplit(1);
  u64* rec = c.rec + (size_t)(epoch & 1) * kPersistMaxGrid * kX1RecWords;
  if (wave == 0) {
    // lane l < 16 publishes word l: sum l / 2, low half (even l) or high half (odd l) - one store instruction, one cache line
    const int vq = (lane >> 1) & (NV - 1);
    T s = 0;
    for (int w = 0; w < kPersistWaves; ++w) s += sm[vq * kPersistWaves + w];
    const u64 bits = (u64)__double_as_longlong((double)s);
    const u64 word = (lane & 1) ? ((bits & 0xffffffff00000000ull) | epoch) : (((bits & 0xffffffffull) << 32) | epoch);
    if (lane < kX1RecWords) {
      if constexpr (LOCAL) __hip_atomic_store(rec + (size_t)slot * kX1RecWords + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_store(rec + (size_t)slot * kX1RecWords + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // (Tried: no barrier here - every wave counts itself in through LDS and the LAST one publishes.  Wave 0 waits 2.3 us of a 7 us
  // exchange in this barrier, but without it the exchange took 8.3 us and the row loops slowed down - 21 us per iteration
  // against 17: the waves that arrive early poll, and their polling competes with the stores of the ones still working.)
  bool good = true;
  {
    const int wd = lane & 15, sub = lane >> 4;               // my word of the record, my record inside a group of four
    u64 w[8];
    bool okl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      w[i] = 0;
      okl[i] = ((wave * 8 + i) * 4 + sub) >= nslots;           // records beyond the grid count as arrived (payload 0)
    }
    unsigned spins = 0;
    // a first poll that finds every record beats two passes: the records of 256 workgroups that finish their row loops together
    // need ~0.6 us to become visible; measured at 2048^2: no delay 11.7, s_sleep 16 .. 32 11.4, 48 11.7, 64 11.9 us per iteration
    if constexpr (!LOCAL) { if (kPollDelay > 0) __builtin_amdgcn_s_sleep(kPollDelay); }
    while (true) {
      bool ok = true;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (!okl[i]) w[i] = __hip_atomic_load(rec + (size_t)((wave * 8 + i) * 4 + sub) * kX1RecWords + wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (!okl[i]) okl[i] = (unsigned)(w[i] & 0xffffffffull) == epoch;
        ok = ok && okl[i];
      }
      if (__all(ok)) break;
      if (++spins > (1u << 22)) { good = false; break; }
      __builtin_amdgcn_s_sleep(kPollSleep);
    }
    tsplit(2);
    // even lanes assemble their sum from their own word (low half) and the neighbour lane's (high half); records of a lane
    // are added in order, then the four records-per-instruction rows, then (after the barrier) the eight waves
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned hi_other = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(w[i] >> 32), 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
      const u64 bits = (w[i] >> 32) | ((u64)hi_other << 32);
      acc += __longlong_as_double((long long)bits);          // (odd lanes add garbage that nobody reads; absent records add 0)
    }
    acc = sum_xor16(acc);                                 // the four records-per-instruction groups of lanes (LDS crossbar)
    acc = sum_xor32(acc);
    if (lane < 2 * NV && !(lane & 1)) sm[64 + wave * NV + (lane >> 1)] = (T)acc;   // this wave's 32 records, value lane / 2
    if (lane == 0) {
      sm[128 + wave] = good ? (T)0 : (T)1;
      if (!good) *c.err = 1;
    }
  }
  __syncthreads();
  tsplit(3);
  {
    // one read fetches all 8 x 8 wave sums (lane l: wave l / 8, value l % 8); butterflies over the wave index leave every lane
    // with the total of value l % 8 - the same bits in every wave of every workgroup (same inputs, same order)
    double t = (double)sm[64 + lane];
    t += lanes_xor8(t);
    t = sum_xor16(t);
    t = sum_xor32(t);
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = (T)read_lane_c(t, q);
    const T bad = sm[128 + (lane & (kPersistWaves - 1))];
    good = !__any(bad != (T)0);
  }
  return good;
}

// ---- The same exchange as a TREE that follows the hardware (chip-wide launches; round 4): workgroup -> XCD leader -> everybody.
//   level 1  every workgroup stores its record WITHOUT sc1 into the records of ITS XCD (slot = 32 xcd + arrival rank on that XCD;
//            the store stays in that XCD's L2); wave 0 of the XCD's leader (arrival rank 0) polls the XCD's records with sc1 loads
//            (L1 bypassed, served by the L2 both share) and adds them in rank order;
//   level 2  the leader publishes the XCD's sums as one record through the fabric (sc1 store); EVERY wave of every workgroup polls
//            the eight XCD records itself (two coalesced loads per lane) and adds them in XCD order - bitwise the same totals in
//            every wave of the chip, no second barrier, no LDS round trip behind the polling.
// Measured (scripts/barrier_bench.hip, 256 workgroups, 3 sums): 2.40 us per exchange against 3.7-5.3 us for the flat all-to-all
// variants (every workgroup polling 256 records through the fabric: 8 MB of polling reads per pass; here 4 KB per XCD at level 1
// and 2 MB at level 2).  hx packs what a workgroup learnt at entry (hier_enter): bits 0-2 XCD, 3-8 arrival rank, 9-14 workgroups
// on my XCD, 15-22 XCDs that hold workgroups.  A wave whose polling gives up sets the workgroup's LDS flag and the launch's
// error word; all waves of the workgroup read the flag behind the next barrier and leave the loop together.
__device__ __forceinline__ unsigned long long* hier_level2(const PersistCtl& c) {
  return c.rec + (size_t)2 * kPersistMaxGrid * kX1RecWords;                 // right behind the level-1 records (kPersistRecWords)
}
// entry of a chip-wide launch: which XCD am I on, how many workgroups does every XCD hold, and which of them am I?  One returning
// atomic per workgroup, then everybody waits for everybody ONCE per launch (c.xcd[0..7] arrivals per XCD, [9] arrivals in all;
// zeroed by the host before the launch).  My place among my XCD's workgroups is my place by WORKGROUP INDEX, not by arrival: the
// leader adds the records in that order, so two launches that the hardware deals to the XCDs the same way add in the same order
// and a solve is reproducible bit for bit from run to run (by arrival order the forward solves of the 2048^2 benchmark took
// 325 - 360 iterations on the same input, now and then 1 005).  Every workgroup leaves its XCD in a table before it counts itself in.
__device__ __forceinline__ unsigned hier_enter(const PersistCtl& c, int* lds2) {      // lds2: [0] hx, [1] launch cannot run, [2] sticky flag of the exchanges
  int* const table = c.xcd + kPersistXcdTable;
  if (threadIdx.x == 0) {
    const int xcc = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7);       // HW_REG_XCC_ID[3:0]
    __hip_atomic_store(table + blockIdx.x, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int arrival = __hip_atomic_fetch_add(c.xcd + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(c.xcd + 9, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);     // (release: my table entry is out before I count)
    unsigned spins = 0;
    bool ok = arrival < 32;
    while (ok && __hip_atomic_load(c.xcd + 9, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.x) {
      if (++spins > (1u << 22)) { ok = false; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    unsigned present = 0, mine = 0;
    for (int x = 0; x < kXcds; ++x) {
      const int n = __hip_atomic_load(c.xcd + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (n > 0) present |= 1u << x;
      if (n > 32) ok = false;
      if (x == xcc) mine = (unsigned)n;
    }
    if (!ok) *c.err = 1;
    lds2[0] = (int)((unsigned)xcc | ((mine & 63u) << 9) | (present << 15));
    lds2[1] = ok ? 0 : 1;
    lds2[2] = 0;
  }
  __syncthreads();
  if (threadIdx.x < 64 && !lds2[1]) {                        // wave 0: the workgroups of my XCD with a smaller index
    const int xcc = lds2[0] & 7;
    int before = 0;
    for (int b = (int)threadIdx.x; b < (int)gridDim.x; b += 64)
      before += (b < (int)blockIdx.x && __hip_atomic_load(table + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == xcc) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    if (threadIdx.x == 0) lds2[0] |= (before & 63) << 3;
  }
  __syncthreads();
  return (unsigned)__builtin_amdgcn_readfirstlane(lds2[0]);
}
// XG (slab instance, round 5): the node's level of the exchange rides on the tree's second level instead of following it.  The XCD
// leaders store their XCD's record straight into EVERY rank's mailbox (system-scope stores over xGMI; the own mailbox included);
// wave w of every workgroup polls the eight XCD records of RANK w in its own mailbox and adds them in XCD order, the rank totals
// meet in LDS behind one barrier and every wave adds them by the same butterfly over the rank index - bitwise the same totals in
// every wave of every GPU.  (Round 4 had a serial level here: workgroup 0 waited for the chip's totals, wrote them to the peers,
// and wave 0 of every workgroup polled again - one more uncached round trip per iteration.)  All eight records of a rank also
// certify that every row this rank stored into a peer's mailbox has completed: its workgroups drained their stores before they
// published, and a leader publishes only after it has seen all workgroups of its XCD.  An XCD that holds no workgroups (small
// grids) is published with zero sums by the leader of the rank's lowest XCD in play.  sl_off: where the SlabCtl sits in the
// kernarg segment - mailbox addresses, rank and world are fetched where they are used (karg), not held in SGPRs across the loops.
constexpr int kX1SmX = 80;                        // LDS words of the node level per parity: [8 ranks][8 sums], 8 flags
template <typename T, int DELAY2, bool XG = false, typename F = NoPrefetch>
__device__ __forceinline__ bool grid_exchange8_hier(const PersistCtl& c, T (&v)[kX1Values], unsigned epoch, T* smem, unsigned hx, int* flag,
                                                    F while_records_travel = F(), unsigned long long* tsub = nullptr, unsigned sl_off = 0,
                                                    T* smx2 = nullptr) {
  unsigned long long t0 = (kPersistDiag && tsub) ? wall_clock64() : 0;
  auto tsplit = [&](int q) __attribute__((always_inline)) {
    if (kPersistDiag && tsub) { const unsigned long long t = wall_clock64(); tsub[q] += t - t0; t0 = t; }
  };
  typedef unsigned long long u64;
  constexpr int NV = kX1Values;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  T* sm = smem + (epoch & 1) * kX1Sm;                       // parity double buffer (one barrier per exchange separates writers and readers)
  {
    double vd[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) vd[q] = (double)v[q];
    const double mine = wave_reduce_scatter8(vd);              // lane l: value l & 7, summed over this wave
    if (lane < NV) sm[lane * kPersistWaves + wave] = (T)mine;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every perimeter store of this wave has completed (see grid_exchange8)
  tsplit(0);
  __syncthreads();
  tsplit(1);
  // A polling pass of an EARLIER exchange gave up somewhere in this workgroup: the flag is sticky, every wave reads it here behind the
  // barrier and all of them leave together at the end of this exchange - the wave that gave up included: it returned "healthy" like
  // its siblings.  (No return from here: an exit in the middle of the iteration loop turns its control flow into exec-mask flow and
  // the iteration counter into a vector register.  The polling loops below give up at once instead: spin0.)
  const bool good = __builtin_amdgcn_readfirstlane(*flag) == 0;
  const unsigned spin0 = good ? 0u : (1u << 30);
  const int xcc = (int)(hx & 7u), rank = (int)((hx >> 3) & 63u), nmine = (int)((hx >> 9) & 63u);
  const unsigned present = (hx >> 15) & 0xffu;
  u64* rec1 = c.rec + (size_t)(epoch & 1) * kPersistMaxGrid * kX1RecWords + (size_t)xcc * 32 * kX1RecWords;
  u64* rec2 = hier_level2(c) + (size_t)(epoch & 1) * kXcds * kX1RecWords;
  // lane l polls word l % 16 of record 4 i + l / 16, i.e. 8-byte word 64 i + l of the record array: ONE per-lane offset, made opaque
  // so that nothing derived from it is hoisted out of the iteration loop into vector registers that live across the row loops
  int lw = lane;
  asm volatile("" : "+v"(lw));
  bool mygood = true;
  if (wave == 0) {
    {
      // lane l < 16 publishes word l: sum l / 2, low half (even l) or high half (odd l) - one store instruction, one cache line
      const int vq = (lane >> 1) & (NV - 1);
      T s = 0;
      for (int w = 0; w < kPersistWaves; ++w) s += sm[vq * kPersistWaves + w];
      const u64 bits = (u64)__double_as_longlong((double)s);
      const u64 word = (lane & 1) ? ((bits & 0xffffffff00000000ull) | epoch) : (((bits & 0xffffffffull) << 32) | epoch);
      if (lane < kX1RecWords) __hip_atomic_store(rec1 + (size_t)rank * kX1RecWords + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (rank == 0) {                                          // the XCD's leader: the records of my XCD (through its L2), in rank order
      // (branch-free passes: all eight loads every time, records beyond the XCD's count masked by one compare against a scalar -
      // per-record arrival flags are eight lane masks = sixteen SGPRs the row loops then spill)
      u64 w[8];
      // (opaque: left visible, the eight bounds lim - 64 i are constants of the launch that live in SGPRs across the row loops - spilled,
      // and a spilled SGPR comes back through v_readlane; recomputed here they are eight scalar subtractions per exchange)
      int lim = nmine * kX1RecWords;                          // words of the XCD's block that belong to records in play
      asm volatile("" : "+s"(lim));
      unsigned spins = spin0;
      while (true) {
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = __hip_atomic_load(rec1 + lw + i * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned bad = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) bad |= (lw < lim - i * 64) ? ((unsigned)(w[i] & 0xffffffffull) ^ epoch) : 0u;
        if (__all(bad == 0)) break;
        if (++spins > (1u << 22)) { mygood = false; break; }
      }
      double acc = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned hi_other = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(w[i] >> 32), 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
        const double val = __longlong_as_double((long long)((w[i] >> 32) | ((u64)hi_other << 32)));   // (odd lanes: garbage that nobody reads)
        acc += (lw < lim - i * 64) ? val : 0.0;
      }
      acc = sum_xor16(acc);
      acc = sum_xor32(acc);                                // even lane 2 q (of every group of 16): the XCD's sum of value q
      const double other = dpp_move<0xB1>(acc);              // odd lanes: the even neighbour's sum
      const u64 bits = (u64)__double_as_longlong((lane & 1) ? other : acc);
      const u64 word = (lane & 1) ? ((bits & 0xffffffff00000000ull) | epoch) : (((bits & 0xffffffffull) << 32) | epoch);
      if constexpr (!XG) {
        if (lane < kX1RecWords) __hip_atomic_store(rec2 + (size_t)xcc * kX1RecWords + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        constexpr unsigned pvo = (unsigned)offsetof(SlabCtl, pv);
        const int world = karg<int>(sl_off + pvo + (unsigned)offsetof(PeerView, world));
        const int myrank = karg<int>(sl_off + pvo + (unsigned)offsetof(PeerView, rank));
        // (measurements only - SlabCtl::hop_ticks > 0: the records leave this many 10 ns ticks late, as if the link had that latency)
        const unsigned hop = karg<unsigned>(sl_off + (unsigned)offsetof(SlabCtl, hop_ticks));
        if (hop) { const unsigned long long t_go = wall_clock64() + hop; while (wall_clock64() < t_go) __builtin_amdgcn_s_sleep(1); }
        const bool lowest = (present & ((1u << xcc) - 1u)) == 0;     // (scalar) the leader that also speaks for the XCDs without workgroups
        for (int p = 0; p < world; ++p) {
          char* mb = karg<char*>(sl_off + pvo + (unsigned)offsetof(PeerView, mbox) + 8u * (unsigned)p);
          if (lane < kX1RecWords) peer_store(reinterpret_cast<peer_u64*>(mb + PeerLayout::xcd_rec(epoch & 1, myrank, xcc)) + lane, word);
          if (lowest && present != 0xffu) {
            for (int x = 0; x < kXcds; ++x)
              if (!((present >> x) & 1u) && lane < kX1RecWords)
                peer_store(reinterpret_cast<peer_u64*>(mb + PeerLayout::xcd_rec(epoch & 1, myrank, x)) + lane, (peer_u64)epoch);
          }
        }
      }
    }
  }
  // the records need a microsecond or two to make their way: work that does not depend on the sums goes here (the row loops
  // are bound by VALU issue, and the SIMDs idle while the exchange is in flight)
  while_records_travel();
  if constexpr (!XG) {
    // every wave: the eight XCD records (lane l: word l % 16 of record 4 i + l / 16), added in XCD order
    u64 w[2];
    unsigned spins = spin0;
    if (DELAY2 > 0) __builtin_amdgcn_s_sleep(DELAY2);
    while (true) {
#pragma unroll
      for (int i = 0; i < 2; ++i) w[i] = __hip_atomic_load(rec2 + lw + i * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned bad = 0;
#pragma unroll
      for (int i = 0; i < 2; ++i)                            // (XCDs without workgroups: nothing to wait for, payload 0)
        bad |= (((present >> (i * 4)) >> (lw >> 4)) & 1u) ? ((unsigned)(w[i] & 0xffffffffull) ^ epoch) : 0u;
      if (__all(bad == 0)) break;
      if (++spins > (1u << 22)) { mygood = false; break; }
      __builtin_amdgcn_s_sleep(kPollSleep);
    }
    tsplit(2);
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned hi_other = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(w[i] >> 32), 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
      const double val = __longlong_as_double((long long)((w[i] >> 32) | ((u64)hi_other << 32)));
      acc += (((present >> (i * 4)) >> (lw >> 4)) & 1u) ? val : 0.0;
    }
    acc = sum_xor16(acc);
    acc = sum_xor32(acc);                                  // lane 2 q: the total of value q - the same bits in every wave of the chip
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = (T)read_lane_c(acc, 2 * q);
    if (!mygood) {                                              // (wave-uniform)
      if (lane == 0) { *flag = 1; *c.err = 1; }
    }
    tsplit(3);
    return good;
  } else {
    // wave w: the eight XCD records of rank w in MY mailbox (every XCD slot of a rank in play is published, see above)
    T* smx = smx2 + (epoch & 1) * kX1SmX;
    constexpr unsigned pvo = (unsigned)offsetof(SlabCtl, pv);
    const int world = karg<int>(sl_off + pvo + (unsigned)offsetof(PeerView, world));
    double acc = 0;
    if (wave < world) {
      const char* own = karg<char*>(sl_off + (unsigned)offsetof(SlabCtl, rows_own)) - PeerLayout::kRows;
      const peer_u64* recs = reinterpret_cast<const peer_u64*>(own + PeerLayout::xcd_rec(epoch & 1, wave, 0));
      peer_u64 w[2];
      unsigned spins = spin0;
      if (DELAY2 > 0) __builtin_amdgcn_s_sleep(DELAY2);
      while (true) {
#pragma unroll
        for (int i = 0; i < 2; ++i) w[i] = peer_load(recs + lw + i * 64);
        unsigned bad = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) bad |= (unsigned)(w[i] & 0xffffffffull) ^ epoch;
        if (__all(bad == 0)) break;
        if (++spins > kPeerSpinLimit) { mygood = false; break; }      // (kPeerSpinLimit < spin0)
        __builtin_amdgcn_s_sleep(kPollSleep);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned hi_other = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(w[i] >> 32), 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
        acc += __longlong_as_double((long long)((w[i] >> 32) | ((peer_u64)hi_other << 32)));
      }
      acc = sum_xor16(acc);
      acc = sum_xor32(acc);                                // lane 2 q: rank w's total of value q (XCD order)
    }
    tsplit(2);
    if (lane < 2 * NV && !(lane & 1)) smx[wave * NV + (lane >> 1)] = (T)acc;       // (a wave without a rank: zeros)
    if (!mygood) {
      if (lane == 0) { *flag = 1; *c.err = 1; }
    }
    __syncthreads();
    {
      // one read fetches the 8 x 8 rank totals (lane l: rank l / 8, value l % 8); the butterfly over the rank index leaves every
      // lane with the node's total of value l % 8 - the same order of additions in every wave of every GPU
      double t = (double)smx[lane];
      t += lanes_xor8(t);
      t = sum_xor16(t);
      t = sum_xor32(t);
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q] = (T)read_lane_c(t, q);
    }
    tsplit(3);
    return good && __builtin_amdgcn_readfirstlane(*flag) == 0;
  }
}

// ---- XCD-local launches (LOCAL, at most 32 workgroups, all on one XCD): the same idea in one level.  Wave 0 publishes the workgroup's
// record (plain store: it stays in the XCD's L2), then EVERY wave polls the group's records itself (sc1 loads: L1 bypassed, served
// by that L2; eight coalesced loads per lane cover 32 records) and adds them in slot order - no second barrier, no LDS round trip
// behind the polling (0.36 us of a 3.5 us iteration at 256^2).  Error handling as in grid_exchange8_hier (sticky LDS flag).
template <typename T>
__device__ __forceinline__ bool grid_exchange8_local(const PersistCtl& c, T (&v)[kX1Values], unsigned epoch, T* smem, int slot, int nslots, int* flag,
                                                     unsigned long long* tsub = nullptr) {
  unsigned long long t0 = (kPersistDiag && tsub) ? wall_clock64() : 0;
  auto tsplit = [&](int q) __attribute__((always_inline)) {
    if (kPersistDiag && tsub) { const unsigned long long t = wall_clock64(); tsub[q] += t - t0; t0 = t; }
  };
  typedef unsigned long long u64;
  constexpr int NV = kX1Values;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  T* sm = smem + (epoch & 1) * kX1Sm;
  {
    double vd[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) vd[q] = (double)v[q];
    const double mine = wave_reduce_scatter8(vd);
    if (lane < NV) sm[lane * kPersistWaves + wave] = (T)mine;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  tsplit(0);
  __syncthreads();
  tsplit(1);
  const bool good = __builtin_amdgcn_readfirstlane(*flag) == 0;      // (sticky, read behind the barrier: all waves leave together, see grid_exchange8_hier)
  const unsigned spin0 = good ? 0u : (1u << 30);
  u64* rec = c.rec + (size_t)(epoch & 1) * kPersistMaxGrid * kX1RecWords;
  int lw = lane;
  asm volatile("" : "+v"(lw));
  bool mygood = true;
  if (wave == 0) {
    const int vq = (lane >> 1) & (NV - 1);
    T s = 0;
    for (int w = 0; w < kPersistWaves; ++w) s += sm[vq * kPersistWaves + w];
    const u64 bits = (u64)__double_as_longlong((double)s);
    const u64 word = (lane & 1) ? ((bits & 0xffffffff00000000ull) | epoch) : (((bits & 0xffffffffull) << 32) | epoch);
    if (lane < kX1RecWords) __hip_atomic_store(rec + (size_t)slot * kX1RecWords + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  {
    u64 w[8];
    const int lim = nslots * kX1RecWords;
    unsigned spins = spin0;
    // (one working wave per SIMD - c.waves = 4: the record needs ~0.2 us to arrive and a first pass that misses it queues in front
    // of the one that would find it: 256^2 3.33 -> 3.15 us per iteration with 8 units, 4: 3.21, 12: 3.23; with two working waves per
    // SIMD - 512 x 256 - any delay loses: 3.84 / 3.83 / 3.92 / 4.00 / 4.10 with 0 / 4 / 8 / 12 / 16)
    if (kLocalDelay > 0 && c.waves < kPersistWaves) __builtin_amdgcn_s_sleep(kLocalDelay);
    while (true) {
#pragma unroll
      for (int i = 0; i < 8; ++i) w[i] = __hip_atomic_load(rec + lw + i * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned bad = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) bad |= (lw < lim - i * 64) ? ((unsigned)(w[i] & 0xffffffffull) ^ epoch) : 0u;
      if (__all(bad == 0)) break;
      if (++spins > (1u << 22)) { mygood = false; break; }
      __builtin_amdgcn_s_sleep(kPollSleep);
    }
    tsplit(2);
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned hi_other = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(w[i] >> 32), 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
      const double val = __longlong_as_double((long long)((w[i] >> 32) | ((u64)hi_other << 32)));
      acc += (lw < lim - i * 64) ? val : 0.0;
    }
    acc = sum_xor16(acc);
    acc = sum_xor32(acc);
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = (T)read_lane_c(acc, 2 * q);
  }
  if (!mygood) {
    if (lane == 0) { *flag = 1; *c.err = 1; }
  }
  tsplit(3);
  return good;
}

// x-neighbours across the lanes of a wave.  The values beyond the two ends of the strip (`ring`: the left neighbour of row j in
// lane j, the right neighbour in lane 48 + j) enter through the `old` operand of the wave shift: one row-local DPP shift brings
// lane j to lane 0 (row_shl:j) or lane 48 + j to lane 63 (row_shr:15-j), the wave shift keeps it there.  2 VALU instructions
// per 32-bit half - through v_readlane + v_mov (SGPR broadcast) it was 3, and the loop is bound by VALU issue.
template <int CTRL>
__device__ __forceinline__ int dpp_mov32(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }   // (no `old` operand to set up)
template <bool UP>
__device__ __forceinline__ int ring_to_end32(int ring, int j) {     // j: constant after unrolling
  if (UP) {
    switch (j) {
      case 0: return ring;
      case 1: return dpp_mov32<0x101>(ring); case 2: return dpp_mov32<0x102>(ring); case 3: return dpp_mov32<0x103>(ring);
      case 4: return dpp_mov32<0x104>(ring); case 5: return dpp_mov32<0x105>(ring); case 6: return dpp_mov32<0x106>(ring);
      case 7: return dpp_mov32<0x107>(ring); case 8: return dpp_mov32<0x108>(ring); case 9: return dpp_mov32<0x109>(ring);
      case 10: return dpp_mov32<0x10a>(ring); case 11: return dpp_mov32<0x10b>(ring); case 12: return dpp_mov32<0x10c>(ring);
      case 13: return dpp_mov32<0x10d>(ring); case 14: return dpp_mov32<0x10e>(ring); default: return dpp_mov32<0x10f>(ring);
    }
  } else {
    switch (j) {
      case 15: return ring;
      case 14: return dpp_mov32<0x111>(ring); case 13: return dpp_mov32<0x112>(ring); case 12: return dpp_mov32<0x113>(ring);
      case 11: return dpp_mov32<0x114>(ring); case 10: return dpp_mov32<0x115>(ring); case 9: return dpp_mov32<0x116>(ring);
      case 8: return dpp_mov32<0x117>(ring); case 7: return dpp_mov32<0x118>(ring); case 6: return dpp_mov32<0x119>(ring);
      case 5: return dpp_mov32<0x11a>(ring); case 4: return dpp_mov32<0x11b>(ring); case 3: return dpp_mov32<0x11c>(ring);
      case 2: return dpp_mov32<0x11d>(ring); case 1: return dpp_mov32<0x11e>(ring); default: return dpp_mov32<0x11f>(ring);
    }
  }
}
// lane l receives `v` of lane l - 1 (UP) or l + 1 (!UP); lane 0 / lane 63 receives the ring value of row j
template <bool UP, typename S>
__device__ __forceinline__ S shift_ring(S v, S ring, int j) {
  constexpr int ctrl = UP ? 0x138 /* wave_shr:1 */ : 0x130 /* wave_shl:1 */;
  if constexpr (sizeof(S) == 8) {
    const unsigned long long b = (unsigned long long)__double_as_longlong((double)v), e = (unsigned long long)__double_as_longlong((double)ring);
    const int elo = ring_to_end32<UP>((int)(unsigned)e, j), ehi = ring_to_end32<UP>((int)(unsigned)(e >> 32), j);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(elo, (int)(unsigned)b, ctrl, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(ehi, (int)(unsigned)(b >> 32), ctrl, 0xf, 0xf, false);
    return (S)__longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  } else {
    const int e = ring_to_end32<UP>(__float_as_int((float)ring), j);
    return (S)__int_as_float(__builtin_amdgcn_update_dpp(e, __float_as_int((float)v), ctrl, 0xf, 0xf, false));
  }
}
// lane layout of the ring columns: lanes [0, R) the left neighbours of rows 0 .. R-1, lanes [48, 48 + R) the right neighbours
__device__ __forceinline__ void ring_lane(int lane, int R, int& side, int& er) {
  side = (lane < R) ? 0 : ((lane >= 48 && lane < 48 + R) ? 1 : 2);
  er = (side == 1) ? lane - 48 : lane;
}

constexpr int kSystem = 17;                       // buffer cache policy sc0 | sc1: system scope (peer-mapped mailboxes)

// RAGGED = true: padded-grid mode (CgArgs::nx_true / ny_true): the rank-1 shift skips the cells of the padding - everything else
// about them is zero by construction (zero coefficients, zero right-hand side).
// LOCAL = true: XCD-local mode for grids that need at most one XCD's worth of workgroups (the small BASELINE configurations,
// where an iteration is nothing but the exchange: 4.3 us with 16 workgroups spread over the chip).  The launch has 8 x c.local_n
// workgroups; every one counts itself in on its XCD (HW_REG_XCC_ID), the workgroup that completes the first quota of c.local_n
// names its XCD the winner, the c.local_n first arrivals there run the solve with their arrival ranks as workgroup numbers and
// everybody else exits.  Some XCD always collects a quota (8 x local_n workgroups over 8 XCDs), whatever the dispatcher does:
// placement decides nothing but speed.  Inside the group, published data and records are stored WITHOUT sc1 (they stay in the
// XCD's L2) and read with sc1 loads (L1 bypassed, L2-served).
template <typename T, typename CT, int R, int NQ, bool RECON, bool SYM, bool SLAB = false, bool RAGGED = false, bool LOCAL = false>
__global__ __launch_bounds__(kPersistThreads) void cg_persist1(CgArgs<T> a, PersistCtl c, int k_begin, int k_end, int sv, int pend,
                                                               std::conditional_t<SLAB, SlabCtl, NoSlab> sl = {}) {
  static_assert(!(SLAB && RAGGED), "a slab is never padded");
  static_assert(!(SLAB && LOCAL), "a slab's neighbours are other GPUs");
  // cache policy of what other workgroups read inside the launch (non-temporal hints on the published rows / the halo loads were
  // measured and lose: the publish -> read path lives on the caches)
  constexpr int kPub = LOCAL ? kPlain : kAgent;
  constexpr int kHalo = kAgent;
  constexpr int V = 16 / sizeof(T);                        // 16-byte lane accesses
  static_assert(R * NQ <= 16 && 2 * R <= 64, "at most 16 rows per wave; the edge columns of a region fit one wave-wide load");
  static_assert(!SLAB || sizeof(T) == 8, "mailbox rows hold 8-byte elements");
  __shared__ T xs[kPersistWaves * NQ * R * 64 * V];        // the solution of my regions (128 KB at 16 rows per wave, fp64)
  __shared__ T smem[2 * kX1Sm + (SLAB ? 2 * kX1SmX : 0)];
  // the direction on the rows below / above my regions: parked in LDS (two reads per pass, one read-modify-write per iteration)
  constexpr bool kParkHalos = (NQ * R < 16) || NQ == 1;     // (two regions of 8 rows: x already fills the LDS)
  __shared__ T halo_s[kParkHalos ? kPersistWaves * NQ * 2 * 64 * V : 1];
  // The ring columns of the direction (and, SYM, the W coefficient of the column right of the strip), by lane as `edge` / `eW`
  // hold them: lanes [0, R) the left neighbours of rows 0 .. R-1, lanes [48, 48 + R) the right ones.  Row jj needs its two ring
  // values in lanes 0 and 63 as the `old` operand of the wave shifts; through the DPP network that is a row-local shift per 32-bit
  // half and side (5 VALU instructions per row and pass in a loop that is bound by VALU issue), through the LDS it is one
  // broadcast read per row and pass that every lane receives (ring_issue) and no VALU slot at all.
  // (the slab variant and the fp64-coefficient fallback have no registers for the values in flight: they keep the DPP shifts)
  constexpr bool kRingLds = sizeof(CT) == 4;
  // kParkRing: the ring-column copies of p AND r (one value per lane) live in the wave's LDS ring block between their two uses per
  // iteration (D's start, U's end) instead of in registers across both row loops
  constexpr bool kParkRing = kRingLds;
  constexpr int kRingBytes = 64 * (int)sizeof(T) + 64 * (int)sizeof(CT) + (kParkRing ? 64 * (int)sizeof(T) : 0);
  constexpr unsigned kRingR = 64u * (unsigned)sizeof(T) + 64u * (unsigned)sizeof(CT);      // byte offset of the parked r column inside a ring block
  __shared__ __attribute__((aligned(16))) unsigned char ring_s[kRingLds ? kPersistWaves * NQ * kRingBytes : 16];
  const int nx = a.nx, ny = a.ny;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: scalar branches, SGPRs
  int wg = blockIdx.x;                                     // XCD-contiguous bands (block b is observed on XCD b % 8)
  if (gridDim.x % kXcds == 0) wg = (blockIdx.x % kXcds) * (gridDim.x / kXcds) + blockIdx.x / kXcds;
  int nslots = (int)gridDim.x;                             // workgroups that take part in the exchanges
  if constexpr (LOCAL) {
    __shared__ int local_rank_s;
    if (threadIdx.x == 0) {
      const int xcc = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7);       // HW_REG_XCC_ID[3:0]
      const int arrival = __hip_atomic_fetch_add(c.xcd + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int rank = -1;
      if (arrival < c.local_n) {
        if (arrival == c.local_n - 1) {                    // my XCD's quota is complete: the first such XCD wins
          int none = 0;
          __hip_atomic_compare_exchange_strong(c.xcd + 8, &none, xcc + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int winner = 0;
        unsigned spins = 0;
        while ((winner = __hip_atomic_load(c.xcd + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
          if (++spins > (1u << 22)) break;                 // (cannot happen: some XCD completes a quota; never hang all the same)
          __builtin_amdgcn_s_sleep(2);
        }
        if (winner == xcc + 1) rank = arrival;
        else if (winner == 0) { *c.err = 1; }
      }
      local_rank_s = rank;
    }
    __syncthreads();
    wg = __builtin_amdgcn_readfirstlane(local_rank_s);     // (wave-uniform BY CONSTRUCTION: an LDS read is a vector value to the compiler - every offset derived
                                                           //  from it became per-lane arithmetic, and the halo loads' resources were built in waterfall loops)
    if (wg < 0) return;                                    // not in the group that runs the solve
    nslots = c.local_n;
  }
  const int slot = LOCAL ? wg : (int)blockIdx.x;           // my exchange record
  // chip-wide launches: the exchange is a tree over the XCDs (grid_exchange8_hier); where am I in it?
  constexpr bool kHier = !LOCAL;
  static_assert(!SLAB || kHier, "the node level of the slab instance rides on the tree exchange");
  unsigned hx = 0;
  constexpr bool kLocalAll = LOCAL;     // XCD-local launches: every wave polls the group's records itself
  __shared__ int hier_s[(kHier || kLocalAll) ? 4 : 1];
  if constexpr (kLocalAll) { if (threadIdx.x == 0) hier_s[2] = 0; }          // (the sticky flag; the barriers of the set-up below publish it)
  if constexpr (kHier) {
    hx = hier_enter(c, hier_s);
    if (hier_s[1]) return;                                 // (every workgroup of the launch fails this the same way)
  }
  int j0[NQ], tx0[NQ];
  bool has[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int reg = (wg * c.waves + wave) * NQ + q;
    has[q] = wave < c.waves && reg < c.nreg;
    const int ty = has[q] ? reg / c.ntx : 0;
    tx0[q] = has[q] ? reg - ty * c.ntx : 0;
    j0[q] = ty * R;
  }
  const unsigned nbytesT = (unsigned)((size_t)nx * ny * sizeof(T)), nbytesC = (unsigned)((size_t)nx * ny * sizeof(CT));
  const unsigned rowT = (unsigned)(nx * sizeof(T)), rowC = (unsigned)(nx * sizeof(CT));
  // SLAB: does region q touch the lower / upper edge of the slab, and is there a GPU beyond it (else: a wall of the global grid)?
  // ONE scalar of edge bits per region, opaque to the optimiser (it would expand them into flags that live in SGPRs across the
  // loop): bit 0 first row = the slab's first row AND a GPU lies below, bit 1 last row = the slab's last row AND a GPU lies above,
  // bit 2 / 3 first / last row = the slab's first / last row.  Everything else the two edge waves of a slab need - mailbox
  // addresses, the row capacity - is fetched from the kernarg segment where it is used (karg): it used to occupy ~30 SGPRs in
  // EVERY wave, which came back as v_readlane reloads in a loop that is bound by VALU issue (113 spilled SGPRs, 17-19 us per
  // iteration against 11 for the single-GPU kernel).
  typedef Persist1Kargs<T, std::conditional_t<SLAB, SlabCtl, NoSlab>> KArgs;
  constexpr unsigned sl_off = (unsigned)offsetof(KArgs, sl);
  constexpr unsigned pv_off = sl_off + (unsigned)offsetof(SlabCtl, pv);
  unsigned ef[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    ef[q] = 0;
    if constexpr (SLAB) {
      const bool bot = has[q] && j0[q] == 0, top = has[q] && j0[q] + R == ny;
      ef[q] = (bot && sl.pv.lower >= 0 ? 1u : 0u) | (top && sl.pv.upper >= 0 ? 2u : 0u) | (bot ? 4u : 0u) | (top ? 8u : 0u);
      ef[q] = (unsigned)__builtin_amdgcn_readfirstlane((int)ef[q]);
      asm volatile("" : "+s"(ef[q]));
    }
  }
  {                                                         // the mirror of the argument list IS the argument list (every instance: the exit block relies on it)
    typedef CgArgs<T> A;
    if (karg<T*>((unsigned)offsetof(KArgs, a) + (unsigned)offsetof(A, r)) != a.r || karg<int*>((unsigned)offsetof(KArgs, c) + (unsigned)offsetof(PersistCtl, err)) != c.err) {
      if (threadIdx.x == 0) *c.err = 1;
      return;
    }
  }
  if constexpr (SLAB) {
    if (karg<char*>(sl_off + (unsigned)offsetof(SlabCtl, rows_own)) != sl.rows_own || karg<size_t>(pv_off + (unsigned)offsetof(PeerView, row_cap)) != sl.pv.row_cap) {
      if (threadIdx.x == 0) *c.err = 1;
      return;
    }
  }
  const unsigned nbytesH = nbytesT + 2 * rowT;              // r / p[] including their halo rows (resources built where they are used)
  // resource of the eight rows of a mailbox (host-level exchange + z' halos) and the byte offset of z' halo row (parity, side)
  // inside them: built by the edge waves where they use them
  // (on = false: a resource of no bytes - every access through it is dropped by the range check: the edge rows' mailbox stores
  // sit in the row loop without a branch; control flow in the middle of the unrolled loop cost the allocator 95 spilled VGPRs)
  auto mailbox_rows = [&](unsigned field_off, int parity, int side, unsigned& zoff, bool on = true) __attribute__((always_inline)) -> rsrc_t {
    const unsigned cap8 = (unsigned)karg<size_t>(pv_off + (unsigned)offsetof(PeerView, row_cap)) * 8u;
    zoff = (unsigned)(4 + parity * 2 + side) * cap8;
    return make_rsrc(karg<char*>(sl_off + field_off), on ? 8u * cap8 : 0u);
  };
  const rsrc_t RoS = make_rsrc(a.oS, nbytesC), RoW = make_rsrc(a.oW, nbytesC), RoE = make_rsrc(a.oE, nbytesC), RoN = make_rsrc(a.oN, nbytesC);
  const rsrc_t RcC = make_rsrc(a.cC, nbytesT);
  // p[0] / p[1]: the direction of the two-kernel path, read on entry and written on exit.  zp[0] / zp[1]: the published z'
  // perimeters, touched by agent-scope (sc1) stores and loads only - never by a plain load, whose copy of a line in the reader
  // XCD's L2 another XCD's write-through store does not invalidate.
  // (r, x, p[]: resources built at the entry and again at the exit - a resource that stays alive across the loop costs 4 SGPRs,
  // and spilled SGPRs come back through v_readlane: VALU slots of a loop that is bound by VALU issue)
  const rsrc_t Rz0 = make_rsrc(a.zp[0], nbytesT), Rz1 = make_rsrc(a.zp[1], nbytesT);
  auto row_wrap = [&](int j, bool& valid) __attribute__((always_inline)) -> int {   // scalar: rows outside wrap or vanish
    valid = true;
    if (j < 0) { if (!a.per_y) valid = false; return ny - 1; }
    if (j >= ny) { if (!a.per_y) valid = false; return 0; }
    return j;
  };

  // ---- the state of the two-kernel path: r and the direction p_{k-1} of my regions into registers, x into LDS
  const T alpha0 = pend ? uniform(a.scal[SC_ALPHA]) : (T)0;   // pend: x still lacks alpha p of the iteration before k_begin
  Vec<T, V> rr[NQ][R], pp[NQ][R];
  // kPack (regions of more than two rows): the END CELLS of a region's rows - what the strips to the left and right read - are
  // published as one packed block per region instead of in place: entry (side, row) = the 16 bytes of lane 0 (side 0) or lane 63
  // (side 1), 2 x R x 16 bytes at the start of the region's SECOND row of the z' buffer (interior rows of that buffer are
  // otherwise unused).  In place every end cell dirtied a cache line of its own in the writer's L2 and was fetched as a line of
  // its own by the reader: 28 + 28 lines per region and iteration against 4 + 4 - and ~1 MB per XCD of L2 capacity that the
  // coefficient rows (4.19 MB per XCD at 2048^2 on a 4 MB L2) were missing.
  constexpr bool kPack = R > 2 && V == 2;
  unsigned vT[NQ], vEnd[NQ];                               // vEnd: where lanes 0 and 63 publish their end cells, beyond any buffer elsewhere
  // copies of r (registers) and of the direction (LDS / registers, `edge`) on the ring around my regions
  Vec<T, V> rhb[NQ], rha[NQ], pnb[NQ], pna[NQ];
  T eR[NQ], edge[NQ];
  bool vb[NQ], va[NQ], vl[NQ], vr[NQ];                      // is there a cell below / above / left / right of region q at all? (entry only)
  unsigned nbits[NQ];                                       // ... the same four as bits of one scalar (the loop's copy)
  auto nbit = [&](int q, unsigned bit) __attribute__((always_inline)) -> bool {
    unsigned b = nbits[q];
    asm volatile("" : "+s"(b));
    return (b & bit) != 0;
  };
  {
    const rsrc_t Rr = make_rsrc(a.r, nbytesT), Rx = make_rsrc(a.x, nbytesT), Rp = make_rsrc(a.p[k_begin & 1], nbytesT);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int cq = (tx0[q] * 64 + lane) * V;
      vT[q] = (unsigned)(cq * sizeof(T));
      vEnd[q] = (lane == 0 || lane == 63) ? vT[q] : 0x80000000u;
      if constexpr (kPack) vEnd[q] = lane == 0 ? (unsigned)(tx0[q] * 64 * V * sizeof(T)) : (lane == 63 ? (unsigned)(tx0[q] * 64 * V * sizeof(T)) + (unsigned)(R * 16) : 0x80000000u);
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
      // the ring: rows below / above (out of range beyond a wall -> 0) and the two neighbouring columns
      // (lane l < R: left neighbour of row l, lane R + l: right neighbour; other lanes and walls read 0)
      const int jb = row_wrap(j0[q] - 1, vb[q]), ja = row_wrap(j0[q] + R, va[q]);
      if constexpr (SLAB) {                                  // no wrap inside a slab: beyond its edges lies a neighbour's row (or a wall)
        vb[q] = !(ef[q] & 4u) || (ef[q] & 1u); va[q] = !(ef[q] & 8u) || (ef[q] & 2u);
        const unsigned hb = (has[q] && vb[q]) ? vT[q] : 0xffffffffu, ha = (has[q] && va[q]) ? vT[q] : 0xffffffffu;
        const rsrc_t RrH = make_rsrc(a.r - nx, nbytesH), RpH = make_rsrc(a.p[k_begin & 1] - nx, nbytesH);
        rhb[q] = bld<T, V>(RrH, hb, (unsigned)j0[q] * rowT);            // (the halo-based resources start one row lower)
        rha[q] = bld<T, V>(RrH, ha, (unsigned)(j0[q] + R + 1) * rowT);
        pnb[q] = bld<T, V>(RpH, hb, (unsigned)j0[q] * rowT);
        pna[q] = bld<T, V>(RpH, ha, (unsigned)(j0[q] + R + 1) * rowT);
        if constexpr (kParkHalos) {                          // parked at once: no register of the entry survives into the loop
          T* hs = halo_s + (size_t)((wave * NQ + q) * 2) * 64 * V + lane * V;
          stv<T, V>(hs, pnb[q]);
          stv<T, V>(hs + 64 * V, pna[q]);
        }
      } else {
      const unsigned hb = (has[q] && vb[q]) ? vT[q] : 0xffffffffu, ha = (has[q] && va[q]) ? vT[q] : 0xffffffffu;
      rhb[q] = bld<T, V>(Rr, hb, (unsigned)jb * rowT);
      rha[q] = bld<T, V>(Rr, ha, (unsigned)ja * rowT);
      pnb[q] = bld<T, V>(Rp, hb, (unsigned)jb * rowT);
      pna[q] = bld<T, V>(Rp, ha, (unsigned)ja * rowT);
      if constexpr (kParkHalos) {
        T* hs = halo_s + (size_t)((wave * NQ + q) * 2) * 64 * V + lane * V;
        stv<T, V>(hs, pnb[q]);
        stv<T, V>(hs + 64 * V, pna[q]);
      }
      }
      int side, er;
      ring_lane(lane, R, side, er);
      int cc = (side == 0) ? tx0[q] * 64 * V - 1 : (tx0[q] + 1) * 64 * V;
      vl[q] = tx0[q] > 0 || a.per_x;
      vr[q] = tx0[q] + 1 < c.ntx || a.per_x;
      // ONE scalar of neighbour bits per region for the loop (bit 0 / 1: cells below / above, bit 2 / 3: left / right, bits 8.. / 16..:
      // the rows the halo loads read), opaque to the optimiser and made opaque again at every use: as four booleans per region
      // they lived in eight SGPR pairs across the row loops (spilled), and the selects "resource of no bytes beyond a wall" were
      // hoisted out of the loop as VECTOR values - the halo loads' resources were then rebuilt in four waterfall loops per iteration
      nbits[q] = (vb[q] ? 1u : 0u) | (va[q] ? 2u : 0u) | (vl[q] ? 4u : 0u) | (vr[q] ? 8u : 0u);
      nbits[q] = (unsigned)__builtin_amdgcn_readfirstlane((int)nbits[q]);
      asm volatile("" : "+s"(nbits[q]));
      if (cc < 0) cc = a.per_x ? nx - 1 : -1;
      else if (cc >= nx) cc = a.per_x ? 0 : -1;
      const unsigned vo = (has[q] && side < 2 && cc >= 0) ? (unsigned)(j0[q] + er) * rowT + (unsigned)(cc * sizeof(T)) : 0xffffffffu;
      eR[q] = bld1<T>(Rr, vo, 0);
      edge[q] = bld1<T>(Rp, vo, 0);
    }
  }
  CgState st = a.state[sv & 1];
  T pz = uniform(a.scal[SC_PZ]), vs = uniform(a.scal[SC_VS]), alpha = uniform(a.scal[SC_ALPHA]);
  const T sc_c = uniform(a.scal[SC_C]);
  T ncells = (T)((double)nx * (double)ny);
  if constexpr (SLAB) ncells = (T)sl.ncells;
  if constexpr (RAGGED) ncells = (T)a.ncells;
  ncells = uniform(ncells);                                 // (in scalar registers, not in a vector register pair held across the loop)
  // RAGGED: are my columns / the columns next to my strip / the rows around my regions cells of the true system?
  bool col_ok[NQ], lcol_ok[NQ], rcol_ok[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    col_ok[q] = !RAGGED || (tx0[q] * 64 + lane) * V < a.nx_true;         // (nx_true is a multiple of V: a lane's cells are all in or all out)
    lcol_ok[q] = !RAGGED || tx0[q] * 64 * V - 1 < a.nx_true;
    rcol_ok[q] = !RAGGED || (tx0[q] + 1) * 64 * V < a.nx_true;
  }
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
    if constexpr (SLAB) {                                   // the all-reduced totals of the previous K2 / segment
#pragma unroll
      for (int q = 0; q < 3; ++q) tB[q] = uniform(a.gB[q]);
    }
  }

  // ---- coefficient pipeline (as in cg_persist): both passes of an iteration stream the coefficient rows of my regions in
  // the same order; the loads of row t + D are issued when row t has been consumed, circularly.
  constexpr int coef_regs = ((SYM ? 2 : 4) * (int)sizeof(CT) * V + (RECON ? 0 : (int)sizeof(T) * V)) / 4;   // VGPRs of a row in flight
  constexpr int NT = NQ * R;
  // (the deep pipeline pays where the loop has registers to spare: the slab variant and the four-array / fp64-coefficient variants
  // keep the round-2 depth, they spill otherwise)
  constexpr bool kDeep = NQ == 1 && NT == 16 && SYM && RECON && sizeof(T) == 8;
  constexpr int depth_max = kDeep ? PISO_PERSIST1_DEPTH : ((NQ == 1) ? 3 : kPersistMaxDepth);
  // (small regions whose rows carry the diagonal too - systems with open boundaries, BASELINE config 4: 8 registers per row - get the
  // registers for FOUR rows in flight as well: with regions of 2 rows that is every row of the wave, i.e. the coefficients stay
  // resident as they do with rebuilt diagonals; with two rows in flight both passes waited for L2: 1024 x 256 4.87 against 4.10 us)
  constexpr int budget = kDeep ? 4 * PISO_PERSIST1_DEPTH : ((NQ == 1 || NT < 16) ? ((NT < 16 && !RECON && SYM) ? 32 : 16) : 8);
  constexpr int Dw = budget / coef_regs < 2 ? 2 : (budget / coef_regs > depth_max ? depth_max : budget / coef_regs);
  constexpr int D = (NT >= Dw) ? Dw : NT;
  constexpr int kBaseLoads = (SYM ? 2 : 4) + (RECON ? 0 : 1);          // vector loads every row issues (some rows one or two more)
  Vec<CT, V> cS[NT], cW[NT], cE[NT], cN[NT], cSh[NQ];
  Vec<T, V> cD[NT];
  CT eW[NQ];
  // byte offset of row j0[q] + jj: recomputed at every use (two scalar instructions) from a value the optimiser cannot see
  // through - hoisted out of the unrolled row loops the 2 x 16 products live in SGPRs that spill, and a spilled SGPR comes back
  // through v_readlane, a VALU slot of a loop that is bound by VALU issue
  auto row_base = [&](int q, int jj, unsigned row_bytes) __attribute__((always_inline)) -> unsigned {
    unsigned j = (unsigned)j0[q];
    asm volatile("" : "+s"(j));
    return (j + (unsigned)jj) * row_bytes;
  };
  auto coef_offset = [&](int q) __attribute__((always_inline)) -> unsigned {
    return (unsigned)((unsigned long long)vT[q] * sizeof(CT) / sizeof(T));
  };
  // Both stencil passes stream the S / W rows of the workgroup's cells, 4.19 MB per XCD and pass at 2048^2 - just above the 4 MB of
  // an XCD's L2, so a cyclic sweep misses almost every time and the row loops run at the speed of the memory fabric (measured 33.5 MB
  // in 3.6 us).  Hence the back-and-forth order: D walks the rows of a wave upwards (t = 0 .. NT-1), U walks them downwards.  The coefficient rows a pass ends with are
  // the rows the next pass starts with: they are still in registers (Dc rows per turn are never reloaded) and the rows behind them
  // are the most recently used lines of the XCD's L2 - a cyclic sweep over a set just above the L2's capacity misses every time,
  // a back-and-forth sweep misses only what does not fit.  (Measured before: both row loops ran at the speed of the memory
  // fabric, ~6 TB/s of coefficient rows, not at the speed of their arithmetic.)
  // kAhead: z' = L p of the first rows of the U pass does not depend on alpha - it is computed WHILE the exchange's records travel
  // (behind the publish, in front of the polling: the SIMDs have nothing else to do for a microsecond or two) and kept in
  // registers; U then skips the stencil of those rows.  With the back-and-forth order these are the rows D ended with: their
  // coefficients are still in registers.  Same instructions on the same registers as D's z': bitwise the same values.
  constexpr int kAhead = (kHier && NQ == 1 && NT == 16 && SYM && RECON && sizeof(T) == 8 && sizeof(CT) == 4) ? PISO_PERSIST1_AHEAD : 0;
  static_assert(kAhead <= Dw || kAhead == 0, "the rows computed ahead are rows whose coefficients D left in registers");
  constexpr bool kShLate = kAhead > 0 && !SLAB && NQ == 1 && SYM;
  // kKeepZ (small regions: a wave's rows are few): z' of D stays in registers until U has used it - U runs no stencil at all (the
  // 16-row instances have no registers for it: they compute z' twice, bitwise the same, and kAhead moves part of that off the path)
  constexpr bool kKeepZ = NT * (int)sizeof(T) * V / 4 <= PISO_PERSIST1_KEEP_Z_REGS;
  // kCountLate (16-row regions, round 6): the residual count is taken in a pass of its own, only in the iterations whose count the
  // stopping test reads ((k + 1) % 5 == 0): 32 fp64 compares less per wave in 4 of 5 iterations of a loop bound by VALU issue
  // (A/B on one box, four rounds: 10.19 - 10.32 against 10.22 - 10.48 us per iteration).  Small regions keep the count in the row loop:
  // their iteration is a latency chain, a branch more on it buys nothing.
  constexpr bool kCountLate = NT == 16 && !RAGGED;
  auto issue_coef = [&](int t) __attribute__((always_inline)) {
    const int q = t / R, jj = t - q * R;
    const unsigned vCq = coef_offset(q);
    const unsigned sT = row_base(q, jj, rowT), sC = row_base(q, jj, rowC);
    cS[t] = bld<CT, V>(RoS, vCq, sC); cW[t] = bld<CT, V>(RoW, vCq, sC);
    if constexpr (!SYM) { cE[t] = bld<CT, V>(RoE, vCq, sC); cN[t] = bld<CT, V>(RoN, vCq, sC); }
    if constexpr (!RECON) cD[t] = bld<T, V>(RcC, vT[q], sT);
  };
  // SYM: W of the first column of the strip to the right (E of my last column; lane R + jj: row jj) and S of the row above the
  // region (N of my last row).  Constants of the launch: loaded ONCE - reloading them with rows 0 / R-1 of every pass put a full
  // memory trip in front of the rows that still used them (0.55 us per iteration at 2048^2).
  // LDS byte address of this wave's ring block, kept in a VGPR (ds instructions take their address from one) and made opaque
  // before every use: the optimiser then cannot merge the reads of the D pass with those of the U pass (which would keep 5
  // registers per row alive across the exchange) and leaves each read where it is issued
  typedef __attribute__((address_space(3))) T lds_T;
  typedef __attribute__((address_space(3))) CT lds_CT;
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  unsigned ring_a = (unsigned)(unsigned long)(lds_u8*)ring_s + (unsigned)(wave * NQ * kRingBytes);
  asm volatile("" : "+v"(ring_a));
  if constexpr (kParkRing) {                                 // the entry's ring columns go to the wave's block at once
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const unsigned rb = ring_a + (unsigned)(q * kRingBytes) + (unsigned)lane * (unsigned)sizeof(T);
      *(lds_T*)(unsigned long)rb = edge[q];
      *(lds_T*)(unsigned long)(rb + kRingR) = eR[q];
    }
  }
  if constexpr (SYM) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      int side, er;
      ring_lane(lane, R, side, er);
      int cc = (tx0[q] + 1) * 64 * V;
      if (cc >= nx) cc = a.per_x ? 0 : -1;
      const unsigned vo = (has[q] && side == 1 && cc >= 0) ? (unsigned)(j0[q] + er) * rowC + (unsigned)(cc * sizeof(CT)) : 0xffffffffu;
      eW[q] = bld1<CT>(RoW, vo, 0);
      if constexpr (kRingLds)
        *(lds_CT*)(unsigned long)(ring_a + (unsigned)(q * kRingBytes + 64 * (int)sizeof(T)) + (unsigned)lane * (unsigned)sizeof(CT)) = eW[q];
      bool valid;
      const int jw = row_wrap(j0[q] + R, valid);
      if (SLAB && (ef[q] & 8u)) cSh[q] = bld<CT, V>(RoN, has[q] ? coef_offset(q) : 0xffffffffu, (unsigned)(ny - 1) * rowC);   // N of my last row
      else
      cSh[q] = bld<CT, V>(RoS, (has[q] && valid) ? coef_offset(q) : 0xffffffffu, (unsigned)jw * rowC);
    }
  }
  // kShLate (with kAhead, one 16-row region, no slab edge): S of the row above the region is requested again in every D pass together
  // with the coefficients of the region's last row - its registers are then free from the rows computed ahead to the next request
  // (across U and the first rows of D) instead of held for the whole launch
  auto reload_csh = [&]() __attribute__((always_inline)) {
    if constexpr (SYM && NQ == 1) {
      bool valid;
      const int jw = row_wrap(j0[0] + R, valid);
      cSh[0] = bld<CT, V>(RoS, (has[0] && valid) ? coef_offset(0) : 0xffffffffu, (unsigned)jw * rowC);
    }
  };
  // the ring values of row t (every lane receives them; lanes 0 / 63 are the ones that matter): issued one row ahead
  T rgl[NT], rgr[NT];
  CT rgw[NT];
  auto ring_issue = [&](int t) __attribute__((always_inline)) {
    if constexpr (!kRingLds) return;
    const int q = t / R, jj = t - q * R;
    asm volatile("" : "+v"(ring_a));
    const unsigned base = ring_a + (unsigned)(q * kRingBytes);
    rgl[t] = *(const lds_T*)(unsigned long)(base + (unsigned)(jj * (int)sizeof(T)));
    rgr[t] = *(const lds_T*)(unsigned long)(base + (unsigned)((48 + jj) * (int)sizeof(T)));
    if constexpr (SYM) rgw[t] = *(const lds_CT*)(unsigned long)(base + (unsigned)(64 * (int)sizeof(T) + (48 + jj) * (int)sizeof(CT)));
  };
  // z' = L p of row t of my regions: summation order of calcZ_v4 (pressure_solve_op.cu.cc:81-90).  D and U both call this on
  // the same registers, so they see bitwise the same z'.
  auto zrow = [&](int t) __attribute__((always_inline)) -> Vec<T, V> {
    const int q = t / R, jj = t - q * R;
    T* hs = halo_s + (kParkHalos ? (size_t)((wave * NQ + q) * 2) * 64 * V + lane * V : 0);
    const Vec<T, V> behind = (jj > 0) ? pp[q][jj > 0 ? jj - 1 : 0] : (kParkHalos ? ldv<T, V>(hs) : pnb[q]);
    const Vec<T, V> cur = pp[q][jj];
    const Vec<T, V> ahead = (jj + 1 < R) ? pp[q][jj + 1 < R ? jj + 1 : jj] : (kParkHalos ? ldv<T, V>(hs + (kParkHalos ? 64 * V : 0)) : pna[q]);
    const T left = kRingLds ? shift_lane<true, T>(cur.v[V - 1], rgl[t]) : shift_ring<true, T>(cur.v[V - 1], edge[q], jj);
    const T right = kRingLds ? shift_lane<false, T>(cur.v[0], rgr[t]) : shift_ring<false, T>(cur.v[0], edge[q], jj);
    Vec<CT, V> kN, kE;
    if constexpr (SYM) {
      kN = (jj + 1 < R) ? cS[t + 1 < NT ? t + 1 : t] : cSh[q];
#pragma unroll
      for (int e = 0; e + 1 < V; ++e) kE.v[e] = cW[t].v[e + 1];
      kE.v[V - 1] = kRingLds ? shift_lane<false, CT>(cW[t].v[0], rgw[t]) : shift_ring<false, CT>(cW[t].v[0], eW[q], jj);
    } else {
      kN = cN[t]; kE = cE[t];
    }
    Vec<T, V> kC, z;
    if constexpr (RECON) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        T d = -(T)cS[t].v[e] - (T)kN.v[e];                    // = (0 - S) - N: the first difference of laplace_op.cu.cc:118-135 is exact
        d -= (T)cW[t].v[e]; d -= (T)kE.v[e];
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
  Vec<T, V> zk[kKeepZ ? NT : 1];                             // kKeepZ: z' of my rows from D to U
  int k = k_begin;                                           // the iteration (SLAB: its parity picks the mailbox rows)
  // perimeter of row jj of region q (what neighbouring regions read): the whole first / last row, else the two end cells
  auto publish = [&](rsrc_t Rd, int q, int jj, const Vec<T, V>& val) __attribute__((always_inline)) {
    const unsigned sT = row_base(q, jj, rowT);
    if (jj == 0 || jj == R - 1) {
      bst<T, V, kPub>(Rd, vT[q], sT, val);
      if constexpr (kPack) bst<T, V, kPub>(Rd, vEnd[q], row_base(q, 1, rowT) + (unsigned)(jj * 16), val);    // (its end cells into the packed block as well)
      if constexpr (SLAB) {
        // my first row is the row ABOVE the lower neighbour's slab (its side 1), my last row the row BELOW the upper one's (side 0)
        if (jj == 0) {
          unsigned zoff;
          const rsrc_t Rm = mailbox_rows((unsigned)offsetof(SlabCtl, rows_lo), k & 1, 1, zoff, (ef[q] & 1u) != 0);
          bst<T, V, kSystem>(Rm, vT[q], zoff, val);
        }
        if (jj == R - 1) {
          unsigned zoff;
          const rsrc_t Rm = mailbox_rows((unsigned)offsetof(SlabCtl, rows_hi), k & 1, 0, zoff, (ef[q] & 2u) != 0);
          bst<T, V, kSystem>(Rm, vT[q], zoff, val);
        }
      }
    } else {
      // the two end cells of the row: lanes 0 and 63 store their cells of the row (16 bytes each; a lane's inner cell is written
      // too and never read), every other lane carries an offset beyond the buffer and its store is dropped by the range check -
      // no exec mask to build, no data to select (2 compares + 2 selects per row in a loop that is bound by VALU issue)
      if constexpr (kPack) {
        bst<T, V, kPub>(Rd, vEnd[q], row_base(q, 1, rowT) + (unsigned)(jj * 16), val);
      } else {
        bst<T, V, kPub>(Rd, vEnd[q], sT, val);
      }
    }
  };
  // ---- z' on the ring, as published by the neighbours (rows below / above, the two neighbouring columns): issued right after
  // the exchange (every record seen = every perimeter store completed), consumed at the END of U - the row loop hides the trip
  T eZ[NQ];
  Vec<T, V> hbZ[NQ], haZ[NQ];
  auto issue_halos = [&](rsrc_t Rz) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      int side, er, lane_now = lane;
      asm volatile("" : "+v"(lane_now));                    // (the ring offset is recomputed here: hoisted, it is one more register across the row loops - spilled)
      ring_lane(lane_now, R, side, er);
      int cc = (side == 0) ? tx0[q] * 64 * V - 1 : (tx0[q] + 1) * 64 * V;
      if (cc < 0) cc = a.per_x ? nx - 1 : -1;
      else if (cc >= nx) cc = a.per_x ? 0 : -1;
      unsigned vo = (side < 2 && cc >= 0) ? (unsigned)(j0[q] + er) * rowT + (unsigned)(cc * sizeof(T)) : 0xffffffffu;
      if constexpr (kPack) {
        // the neighbouring strip's packed block: its side-1 entry (its lane 63, whose last cell is my left neighbour) or its side-0
        // entry (its lane 0, first cell), row `er`
        const int strip = (cc >= 0) ? cc / (64 * V) : 0;
        const unsigned ent = (side == 0) ? (unsigned)(R * 16 + er * 16 + (V - 1) * (int)sizeof(T)) : (unsigned)(er * 16);
        vo = (side < 2 && cc >= 0) ? (unsigned)(j0[q] + 1) * rowT + (unsigned)(strip * 64 * V * (int)sizeof(T)) + ent : 0xffffffffu;
      }
      eZ[q] = bld1<T, kHalo>(Rz, vo, 0);
      bool vbq_unused, vaq_unused;
      const int jb = row_wrap(j0[q] - 1, vbq_unused), ja = row_wrap(j0[q] + R, vaq_unused);
      const bool vbq = nbit(q, 1u), vaq = nbit(q, 2u);
      if constexpr (SLAB) {                                  // (wave-uniform branches: no per-lane offset registers to keep)
        if (!(ef[q] & 4u)) hbZ[q] = bld<T, V, kHalo>(Rz, vT[q], (unsigned)(j0[q] - 1) * rowT);
        else if (ef[q] & 1u) {
          unsigned zoff;
          const rsrc_t Rm = mailbox_rows((unsigned)offsetof(SlabCtl, rows_own), k & 1, 0, zoff);
          hbZ[q] = bld<T, V, kSystem>(Rm, vT[q], zoff);
        } else {
#pragma unroll
          for (int e = 0; e < V; ++e) hbZ[q].v[e] = 0;
        }
        if (!(ef[q] & 8u)) haZ[q] = bld<T, V, kHalo>(Rz, vT[q], (unsigned)(j0[q] + R) * rowT);
        else if (ef[q] & 2u) {
          unsigned zoff;
          const rsrc_t Rm = mailbox_rows((unsigned)offsetof(SlabCtl, rows_own), k & 1, 1, zoff);
          haZ[q] = bld<T, V, kSystem>(Rm, vT[q], zoff);
        } else {
#pragma unroll
          for (int e = 0; e < V; ++e) haZ[q].v[e] = 0;
        }
      } else {
      // (beyond a wall: a resource of no bytes - the load is dropped by the range check and returns 0; a per-lane offset that says
      // the same is one more vector register across the row loops)
      hbZ[q] = bld<T, V, kHalo>(make_rsrc(a.zp[k & 1], (unsigned)__builtin_amdgcn_readfirstlane((int)(vbq ? nbytesT : 0u))), vT[q], (unsigned)jb * rowT);
      haZ[q] = bld<T, V, kHalo>(make_rsrc(a.zp[k & 1], (unsigned)__builtin_amdgcn_readfirstlane((int)(vaq ? nbytesT : 0u))), vT[q], (unsigned)ja * rowT);
      }
    }
  };
  if (has[0]) {
#pragma unroll
    for (int t = 0; t < D; ++t) issue_coef(t);
  }

  unsigned epoch = c.epoch0;
  bool healthy = true, first = true;
  unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tsub[4] = {0, 0, 0, 0}, tlast = (kPersistDiag && c.timing) ? wall_clock64() : 0;
  auto tick = [&](int slot) __attribute__((always_inline)) {     // diagnostic builds only (-DPISO_PERSIST_DIAG): D / exchange / U clocks
    if (kPersistDiag && c.timing) { const unsigned long long t = wall_clock64(); tacc[slot] += t - tlast; tlast = t; }
  };
  // the stopping test of iteration k_begin was left to this launch by the previous one (pressure_solve_op.cu.cc:312-335)
  if (!st.done && k > 0 && (k % 5) == 0) {                  // (!done: a launch queued behind a converged one changes nothing)
    const int exceeded = tB[2] > 0;
    if (st.flag && !exceeded) { st.done = 1; st.iterations = k; }
    else st.flag = 1;
  }
  T rz_next = tB[0], sumr = tB[1], cnt_last = tB[2];          // r_k.z'_{k-1}, sum r_k, #{|r_k| >= accuracy} for the current k
  T lU[2] = {0, 0};                                           // local sum r, count from the last U (for the next exchange)
  const T accuracy = uniform((T)a.accuracy);
  // beta of iteration k needs nothing but the sums of exchange k - 1: it is computed right behind them, in the shadow of U's wait for
  // the neighbours' rows, instead of as a dependent chain of ~25 instructions in front of D (same expression, same operands)
  T beta = uniform(-(rz_next + vs * sumr) / pz);              // (:351-352), unguarded as coded
  for (; k < k_end && healthy && !st.done; ++k) {
    // consecutive iterations alternate buffers: a neighbour's loads of z'_k are consumed before it publishes its record k + 1,
    // which everybody needs before writing the same buffer again in iteration k + 2
    rsrc_t Rz;
    // (16-row regions and the slab instances are short of scalar registers: the buffer's address comes from the kernarg segment, not
    // from four SGPRs held across the loop - 83 -> 58 spilled SGPRs in the 16-row slab instance, 28 -> 23 in the plain one.  Small
    // regions keep the descriptors: their iteration is a latency chain and the scalar load sits on it - 256^2 2.70 -> 2.79 us)
    if constexpr (SLAB || NT == 16) {
      typedef CgArgs<T> A;
      Rz = make_rsrc(karg<T*>((unsigned)offsetof(KArgs, a) + (unsigned)offsetof(A, zp) + 8u * (unsigned)(k & 1)), nbytesT);
    } else {
      Rz = (k & 1) ? Rz1 : Rz0;
    }
    // ---- D(k): p = r + beta p on my cells and on the ring; z' = L p; sums; the perimeter of z' goes out
    T sD[kX1Values] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (has[0]) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int jj = 0; jj < R; ++jj) {
#pragma unroll
          for (int e = 0; e < V; ++e) pp[q][jj].v[e] = fma(beta, pp[q][jj].v[e], rr[q][jj].v[e]);
        }
        if constexpr (kParkRing) {
          const unsigned rb = ring_a + (unsigned)(q * kRingBytes) + (unsigned)lane * (unsigned)sizeof(T);
          const T e_old = *(lds_T*)(unsigned long)rb, r_col = *(lds_T*)(unsigned long)(rb + kRingR);
          *(lds_T*)(unsigned long)rb = fma(beta, e_old, r_col);
        } else {
        edge[q] = fma(beta, edge[q], eR[q]);
        if constexpr (kRingLds) *(lds_T*)(unsigned long)(ring_a + (unsigned)(q * kRingBytes) + (unsigned)lane * (unsigned)sizeof(T)) = edge[q];
        }
        T* hs = halo_s + (kParkHalos ? (size_t)((wave * NQ + q) * 2) * 64 * V + lane * V : 0);
        if constexpr (kParkHalos) {
          pnb[q] = ldv<T, V>(hs); pna[q] = ldv<T, V>(hs + 64 * V);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
          pnb[q].v[e] = fma(beta, pnb[q].v[e], rhb[q].v[e]);
          pna[q].v[e] = fma(beta, pna[q].v[e], rha[q].v[e]);
        }
        if constexpr (kParkHalos) {
          stv<T, V>(hs, pnb[q]);
          stv<T, V>(hs + 64 * V, pna[q]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      ring_issue(0);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int q = t / R, jj = t - q * R;
        if (t + 1 < NT) ring_issue(t + 1);
        const Vec<T, V> z = zrow(t);
        if constexpr (kKeepZ) zk[t] = z;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          sD[0] += pp[q][jj].v[e];
          sD[1] = fma(pp[q][jj].v[e], rr[q][jj].v[e], sD[1]);
          sD[2] = fma(pp[q][jj].v[e], z.v[e], sD[2]);
          sD[3] = fma(rr[q][jj].v[e], z.v[e], sD[3]);
          sD[4] = fma(z.v[e], z.v[e], sD[4]);
          sD[5] += z.v[e];
        }
        publish(Rz, q, jj, z);
        PISO_SB_A1;
        // (rows 0 .. D-1 of the U pass are issued behind the drain of the perimeter stores, see the exchange: issued here they would
        // be in flight when the wave waits for vmcnt(0), and the wait would cover their trip as well)
        if constexpr (D < NT) { if (t + D < NT) issue_coef(t + D); }
        if constexpr (kShLate) { if (t + D == NT - 1) reload_csh(); }
        PISO_SB_A2;
      }
    }
    sD[6] = lU[0]; sD[7] = lU[1];
    ++epoch;
    tick(0);
    Vec<T, V> zs[kAhead > 0 ? kAhead : 1];
    auto z_ahead = [&]() __attribute__((always_inline)) {
      if constexpr (kAhead > 0) {
        if (has[0]) {
          ring_issue(NT - 1);
#pragma unroll
          for (int tt = 0; tt < kAhead; ++tt) {
            if (tt + 1 < kAhead) ring_issue(NT - 2 - tt);
            zs[tt] = zrow(NT - 1 - tt);
          }
        }
      }
    };
    if constexpr (kHier) healthy = grid_exchange8_hier<T, (SLAB ? kPollDelay2Xg : (kAhead > 0 ? kPollDelay2 : kPollDelay2NoAhead)), SLAB>(c, sD, epoch, smem, hx, hier_s + 2, z_ahead, (kPersistDiag && c.timing) ? tsub : nullptr, sl_off, smem + 2 * kX1Sm);
    else if constexpr (kLocalAll) healthy = grid_exchange8_local<T>(c, sD, epoch, smem, slot, nslots, hier_s + 2, (kPersistDiag && c.timing) ? tsub : nullptr);
    else healthy = grid_exchange8<T, LOCAL>(c, sD, epoch, smem, slot, nslots, NoPrefetch(), (kPersistDiag && c.timing) ? tsub : nullptr);
    tick(1);
    if (!healthy) break;
    // ---- the stopping test of iteration k, one exchange late but before anything moves (x = x_k): (:312-335)
    if (!first) {
      sumr = sD[6]; cnt_last = sD[7];
      if (k > 0 && (k % 5) == 0) {
        const int exceeded = cnt_last > 0;
        if (st.flag && !exceeded) { st.done = 1; st.iterations = k; }
        else st.flag = 1;
      }
      if (st.done) break;
    }
    first = false;
    // ---- alpha (:301-302) and, by one-step recurrences from the direct sums, what beta of the next iteration needs
    vs = uniform(sc_c * sD[0]);
    pz = uniform(sD[2] + vs * sD[0]);
    alpha = uniform((absval(pz) > 0) ? sD[1] / pz : (T)0);
    rz_next = uniform(sD[3] - alpha * (sD[4] + vs * sD[5]));
    sumr = uniform(sumr - alpha * (sD[5] + ncells * vs));
    beta = uniform(-(rz_next + vs * sumr) / pz);             // ... of iteration k + 1 (see above)
    // ---- U(k): z' again, x += alpha p, r -= alpha (z' + vs) on my cells and on the ring
    lU[0] = 0; lU[1] = 0;
    int cnt_wave = 0;                                        // #{|r_{k+1}| >= accuracy} of the whole wave, counted on the scalar unit
    if (has[0]) {
      issue_halos(Rz);                                       // (requested later in U, the stall moves into D: measured)
      if constexpr (kAhead < NT && !kKeepZ) ring_issue(NT - 1 - kAhead);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int t = NT - 1 - tt;                          // U walks the rows downwards
        const int q = t / R, jj = t - q * R;
        if constexpr (!kKeepZ) { if (tt + 1 < NT && tt + 1 > kAhead) ring_issue(t - 1); }     // (row kAhead's values were requested in front of the loop)
        T* xl = xs + (size_t)(wave * NQ + q) * R * 64 * V + lane * V + jj * 64 * V;
        Vec<T, V> xv = ldv<T, V>(xl);                        // x += alpha p (:303); the LDS latency hides under the stencil
        Vec<T, V> z;
        if constexpr (kKeepZ) z = zk[t];
        else z = (tt < kAhead) ? zs[tt < kAhead ? tt : 0] : zrow(t);
#pragma unroll
        for (int e = 0; e < V; ++e) xv.v[e] = fma(alpha, pp[q][jj].v[e], xv.v[e]);
        stv<T, V>(xl, xv);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          T vsc = vs;
          if constexpr (RAGGED) vsc = (col_ok[q] && j0[q] + jj < a.ny_true) ? vs : (T)0;
          const T rn = fma(-alpha, z.v[e] + vsc, rr[q][jj].v[e]);
          rr[q][jj].v[e] = rn;
          lU[0] += rn;
          if constexpr (!kCountLate) {
            // (one compare per cell; ballot + popcount + add run on the scalar unit - the loop is bound by VALU issue.  A NaN counts.)
            // (inline asm: left to the compiler the sixteen rows' masks are parked in VGPR lanes - v_writelane / v_readlane pairs,
            // VALU slots - and counted after the loop)
            const unsigned long long over = __ballot(!(absval(rn) < accuracy));
            int ones;
            asm volatile("s_bcnt1_i32_b64 %1, %2\n\ts_add_i32 %0, %0, %1" : "+s"(cnt_wave), "=&s"(ones) : "s"(over) : "scc");
          }
        }
        PISO_SB_B1;
        if constexpr (D < NT) {
          if (t - D >= 0) issue_coef(t - D);                 // downwards; rows D-1 .. 0 stay in registers for the next D pass
        }
        PISO_SB_B2;
      }
      if constexpr (kCountLate) {
        // #{|r_{k+1}| >= accuracy} is looked at by the stopping test of iteration k + 1 only if (k + 1) % 5 == 0 (:312-335): counted in a
        // pass of its own over the registers, in those iterations only (one wave-uniform branch; 32 fp64 compares less in 4 of 5 iterations)
        if ((k + 1) % 5 == 0) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int q = t / R, jj = t - q * R;
#pragma unroll
            for (int e = 0; e < V; ++e) {
              const unsigned long long over = __ballot(!(absval(rr[q][jj].v[e]) < accuracy));
              int ones;
              asm volatile("s_bcnt1_i32_b64 %1, %2\n\ts_add_i32 %0, %0, %1" : "+s"(cnt_wave), "=&s"(ones) : "s"(over) : "scc");
            }
          }
        }
      }
      // the ring: the same update with the neighbours' z' (beyond a wall there is no cell: the copies stay 0)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        T vsb = nbit(q, 1u) ? vs : (T)0, vsa = nbit(q, 2u) ? vs : (T)0;
        if constexpr (RAGGED) {
          if (!(col_ok[q] && j0[q] - 1 < a.ny_true)) vsb = 0;
          if (!(col_ok[q] && j0[q] + R < a.ny_true)) vsa = 0;
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
          rhb[q].v[e] = fma(-alpha, hbZ[q].v[e] + vsb, rhb[q].v[e]);
          rha[q].v[e] = fma(-alpha, haZ[q].v[e] + vsa, rha[q].v[e]);
        }
        int side, er_unused, lane_now = lane;
        if constexpr (SLAB) asm volatile("" : "+v"(lane_now));   // (slab variant: recomputed here instead of a register held across the loop)
        ring_lane(lane_now, R, side, er_unused);
        T vse = (side == 0) ? (nbit(q, 4u) ? vs : (T)0) : ((side == 1) ? (nbit(q, 8u) ? vs : (T)0) : (T)0);
        if constexpr (RAGGED) {
          const bool ok = (side == 0 ? lcol_ok[q] : rcol_ok[q]) && j0[q] + er_unused < a.ny_true;
          if (!ok) vse = 0;
        }
        if constexpr (kParkRing) {
          const unsigned rb = ring_a + (unsigned)(q * kRingBytes) + (unsigned)lane * (unsigned)sizeof(T) + kRingR;
          *(lds_T*)(unsigned long)rb = fma(-alpha, eZ[q] + vse, (T) * (lds_T*)(unsigned long)rb);
        } else {
        eR[q] = fma(-alpha, eZ[q] + vse, eR[q]);
        }
      }
      lU[1] = (lane == 0) ? (T)cnt_wave : (T)0;              // (the exchange adds the lanes of a wave)
    }
    tick(2);
  }
  if (kPersistDiag && c.timing && threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 5; ++q) c.timing[q * nslots + slot] += tacc[q];            // (XCD-local mode: the group's ranks, not the launch's blocks)
#pragma unroll
    for (int q = 0; q < 4; ++q) c.timing[(5 + q) * nslots + slot] += tsub[q];
    c.timing[9 * nslots + slot] = (unsigned long long)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7);   // the XCD I ran on
    c.timing[10 * nslots + slot] = (unsigned long long)wg;                                                              // the band of regions I own
  }
  // ---- the last U's sum r and count are only known per workgroup: one more exchange (once per segment)
  T tOut[3] = {rz_next, sumr, cnt_last};
  if (!first && healthy && !st.done) {
    T sX[kX1Values] = {0, 0, 0, 0, 0, 0, lU[0], lU[1]};
    ++epoch;
    if constexpr (kHier) healthy = grid_exchange8_hier<T, kPollDelay2NoAhead, SLAB>(c, sX, epoch, smem, hx, hier_s + 2, NoPrefetch(), nullptr, sl_off, smem + 2 * kX1Sm);     // (nothing to compute ahead)
    else if constexpr (kLocalAll) healthy = grid_exchange8_local<T>(c, sX, epoch, smem, slot, nslots, hier_s + 2);
    else healthy = grid_exchange8<T, LOCAL>(c, sX, epoch, smem, slot, nslots);
    tOut[1] = sX[6]; tOut[2] = sX[7];
  }

  // ---- back to the global-memory state of the two-kernel path (iteration k reads its direction from p[k & 1])
  // (the pointers of the exit are read again from the kernarg segment: held in SGPRs across the loop they are spilled, and the
  // reloads of OTHER spilled values land in the row loops)
  struct { T *r, *x, *pk, *partsB, *scal; const T* gB; CgState* state; int nB; } ax;       // (pk: p[k & 1])
  struct { int* err; } cx;
  {
    typedef CgArgs<T> A;
    constexpr unsigned ao = (unsigned)offsetof(KArgs, a), co = (unsigned)offsetof(KArgs, c);
    ax.r = karg<T*>(ao + (unsigned)offsetof(A, r)); ax.x = karg<T*>(ao + (unsigned)offsetof(A, x));
    ax.pk = karg<T*>(ao + (unsigned)offsetof(A, p) + 8u * (unsigned)(k & 1));
    ax.partsB = karg<T*>(ao + (unsigned)offsetof(A, partsB)); ax.scal = karg<T*>(ao + (unsigned)offsetof(A, scal));
    ax.gB = karg<const T*>(ao + (unsigned)offsetof(A, gB)); ax.state = karg<CgState*>(ao + (unsigned)offsetof(A, state));
    ax.nB = karg<int>(ao + (unsigned)offsetof(A, nB));
    cx.err = karg<int*>(co + (unsigned)offsetof(PersistCtl, err));
  }
  {
    const rsrc_t Rr = make_rsrc(ax.r, nbytesT), Rx = make_rsrc(ax.x, nbytesT), Rp = make_rsrc(ax.pk, nbytesT);
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
        if constexpr (SLAB) {                               // the two-kernel slab path continues from the halo rows of r and p
          const rsrc_t RrH = make_rsrc(ax.r - nx, nbytesH), RpH = make_rsrc(ax.pk - nx, nbytesH);
          T* hs = halo_s + (kParkHalos ? (size_t)((wave * NQ + q) * 2) * 64 * V + lane * V : 0);
          if constexpr (kParkHalos) { pnb[q] = ldv<T, V>(hs); pna[q] = ldv<T, V>(hs + 64 * V); }
          if (ef[q] & 1u) { bst<T, V>(RrH, vT[q], 0u, rhb[q]); bst<T, V>(RpH, vT[q], 0u, pnb[q]); }
          if (ef[q] & 2u) { bst<T, V>(RrH, vT[q], (unsigned)(ny + 1) * rowT, rha[q]); bst<T, V>(RpH, vT[q], (unsigned)(ny + 1) * rowT, pna[q]); }
        }
      }
  }
  if (wg == 0) {                                           // (= blockIdx.x 0, or the first arrival of the XCD-local group)
    // the next launch (cg_k1 with do_check, or another segment) finds the last K2-totals in record 0 of partsB
    for (int b = threadIdx.x; b < ax.nB; b += kPersistThreads) {
#pragma unroll
      for (int q = 0; q < 3; ++q) ax.partsB[q * kMaxPartials + b] = (b == 0) ? tOut[q] : (T)0;
    }
    if constexpr (SLAB) {
      if (threadIdx.x == 0) { T* g = const_cast<T*>(ax.gB); g[0] = tOut[0]; g[1] = tOut[1]; g[2] = tOut[2]; }   // (already summed over all GPUs)
    }
    if (threadIdx.x == 0) {
      ax.scal[SC_PZ] = pz; ax.scal[SC_VS] = vs; ax.scal[SC_ALPHA] = alpha;
      ax.state[0] = st; ax.state[1] = st;
      if (!healthy) *cx.err = 1;
    }
  }
}

}  // namespace piso
