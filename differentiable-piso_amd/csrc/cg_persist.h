// What the persistent CG kernels share (cg_persist1.h: the segment kernel; cg_tiny.h: one workgroup; cg_slab.hip: the slab
// variant): launch shape, the control block, wave-level helpers on the DPP network, buffer-resource loads / stores with a cache
// policy, lane shifts.  (Rounds 1-2 also kept a first persistent kernel here - two grid exchanges per iteration with the reference's
// recurrences, z' never stored; `cg_persist1` replaced it for fp64 in round 2 and for fp32 in round 3, DESIGN.md 3.1 has its
// measurements.)
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
  int waves;            // waves of a workgroup that own regions: 8, or 4 (one per SIMD: nobody waits for a SIMD's other wave; small grids)
};
#ifdef PISO_PERSIST_DIAG
constexpr bool kPersistDiag = true;     // per-phase clocks of wave 0 (PISO_CG_PERSIST_TIMING=1); costs a few registers
#else
constexpr bool kPersistDiag = false;
#endif
constexpr int kPersistMaxGrid = 256;   // workgroups (one per CU); the exchange keeps kPersistMaxGrid / 64 records per lane in registers
// workspace of the exchanges, in 4-byte words: level-1 records [2 parities][kPersistMaxGrid] x 128 B, then the eight XCD records of
// the tree's second level [2][8] x 128 B (cg_persist1.h: grid_exchange8_hier), then the control words (XCD arrival counters at 0,
// the workgroups' XCDs at kPersistXcdTable, the error flag 16 words from the end).  Records and arrival counters are zeroed before every launch (kPersistZeroBytes).
constexpr size_t kPersistRecWords = (size_t)2 * kPersistMaxGrid * 32 + (size_t)2 * 8 * 32;
constexpr int kPersistXcdTable = 16;    // word offset (from PersistCtl::xcd) of the table "XCD of workgroup b", kPersistMaxGrid entries (hier_enter)
constexpr size_t kPersistWsWordsAll = kPersistRecWords + 64 + kPersistMaxGrid;
constexpr size_t kPersistZeroBytes = kPersistRecWords * 4 + 16 * sizeof(int);
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

}  // namespace piso
