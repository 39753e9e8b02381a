// Peer-mapped mailboxes: the transport of the slab-decomposed solvers inside one node (new design, the
// reference is single-GPU).
//
// Every rank owns ONE uncached, fine-grained device allocation (its mailbox), exported with hipIpcGetMemHandle and mapped by
// all other ranks (xGMI peer access).  Whatever crosses GPUs is WRITTEN by the producer straight into the consumer's mailbox
// (one hop, no library call between kernels) and READ locally by the consumer:
//   * reductions travel as tagged 8-byte words {32 payload bits | 32-bit sequence tag}: single-copy atomic, so a word that
//     carries the expected tag is complete by itself - no flag, no fence (the protocol of the persistent CG kernel's grid
//     exchange, cg_persist1.h, at system scope);
//   * halo rows (64-bit payloads) are followed by a system-scope RELEASE store of their sequence number; the consumer
//     ACQUIREs it before reading the row;
//   * everything alternates between two slots by sequence parity; a rank can run at most one collective ahead of its slowest
//     peer (it needs that peer's contribution to finish the current one), so two slots suffice.
// All accesses to a mailbox are system-scope atomics (sc0 sc1): they bypass the non-coherent cache levels on both sides.
#pragma once
#include "piso_common.h"

namespace piso {

constexpr int kMaxRanks = 8;                   // GPUs of one node
constexpr int kPeerRecWords = 16;              // 8-byte words per record: 8 sums x {low half | tag, high half | tag}
constexpr int kPeerXcds = 8;                   // XCDs of one GPU (= kXcds of piso_common.h)
constexpr unsigned kPeerSpinLimit = 1u << 24;  // polling passes before a wait gives up (seconds; a dead peer must not hang the node)

// byte offsets inside a mailbox; row_cap = capacity of a halo row in elements of 8 bytes
struct PeerLayout {
  static constexpr size_t kRecBytes = kPeerRecWords * 8;
  static constexpr size_t ar_rec(int parity, int src) { return ((size_t)parity * kMaxRanks + src) * kRecBytes; }             // host-level all-reduce
  static constexpr size_t ex_flag(int parity, int side) { return (size_t)2 * kMaxRanks * kRecBytes + ((size_t)parity * 2 + side) * 128; }
  // persistent kernel: the records of the XCD leaders of EVERY rank (round 5: the leaders store straight into the peers' mailboxes -
  // no "GPU total" level between the chip's exchange and the node's), [2 parities][kMaxRanks][kPeerXcds] records of 128 bytes
  // two words for the ping-pong of piso_comm_pingpong: [0] where the initiator receives, [1] where the responder receives
  static constexpr size_t pp_flag(int which) { return (size_t)2 * kMaxRanks * kRecBytes + (4 + (size_t)which) * 128; }
  static constexpr size_t kXcdRecs = (size_t)2 * kMaxRanks * kRecBytes + 6 * 128;
  static constexpr size_t xcd_rec(int parity, int src, int xcd) { return kXcdRecs + (((size_t)parity * kMaxRanks + src) * kPeerXcds + xcd) * kRecBytes; }
  static constexpr size_t kXcdRecBytes = (size_t)2 * kMaxRanks * kPeerXcds * kRecBytes;
  static constexpr size_t kRows = kXcdRecs + kXcdRecBytes;
  // side 0: the row BELOW my slab (written by my lower neighbour), side 1: the row ABOVE it (written by my upper neighbour)
  static constexpr size_t ex_row(int parity, int side, size_t row_cap) { return kRows + ((size_t)parity * 2 + side) * row_cap * 8; }   // host-level halo exchange
  static constexpr size_t z_row(int parity, int side, size_t row_cap) { return kRows + (4 + (size_t)parity * 2 + side) * row_cap * 8; }   // persistent kernel: z' halo rows
  static size_t bytes(size_t row_cap) { return align_up(kRows + 8 * row_cap * 8, 4096); }
};

// what a kernel needs to talk to the other ranks (passed by value)
struct PeerView {
  char* mbox[kMaxRanks];     // mbox[r]: rank r's mailbox in MY address space (mbox[rank] is my own)
  int rank, world;
  int lower, upper;          // ring neighbours along y (-1: none, i.e. a wall / open boundary below or above the global grid)
  size_t row_cap;
};

typedef unsigned long long peer_u64;
__device__ __forceinline__ void peer_store(peer_u64* p, peer_u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ peer_u64 peer_load(const peer_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// one double as two tagged words
__device__ __forceinline__ peer_u64 peer_tagged(double v, int half, unsigned tag) {
  const peer_u64 bits = (peer_u64)__double_as_longlong(v);
  return half ? ((bits & 0xffffffff00000000ull) | tag) : (((bits & 0xffffffffull) << 32) | tag);
}
__device__ __forceinline__ double peer_untag(peer_u64 lo_word, peer_u64 hi_word) {
  return __longlong_as_double((long long)((lo_word >> 32) | (hi_word & 0xffffffff00000000ull)));
}

// One wave sums up to 8 values over the ranks through the all-reduce records of the mailboxes.  Lane l < nwords carries half
// l & 1 of value l >> 1 (`v_lane` = that value) as a tagged word into words [word0, word0 + nwords) of every rank's record
// (parity seq & 1, source = my rank), then polls the `world` records of its own mailbox and adds them in rank order: bitwise the
// same sums on every rank.  Returns the sum of value l >> 1 in the EVEN lanes below nwords.
__device__ __forceinline__ double peer_wave_sum(const PeerView& pv, double v_lane, int nwords, int word0, unsigned seq, bool* good) {
  const int lane = threadIdx.x & 63;
  const bool active = lane < nwords;
  const peer_u64 word = peer_tagged(v_lane, lane & 1, seq);
  if (active)
    for (int p = 0; p < pv.world; ++p)
      peer_store(reinterpret_cast<peer_u64*>(pv.mbox[p] + PeerLayout::ar_rec(seq & 1, pv.rank)) + word0 + lane, word);
  double acc = 0;
  for (int r = 0; r < pv.world; ++r) {
    peer_u64 w = 0;
    unsigned spins = 0;
    while (true) {
      if (active) w = peer_load(reinterpret_cast<const peer_u64*>(pv.mbox[pv.rank] + PeerLayout::ar_rec(seq & 1, r)) + word0 + lane);
      if (__all(!active || (unsigned)(w & 0xffffffffull) == seq)) break;
      if (++spins > kPeerSpinLimit) { *good = false; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    const peer_u64 wo = __shfl_down(w, 1, 64);
    acc += peer_untag(w, wo);
  }
  return acc;
}

// The error flag of a solve, agreed over the ranks: a wait that gave up on ONE rank (a slow peer, not a dead one) must fail the call
// on EVERY rank - a rank that alone returned an error would leave its peers waiting in the next collective, on the device or in
// torch.distributed.  One wave sums the flags (a dead transport makes this wait give up on every rank, which is an error too).
template <int kUnused = 0>
__global__ void peer_agree_on_error(PeerView pv, int* err, unsigned seq) {
  const int lane = threadIdx.x & 63;
  bool good = true;
  const double acc = peer_wave_sum(pv, lane < 2 ? (double)*err : 0.0, 2, 0, seq, &good);
  if (lane == 0 && (!good || acc != 0.0)) *err = 1;
}

// Mailbox ping-pong between ranks a (initiator) and b (responder): `iters` round trips of one tagged 8-byte word, written straight into
// the other rank's mailbox and polled locally - the hop every exchange of the slab solvers pays (bench.py: sharded.hop_us_matrix).
// a == b: the rank talks to its own mailbox (the cost of the uncached access path without a link).  One thread.
template <int kUnused = 0>
__global__ void peer_pingpong(PeerView pv, int a, int b, int iters, unsigned seq0, int* err) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const bool init = pv.rank == a;
  const int other = init ? b : a;
  const peer_u64* mine = reinterpret_cast<const peer_u64*>(pv.mbox[pv.rank] + PeerLayout::pp_flag(init ? 0 : 1));
  peer_u64* theirs = reinterpret_cast<peer_u64*>(pv.mbox[other] + PeerLayout::pp_flag(init ? 1 : 0));
  if (a == b) theirs = reinterpret_cast<peer_u64*>(pv.mbox[pv.rank] + PeerLayout::pp_flag(0));     // (ring of one: my own word, there and back)
  for (int i = 0; i < iters; ++i) {
    const peer_u64 tag = (peer_u64)(seq0 + (unsigned)i);
    if (init) peer_store(theirs, tag);
    unsigned spins = 0;
    while (peer_load(mine) != tag) {
      if (++spins > kPeerSpinLimit) { *err = 1; return; }
    }
    if (!init) peer_store(theirs, tag);
  }
}

// Segments of a globally indexed vector that cross a slab edge (element offsets into the vector; at most 3 segments per message)
struct HaloMsg {
  int count;
  int off[3], len[3];
};
// Block 0 pushes `to_upper` into the upper neighbour's mailbox (side 0: what lies below ITS slab), block 1 `to_lower` into the
// lower neighbour's (side 1); a system-scope release store of `seq` follows the data.  Then block 0 waits for the message from
// below and stores it at `from_lower`'s offsets of my copy of the vector, block 1 the same for the message from above.  One
// element travels as one 8-byte word.  Every rank pushes before it waits.
template <typename T>
__global__ __launch_bounds__(256) void peer_exchange_segments(PeerView pv, T* vec, HaloMsg to_upper, HaloMsg to_lower, HaloMsg from_lower,
                                                              HaloMsg from_upper, unsigned seq, int* err) {
  const int side = blockIdx.x;                             // out: 0 to the upper neighbour, 1 to the lower; in: 0 from below, 1 from above
  const int dst = side == 0 ? pv.upper : pv.lower;
  const int par = seq & 1;
  if (dst >= 0) {
    const HaloMsg& m = side == 0 ? to_upper : to_lower;
    peer_u64* row = reinterpret_cast<peer_u64*>(pv.mbox[dst] + PeerLayout::ex_row(par, side, pv.row_cap));
    int base = 0;
    for (int q = 0; q < m.count; ++q) {
      for (int i = threadIdx.x; i < m.len[q]; i += 256) peer_store(row + base + i, (peer_u64)__double_as_longlong((double)vec[m.off[q] + i]));
      base += m.len[q];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_store(reinterpret_cast<peer_u64*>(pv.mbox[dst] + PeerLayout::ex_flag(par, side)), (peer_u64)seq, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const int from = side == 0 ? pv.lower : pv.upper;
  if (from < 0) return;
  __shared__ int ok_s;
  if (threadIdx.x == 0) {
    const peer_u64* flag = reinterpret_cast<const peer_u64*>(pv.mbox[pv.rank] + PeerLayout::ex_flag(par, side));
    unsigned spins = 0;
    int ok = 1;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != (peer_u64)seq) {
      if (++spins > kPeerSpinLimit) { ok = 0; *err = 1; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    ok_s = ok;
  }
  __syncthreads();
  if (!ok_s) return;
  const HaloMsg& m = side == 0 ? from_lower : from_upper;
  const peer_u64* row = reinterpret_cast<const peer_u64*>(pv.mbox[pv.rank] + PeerLayout::ex_row(par, side, pv.row_cap));
  int base = 0;
  for (int q = 0; q < m.count; ++q) {
    for (int i = threadIdx.x; i < m.len[q]; i += 256) vec[m.off[q] + i] = (T)__longlong_as_double((long long)peer_load(row + base + i));
    base += m.len[q];
  }
}

}  // namespace piso
