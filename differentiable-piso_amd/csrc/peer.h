// Peer-mapped mailboxes: the transport of the slab-decomposed solvers inside one node (new design, the
// reference is single-GPU).
//
// Every rank owns ONE uncached, fine-grained device allocation (its mailbox), exported with hipIpcGetMemHandle and mapped by
// all other ranks (xGMI peer access).  Whatever crosses GPUs is WRITTEN by the producer straight into the consumer's mailbox
// (one hop, no library call between kernels) and READ locally by the consumer:
//   * reductions travel as tagged 8-byte words {32 payload bits | 32-bit sequence tag}: single-copy atomic, so a word that
//     carries the expected tag is complete by itself - no flag, no fence (the protocol of the persistent CG kernel's grid
//     exchange, cg_persist.h, at system scope);
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
constexpr unsigned kPeerSpinLimit = 1u << 24;  // polling passes before a wait gives up (seconds; a dead peer must not hang the node)

// byte offsets inside a mailbox; row_cap = capacity of a halo row in elements of 8 bytes
struct PeerLayout {
  static constexpr size_t kRecBytes = kPeerRecWords * 8;
  static constexpr size_t ar_rec(int parity, int src) { return ((size_t)parity * kMaxRanks + src) * kRecBytes; }             // host-level all-reduce
  static constexpr size_t x_rec(int parity, int src) { return (size_t)2 * kMaxRanks * kRecBytes + ar_rec(parity, src); }     // persistent kernel's GPU records
  static constexpr size_t ex_flag(int parity, int side) { return (size_t)4 * kMaxRanks * kRecBytes + ((size_t)parity * 2 + side) * 128; }
  static constexpr size_t kRows = (size_t)4 * kMaxRanks * kRecBytes + 4 * 128;
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

}  // namespace piso
