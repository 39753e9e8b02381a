// The communicator of the slab-decomposed solvers (cg_slab.hip, bicgstab.hip): RCCL or peer-mapped mailboxes (peer.h).
#pragma once
#include <rccl/rccl.h>

#include "peer.h"

namespace piso {

enum { TRANSPORT_RCCL = 1, TRANSPORT_PEER = 2 };
struct PisoComm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  int transport = TRANSPORT_RCCL;
  // peer transport
  char* mbox[kMaxRanks] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool connected = false;
  size_t row_cap = 0, mbox_bytes = 0;
  // how the mailboxes were mapped: 0 = hipIpc handles (piso_comm_peer_create), 1 = virtual-memory allocations shared as POSIX file
  // descriptors (piso_comm_peer_create_fd: hipMemCreate / hipMemExportToShareableHandle / hipMemImportFromShareableHandle)
  int vmm = 0;
  hipMemGenericAllocationHandle_t vmm_handle[kMaxRanks] = {};
  size_t vmm_bytes = 0;
  unsigned seq_pp = 0;                // ping-pong tags (piso_comm_pingpong)
  unsigned seq_ar = 0, seq_ex = 0;    // sequence numbers of the host-level collectives (advance identically on every rank)
  unsigned launches = 0;              // persistent slab launches so far: the high half of their exchange tags
  int* err = nullptr;                 // device flag: a wait on a peer gave up
  int persist_fallbacks = 0;          // solves restarted on the two-kernel iteration after a persistent segment failed
  long long persist_iterations = 0;   // CG iterations executed inside persistent slab segments
  long long verify_runs = 0;          // slab solves checked against the true residual after persistent segments ...
  int verify_failures = 0;            // ... and found wanting on some rank: restarted on the two-kernel iteration
};

// RCCL transport of what the peer kernels do through the mailboxes (defined in cg_slab.hip, where the RCCL entry points live):
//   * the four halo messages of a globally indexed vector {to upper, to lower, from lower, from upper}: grouped send / recv of the
//     segments, straight from / into the vector (no staging).  Sends and receives between one pair of ranks are matched in issue
//     order, and with one or two ranks the lower and the upper neighbour are the same peer: every rank issues "to upper" before
//     "to lower" and "from lower" before "from upper".  dtype: 0 float, 1 double, 2 int32;
//   * in-place sum of `count` doubles / ints over the ranks.
int comm_rccl_exchange_segments(PisoComm* pc, void* vec, int dtype, const HaloMsg* m4, hipStream_t stream);
int comm_rccl_allreduce_f64(PisoComm* pc, double* buf, int count, hipStream_t stream);
int comm_rccl_allreduce_i32(PisoComm* pc, int* buf, int count, hipStream_t stream);

inline PeerView make_view(const PisoComm* pc, bool periodic_y) {
  PeerView v;
  for (int r = 0; r < kMaxRanks; ++r) v.mbox[r] = pc->mbox[r];
  v.rank = pc->rank; v.world = pc->world; v.row_cap = pc->row_cap;
  v.lower = (pc->rank > 0) ? pc->rank - 1 : (periodic_y ? pc->world - 1 : -1);
  v.upper = (pc->rank < pc->world - 1) ? pc->rank + 1 : (periodic_y ? 0 : -1);
  return v;
}


}  // namespace piso
