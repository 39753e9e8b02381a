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
  unsigned seq_ar = 0, seq_ex = 0;    // sequence numbers of the host-level collectives (advance identically on every rank)
  unsigned launches = 0;              // persistent slab launches so far: the high half of their exchange tags
  int* err = nullptr;                 // device flag: a wait on a peer gave up
  int persist_fallbacks = 0;          // solves restarted on the two-kernel iteration after a persistent segment failed
  long long persist_iterations = 0;   // CG iterations executed inside persistent slab segments
  long long verify_runs = 0;          // slab solves checked against the true residual after persistent segments ...
  int verify_failures = 0;            // ... and found wanting on some rank: restarted on the two-kernel iteration
};

inline PeerView make_view(const PisoComm* pc, bool periodic_y) {
  PeerView v;
  for (int r = 0; r < kMaxRanks; ++r) v.mbox[r] = pc->mbox[r];
  v.rank = pc->rank; v.world = pc->world; v.row_cap = pc->row_cap;
  v.lower = (pc->rank > 0) ? pc->rank - 1 : (periodic_y ? pc->world - 1 : -1);
  v.upper = (pc->rank < pc->world - 1) ? pc->rank + 1 : (periodic_y ? 0 : -1);
  return v;
}


}  // namespace piso
