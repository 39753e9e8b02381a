// Slab-decomposed pressure CG: the grid is cut into 1-D slabs along y (contiguous row blocks), one rank per GPU.
// SURVEY.md 8(e).  The reference has no multi-GPU path; this is new design.
//
// Per CG iteration and rank:  K1  ->  3-double all-reduce  ->  K2  ->  3-double all-reduce + one-row halo exchange of r.
//   * the kernels are the single-GPU ones (cg_kernels.h) in "halo mode": r, p[2] and x carry one halo row below and above
//     the owned rows; K1 recomputes the new direction on the halo rows from (r_halo, p_halo) and keeps it, so only r has to
//     be exchanged (one row = nx * 8 bytes per neighbour and iteration);
//   * per-block partial sums are collapsed to 3 scalars on the device, all-reduced over RCCL (xGMI), and read back by the
//     next kernel's prologue -- no host round trip; every rank evaluates the same stopping test on the same numbers;
//   * the rank-1 shift uses the global sum |diag| and the global cell count.
// Communication goes through a tiny interface with three implementations:
//   * PEER (default inside a node): every rank owns a peer-mapped mailbox (peer.h); reductions and halo rows are written by
//     small kernels straight into the consumers' mailboxes - no library call, no host round trip.  With this transport the
//     NORMAL iterations run inside the persistent kernel cg_persist1<..., SLAB = true>: r, p, x of the slab stay on chip, the
//     perimeter rows at the slab edges and the per-GPU totals cross xGMI from inside the kernel (one extra hop per iteration);
//     resets, the first iteration and shapes the kernel cannot tile use the two-kernel iteration below;
//   * RCCL (librccl is dlopen'ed on first use, so the library has no link-time dependency on it): two-kernel iteration only;
//   * an in-process LOOPBACK that runs G virtual ranks on one device in lock-step -- the test harness for the multi-rank index
//     logic on a single-GPU box (tests/test_gpu_slab.py).
// The peer transport is exercised on a single-GPU box as well: several PROCESSES share the device, export their mailboxes
// through hipIpc handles and run their kernels concurrently (tests/test_gpu_multiproc.py).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <vector>

#include "cg_kernels.h"
#include "cg_persist1.h"
#include "options.h"
#include "peer.h"
#include "slab_comm.h"

namespace piso {

// ------------------------------------------------------------------------------------------------ RCCL (lazy)
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;

static int load_rccl() {
  if (g_rccl.handle) return PISO_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
  if (!h) { set_error_msg("cannot dlopen librccl"); return PISO_ERR_HIP; }
#define PISO_SYM(field, sym) \
  *reinterpret_cast<void**>(&g_rccl.field) = dlsym(h, sym); \
  if (!g_rccl.field) { set_error_msg("librccl lacks " sym); return PISO_ERR_HIP; }
  PISO_SYM(GetUniqueId, "ncclGetUniqueId") PISO_SYM(CommInitRank, "ncclCommInitRank") PISO_SYM(CommDestroy, "ncclCommDestroy")
  PISO_SYM(AllReduce, "ncclAllReduce") PISO_SYM(AllGather, "ncclAllGather") PISO_SYM(Send, "ncclSend") PISO_SYM(Recv, "ncclRecv")
  PISO_SYM(GroupStart, "ncclGroupStart") PISO_SYM(GroupEnd, "ncclGroupEnd") PISO_SYM(GetErrorString, "ncclGetErrorString")
#undef PISO_SYM
  g_rccl.handle = h;
  return PISO_OK;
}

#define PISO_NCCL_CHECK(expr)                                                   \
  do {                                                                          \
    ncclResult_t _r = (expr);                                                   \
    if (_r != ncclSuccess) {                                                    \
      char buf[256];                                                            \
      snprintf(buf, sizeof(buf), "%s: %s", #expr, g_rccl.GetErrorString(_r));   \
      set_error_msg(buf);                                                       \
      return PISO_ERR_HIP;                                                      \
    }                                                                           \
  } while (0)

// ------------------------------------------------------------------------------------------------ per-rank context
template <typename T>
struct SlabRank {
  CgArgs<T> a;          // r, p[], x point at row 0 of buffers that own one halo row below (row -1) and above (row ny)
  T *rbase, *pbase[2], *xbase;
  T* g;                 // [12]: gA[0..2], pad, gB[4..6], pad, gS[8..10] (sum |diag|, #not-f32, #not-recon)
  T* oT; float* oF; T* cC;
  int* flags;
  const T* L;
  T* x_out;             // owned rows of the caller's output
  int rank;             // position in the slab ring
  unsigned* persist_ws; // exchange records + error flag of the persistent kernel
};

// collapse per-block partial records into `count` scalars (fixed order)
template <typename T>
__global__ __launch_bounds__(kBlock) void slab_collapse(const T* __restrict__ parts, int records, int count, T* __restrict__ out) {
  __shared__ T smem[16];
  for (int q = 0; q < count; ++q) {
    T v[1] = {0};
    for (int b = threadIdx.x; b < records; b += kBlock) v[0] += parts[q * kMaxPartials + b];
    block_sum<T, 1>(v, smem);
    if (threadIdx.x == 0) out[q] = v[0];
  }
}
template <typename T>
__global__ void slab_flags_to_sums(const int* flags, T* out) {
  if (threadIdx.x == 0) { out[1] = (T)flags[0]; out[2] = (T)flags[1]; out[3] = (T)flags[2]; }
}
// error flags of a persistent segment (its own exchanges, the waits of the host-level collectives) as a summable value
template <typename T>
__global__ void slab_err_to_sum(const int* seg_err, const int* comm_err, T* out) {
  if (threadIdx.x == 0) out[0] = (T)((*seg_err != 0 || *comm_err != 0) ? 1 : 0);
}
// verification of a slab solve (cg_verify_gap): did the gap exceed its bound on this rank?  (summable)
template <typename T>
__global__ void slab_gap_to_sum(const unsigned* out2, T* out, int force) {
  if (threadIdx.x == 0) {
    const float gap = __uint_as_float(out2[0]), scale = __uint_as_float(out2[1]);
    out[0] = (T)(((gap > 1e-5f * scale && gap > 1e-30f) || force) ? 1 : 0);
  }
}
constexpr size_t kSlabPersistWsWords = kPersistWsWordsAll;   // exchange records + control words (cg_persist.h)
// loopback all-reduce: bufs of the G virtual ranks live `stride` apart; sum in rank order, write to all
template <typename T>
__global__ void loop_allreduce(T* base, int G, size_t stride, int count) {
  const int q = threadIdx.x;
  if (q >= count) return;
  T s = 0;
  for (int r = 0; r < G; ++r) s += base[(size_t)r * stride + q];
  for (int r = 0; r < G; ++r) base[(size_t)r * stride + q] = s;
}
template <typename T>
__global__ void slab_copy_rows(const T* __restrict__ src, T* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// ---- peer transport: one wave sums `count` <= 8 values of every rank.  Lane l < 2 count carries half l & 1 of value l / 2 as a
// tagged word to every rank's mailbox (mine included), then polls the `world` records of its own mailbox and adds them in
// rank order: every rank obtains bitwise the same sums.
__global__ void peer_allreduce(PeerView pv, double* g, int count, unsigned seq, int* err) {
  const int lane = threadIdx.x;
  bool good = true;
  const double acc = peer_wave_sum(pv, lane < 2 * count ? g[lane >> 1] : 0.0, 2 * count, 0, seq, &good);
  if (lane < 2 * count && (lane & 1) == 0) g[lane >> 1] = acc;
  if (!good && lane == 0) *err = 1;
}

// ---- peer transport: halo rows.  Block 0 writes my top row into the upper neighbour's mailbox (the row BELOW its slab, side 0),
// block 1 my bottom row into the lower neighbour's (the row ABOVE its slab, side 1); a system-scope release store of the
// sequence number follows the data.  Then block 0 waits for the row below my slab, block 1 for the row above it, and copies it
// to the halo row.  Every rank pushes before it waits: no ordering between ranks is needed.
__global__ __launch_bounds__(kBlock) void peer_exchange_rows(PeerView pv, const double* bottom_row, const double* top_row,
                                                             double* halo_below, double* halo_above, int nx, unsigned seq, int* err) {
  const int side_out = blockIdx.x;                         // 0: to the upper neighbour, 1: to the lower neighbour
  const int dst = side_out == 0 ? pv.upper : pv.lower;
  const int par = seq & 1;
  if (dst >= 0) {
    const double* src = side_out == 0 ? top_row : bottom_row;
    peer_u64* row = reinterpret_cast<peer_u64*>(pv.mbox[dst] + PeerLayout::ex_row(par, side_out, pv.row_cap));
    for (int i = threadIdx.x; i < nx; i += kBlock) peer_store(row + i, (peer_u64)__double_as_longlong(src[i]));
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_store(reinterpret_cast<peer_u64*>(pv.mbox[dst] + PeerLayout::ex_flag(par, side_out)), (peer_u64)seq, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const int side_in = blockIdx.x;                          // 0: the row below my slab (from the lower neighbour), 1: the row above
  const int from = side_in == 0 ? pv.lower : pv.upper;
  if (from < 0) return;
  __shared__ int ok_s;
  if (threadIdx.x == 0) {
    const peer_u64* flag = reinterpret_cast<const peer_u64*>(pv.mbox[pv.rank] + PeerLayout::ex_flag(par, side_in));
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
  const peer_u64* row = reinterpret_cast<const peer_u64*>(pv.mbox[pv.rank] + PeerLayout::ex_row(par, side_in, pv.row_cap));
  double* halo = side_in == 0 ? halo_below : halo_above;
  for (int i = threadIdx.x; i < nx; i += kBlock) halo[i] = __longlong_as_double((long long)peer_load(row + i));
}

int comm_rccl_exchange_segments(PisoComm* pc, void* vec, int dtype, const HaloMsg* m, hipStream_t s) {
  if (!pc || pc->transport != TRANSPORT_RCCL || !pc->comm) { set_error_msg("comm_rccl_exchange_segments: not an RCCL communicator"); return PISO_ERR_INVALID_ARG; }
  const int rank = pc->rank, world = pc->world;
  const int lo = rank > 0 ? rank - 1 : world - 1, hi = rank < world - 1 ? rank + 1 : 0;          // always a ring (as the peer kernels)
  const ncclDataType_t dt = dtype == 0 ? ncclFloat : (dtype == 1 ? ncclDouble : ncclInt32);
  const size_t es = dtype == 1 ? 8 : 4;
  char* base = static_cast<char*>(vec);
  PISO_NCCL_CHECK(g_rccl.GroupStart());
  for (int q = 0; q < m[0].count; ++q) PISO_NCCL_CHECK(g_rccl.Send(base + (size_t)m[0].off[q] * es, (size_t)m[0].len[q], dt, hi, pc->comm, s));
  for (int q = 0; q < m[2].count; ++q) PISO_NCCL_CHECK(g_rccl.Recv(base + (size_t)m[2].off[q] * es, (size_t)m[2].len[q], dt, lo, pc->comm, s));
  for (int q = 0; q < m[1].count; ++q) PISO_NCCL_CHECK(g_rccl.Send(base + (size_t)m[1].off[q] * es, (size_t)m[1].len[q], dt, lo, pc->comm, s));
  for (int q = 0; q < m[3].count; ++q) PISO_NCCL_CHECK(g_rccl.Recv(base + (size_t)m[3].off[q] * es, (size_t)m[3].len[q], dt, hi, pc->comm, s));
  PISO_NCCL_CHECK(g_rccl.GroupEnd());
  return PISO_OK;
}
int comm_rccl_allreduce_f64(PisoComm* pc, double* buf, int count, hipStream_t s) {
  if (!pc || pc->transport != TRANSPORT_RCCL || !pc->comm) { set_error_msg("comm_rccl_allreduce: not an RCCL communicator"); return PISO_ERR_INVALID_ARG; }
  PISO_NCCL_CHECK(g_rccl.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, pc->comm, s));
  return PISO_OK;
}
int comm_rccl_allreduce_i32(PisoComm* pc, int* buf, int count, hipStream_t s) {
  if (!pc || pc->transport != TRANSPORT_RCCL || !pc->comm) { set_error_msg("comm_rccl_allreduce: not an RCCL communicator"); return PISO_ERR_INVALID_ARG; }
  PISO_NCCL_CHECK(g_rccl.AllReduce(buf, buf, (size_t)count, ncclInt32, ncclSum, pc->comm, s));
  return PISO_OK;
}

// ------------------------------------------------------------------------------------------------ communication
template <typename T>
struct Comm {
  int world;              // slabs in the ring
  bool periodic_y;
  PisoComm* rccl;         // NULL = loopback over the local ranks; else the communicator (RCCL or peer transport)
  size_t g_stride;        // loopback: distance between consecutive ranks' g buffers
  bool peer() const { return rccl && rccl->transport == TRANSPORT_PEER; }

  // sum `count` values at offset `off` of every rank's g buffer
  int allreduce(std::vector<SlabRank<T>>& R, int off, int count, hipStream_t s) {
    if (peer()) {
      if constexpr (sizeof(T) == 8) {
        if (world == 1) return PISO_OK;
        peer_allreduce<<<1, 64, 0, s>>>(make_view(rccl, periodic_y), reinterpret_cast<double*>(R[0].g) + off, count, ++rccl->seq_ar, rccl->err);
      } else {
        set_error_msg("peer transport: fp64 only"); return PISO_ERR_INVALID_ARG;
      }
    } else if (rccl) {
      if (world == 1) return PISO_OK;
      PISO_NCCL_CHECK(g_rccl.AllReduce(R[0].g + off, R[0].g + off, count, sizeof(T) == 8 ? ncclDouble : ncclFloat, ncclSum,
                                       rccl->comm, s));
    } else {
      loop_allreduce<T><<<1, 64, 0, s>>>(R[0].g + off, world, g_stride, count);
    }
    return PISO_OK;
  }
  // fill the halo rows (row -1 and row ny) of `which` (0: r, 1: x) from the neighbours' edge rows
  int exchange(std::vector<SlabRank<T>>& R, int which, hipStream_t s) {
    const int nx = R[0].a.nx;
    auto base0 = [&](SlabRank<T>& k) { return which == 0 ? k.a.r : k.a.x; };   // row 0
    if (peer()) {
      if constexpr (sizeof(T) == 8) {
        SlabRank<T>& me = R[0];
        double* row0 = reinterpret_cast<double*>(base0(me));
        const int ny = me.a.ny;
        if (nx > (int)rccl->row_cap) { set_error_msg("peer transport: row longer than the mailbox rows"); return PISO_ERR_INVALID_ARG; }
        peer_exchange_rows<<<2, kBlock, 0, s>>>(make_view(rccl, periodic_y), row0, row0 + (size_t)(ny - 1) * nx, row0 - nx,
                                                row0 + (size_t)ny * nx, nx, ++rccl->seq_ex, rccl->err);
      } else {
        set_error_msg("peer transport: fp64 only"); return PISO_ERR_INVALID_ARG;
      }
    } else if (rccl) {
      SlabRank<T>& me = R[0];
      const int ny = me.a.ny, rank = rccl->rank;
      const int lo = (rank > 0) ? rank - 1 : (periodic_y ? world - 1 : -1);
      const int hi = (rank < world - 1) ? rank + 1 : (periodic_y ? 0 : -1);
      T* row0 = base0(me);
      const ncclDataType_t dt = sizeof(T) == 8 ? ncclDouble : ncclFloat;
      // Sends and receives between one pair of ranks are matched in issue order, and with 1 or 2 ranks the lower and the
      // upper neighbour are the same peer: every rank issues the UPWARD transfer first, then the DOWNWARD one.
      PISO_NCCL_CHECK(g_rccl.GroupStart());
      if (hi >= 0) PISO_NCCL_CHECK(g_rccl.Send(row0 + (size_t)(ny - 1) * nx, nx, dt, hi, rccl->comm, s));   // my top row -> upper's lower halo
      if (lo >= 0) PISO_NCCL_CHECK(g_rccl.Recv(row0 - nx, nx, dt, lo, rccl->comm, s));                      // lower's top row -> my lower halo
      if (lo >= 0) PISO_NCCL_CHECK(g_rccl.Send(row0, nx, dt, lo, rccl->comm, s));                           // my bottom row -> lower's upper halo
      if (hi >= 0) PISO_NCCL_CHECK(g_rccl.Recv(row0 + (size_t)ny * nx, nx, dt, hi, rccl->comm, s));         // upper's bottom row -> my upper halo
      PISO_NCCL_CHECK(g_rccl.GroupEnd());
    } else {
      for (int r = 0; r < world; ++r) {
        SlabRank<T>& me = R[r];
        const int ny = me.a.ny;
        const int lo = (r > 0) ? r - 1 : (periodic_y ? world - 1 : -1);
        const int hi = (r < world - 1) ? r + 1 : (periodic_y ? 0 : -1);
        T* row0 = base0(me);
        if (lo >= 0) slab_copy_rows<T><<<4, 256, 0, s>>>(base0(R[lo]) + (size_t)(R[lo].a.ny - 1) * nx, row0 - nx, nx);
        if (hi >= 0) slab_copy_rows<T><<<4, 256, 0, s>>>(base0(R[hi]), row0 + (size_t)ny * nx, nx);
      }
    }
    return PISO_OK;
  }
};

// ------------------------------------------------------------------------------------------------ driver
template <typename T>
static size_t slab_rank_bytes(int nx, int nyl) {
  const size_t n = (size_t)nx * nyl, nh = (size_t)nx * (nyl + 2);
  size_t b = 0;
  b += align_up(n * sizeof(T), 256) * (1 + 4 + 1);          // cC, oT(4), z
  b += align_up(4 * n * sizeof(float), 256);                 // oF
  b += align_up(nh * sizeof(T), 256) * 4;                    // r, p0, p1, x with halos
  b += align_up(n * sizeof(T), 256) * 2;                     // z' perimeter buffers of the persistent kernel
  b += 3 * align_up(3 * kMaxPartials * sizeof(T), 256);
  b += 4 * 256 + align_up(16 * sizeof(T), 256);
  b += align_up(kSlabPersistWsWords * sizeof(unsigned), 256);
  return b + 4096;
}

struct SlabPinned { CgState st; int err; int pad[3]; double errsum; };
static thread_local SlabPinned* tl_slab_pinned = nullptr;

template <typename T, typename CT, bool RECON, bool SYMV>
static void launch_slab_segment(int R, int grid, const CgArgs<T>& a, const PersistCtl& pc, int kb, int ke, int sv, int pend,
                                const SlabCtl& sl, hipStream_t stream) {
  if constexpr (sizeof(T) == 8) {
    switch (R) {
      case 2: cg_persist1<T, CT, 2, 2, RECON, SYMV, true><<<grid, kPersistThreads, 0, stream>>>(a, pc, kb, ke, sv, pend, sl); break;
      case 4: cg_persist1<T, CT, 4, 2, RECON, SYMV, true><<<grid, kPersistThreads, 0, stream>>>(a, pc, kb, ke, sv, pend, sl); break;
      default: if constexpr (sizeof(CT) == 4 && SYMV) cg_persist1<T, CT, 16, 1, RECON, SYMV, true><<<grid, kPersistThreads, 0, stream>>>(a, pc, kb, ke, sv, pend, sl); break;
    }
  }
}
template <typename T, typename CT, bool RECON, bool SYMV>
static const void* slab_segment_kernel(int R) {
  if constexpr (sizeof(T) == 8) {
    switch (R) {
      case 2: return reinterpret_cast<const void*>(&cg_persist1<T, CT, 2, 2, RECON, SYMV, true>);
      case 4: return reinterpret_cast<const void*>(&cg_persist1<T, CT, 4, 2, RECON, SYMV, true>);
      default:                                              // (fp64 coefficients or an unsymmetric matrix - a general system - have no 16-row instance: it spills; cg.hip kHas16)
        if constexpr (sizeof(CT) == 4 && SYMV) return reinterpret_cast<const void*>(&cg_persist1<T, CT, 16, 1, RECON, SYMV, true>);
        else return nullptr;
    }
  }
  return nullptr;
}

// returns PISO_OK, an error, or kSlabRetry: a persistent segment failed on some rank -> the caller re-initialises the solve and
// calls again with allow_persist = false
constexpr int kSlabRetry = -1000;
template <typename T, typename CT, int V, bool RECON>
static int slab_iterate(std::vector<SlabRank<T>>& R, Comm<T>& comm, float accuracy, int max_iterations, int reset,
                        int* iterations_out, hipStream_t stream, bool symmetric, bool allow_persist, double global_cells,
                        unsigned* persist_ws) {
  const int nloc = (int)R.size();
  std::vector<int> g1(nloc), g2(nloc), gflat(nloc);
  for (int q = 0; q < nloc; ++q) {
    CgArgs<T>& a = R[q].a;
    const size_t n = (size_t)a.nx * a.ny;
    a.ntx = (a.nx + 64 * V - 1) / (64 * V);
    int rpw = (int)(((long long)a.ny * a.ntx) / (4 * 1024));
    rpw = rpw < 2 ? 2 : (rpw > 16 ? 16 : rpw);
    a.rows_per_wave = rpw;
    a.nty = (a.ny + 4 * rpw - 1) / (4 * rpw);
    a.accuracy = accuracy;
    g1[q] = grid_for((long long)a.ntx * a.nty, 1, 1024);
    g2[q] = grid_for((long long)((n / V + kBlock - 1) / kBlock), 4);
    gflat[q] = grid_for((long long)n, kBlock * 4);
    a.nA = g1[q]; a.nB = g2[q];
    a.gA = R[q].g; a.gB = R[q].g + 4;
  }
  if (!tl_slab_pinned) PISO_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&tl_slab_pinned), sizeof(SlabPinned), hipHostMallocDefault));

  // ---- persistent segments (peer transport, one rank per process): the NORMAL iterations of the slab run inside
  // cg_persist1<..., SLAB>; every rank takes the same decision (same shape, same options, failures are all-reduced)
  PersistShape shape;
  PersistCtl pc;
  pc.rec = nullptr; pc.err = nullptr; pc.nreg = 0; pc.ntx = 0; pc.timing = nullptr; pc.epoch0 = 0; pc.xcd = nullptr; pc.local_n = 0; pc.waves = kPersistWaves;
  SlabCtl sl;
  constexpr bool kCanSym = RECON && sizeof(CT) == 4;
  if (sizeof(T) == 8 && V == 16 / (int)sizeof(T) && allow_persist && comm.peer() && nloc == 1 && opt(OPT_CG_PERSIST) != 0 &&
      R[0].a.nx <= (int)comm.rccl->row_cap) {
    int dev = 0, cus = 0, per_cu = 0;
    PISO_HIP_CHECK(hipGetDevice(&dev));
    PISO_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    shape = persist_shape(R[0].a.nx, R[0].a.ny, V, cus, opt(OPT_CG_PERSIST_R));
    if (shape.R == 8) shape.R = 0;                          // (two regions of 8 rows: cg_persist1 spills registers there)
    if (shape.R == 16 && (sizeof(CT) == 8 || !(kCanSym && symmetric))) {     // (no 16-row slab instance for general matrices: regions of 4 / 2 rows, or two kernels)
      shape = PersistShape();
      if (opt(OPT_CG_PERSIST_R) <= 0) { shape = persist_shape(R[0].a.nx, R[0].a.ny, V, cus, 4); if (!shape.R) shape = persist_shape(R[0].a.nx, R[0].a.ny, V, cus, 2); }
    }
    if (shape.R && (size_t)R[0].a.nx * R[0].a.ny < 16384 && opt(OPT_CG_PERSIST) != 1) shape.R = 0;
    if (shape.R) {
      const void* kfn = slab_segment_kernel<T, CT, RECON, false>(shape.R);
      if constexpr (kCanSym) { if (symmetric) kfn = slab_segment_kernel<T, CT, RECON, true>(shape.R); }
      if (!kfn) shape.R = 0;
      else {
        PISO_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, kPersistThreads, 0));
        if ((long long)per_cu * cus < shape.grid || shape.grid > kPersistMaxGrid) shape.R = 0;
      }
    }
    if (shape.R) {
      pc.rec = reinterpret_cast<unsigned long long*>(persist_ws);
      pc.err = reinterpret_cast<int*>(persist_ws + kSlabPersistWsWords - 16);
      pc.nreg = shape.nreg; pc.ntx = shape.ntx;
      pc.xcd = reinterpret_cast<int*>(persist_ws + kPersistRecWords);
      PISO_HIP_CHECK(hipMemsetAsync(persist_ws, 0, kSlabPersistWsWords * sizeof(unsigned), stream));
      sl.pv = make_view(comm.rccl, comm.periodic_y);
      sl.ncells = global_cells;
      sl.rows_own = sl.pv.mbox[sl.pv.rank] + PeerLayout::kRows;
      sl.rows_lo = sl.pv.mbox[sl.pv.lower >= 0 ? sl.pv.lower : sl.pv.rank] + PeerLayout::kRows;
      sl.rows_hi = sl.pv.mbox[sl.pv.upper >= 0 ? sl.pv.upper : sl.pv.rank] + PeerLayout::kRows;
      sl.hop_ticks = opt(OPT_SLAB_HOP_TICKS) > 0 ? (unsigned)opt(OPT_SLAB_HOP_TICKS) : 0u;
    }
  }
  int seg_len = (int)(40000.0 / ((double)R[0].a.nx * R[0].a.ny * 8.5e-6 + 4.0));   // ~10 ms of work per segment at 2048^2 per GPU (as cg.hip)
  seg_len = seg_len < 50 ? 50 : (seg_len > 2000 ? 2000 : seg_len);
  if (opt(OPT_CG_SEGMENT) > 0) seg_len = opt(OPT_CG_SEGMENT);

  auto k1 = [&](int k, int mode, int sv, int chk, int pend) -> int {
    for (int q = 0; q < nloc; ++q) cg_k1<T, CT, V, RECON><<<g1[q], kBlock, 0, stream>>>(R[q].a, k, mode, sv, chk, pend);
    for (int q = 0; q < nloc; ++q) slab_collapse<T><<<1, kBlock, 0, stream>>>(R[q].a.partsA, g1[q], 3, R[q].g);
    return comm.allreduce(R, 0, 3, stream);
  };
  auto k2 = [&](int k, int sv) -> int {
    for (int q = 0; q < nloc; ++q) cg_k2<T, V><<<g2[q], kBlock, 0, stream>>>(R[q].a, k, sv);
    for (int q = 0; q < nloc; ++q) slab_collapse<T><<<1, kBlock, 0, stream>>>(R[q].a.partsB, g2[q], 3, R[q].g + 4);
    int rc = comm.allreduce(R, 4, 3, stream);
    if (rc != PISO_OK) return rc;
    return comm.exchange(R, 0, stream);                  // halo rows of the new residual
  };

  int sv = 0, stop_it = -1, k_last = -1, segments_run = 0;
  bool pending = false, finished = false;
  const int batch = 25;
  for (int k = 0; k < max_iterations && !finished; ++k) {
    const bool is_reset = ((k + 1) % reset == 0);
    int rc = PISO_OK;
    if (shape.R && k > 0 && !is_reset) {
      // NORMAL iterations [k, ke) in one launch: up to the next reset iteration / the end / one segment length
      int ke = max_iterations;
      { const int next_reset = ((k + 1 + reset - 1) / reset) * reset - 1; if (next_reset < ke) ke = next_reset; }
      if (ke > k + seg_len) ke = k + seg_len;
      if (ke > k) {
        PisoComm* c = comm.rccl;
        // tags: a 16-bit launch counter (the same on every rank) above a 16-bit exchange counter.  When the counter wraps, the
        // records of 65536 launches ago could pass for new ones: everybody waits for everybody, then clears its own.
        if ((c->launches & 0xffffu) == 0 && c->launches > 0) {
          rc = comm.allreduce(R, 12, 1, stream);
          if (rc == PISO_OK) PISO_HIP_CHECK(hipMemsetAsync(c->mbox[c->rank] + PeerLayout::kXcdRecs, 0, PeerLayout::kXcdRecBytes, stream));
          if (rc == PISO_OK) rc = comm.allreduce(R, 12, 1, stream);
          if (rc != PISO_OK) return rc;
        }
        pc.epoch0 = (c->launches++ & 0xffffu) << 16;
        PISO_HIP_CHECK(hipMemsetAsync(pc.rec, 0, kPersistZeroBytes, stream));   // records (both levels) + XCD arrivals
        bool launched = false;
        if constexpr (kCanSym) {
          if (symmetric) { launch_slab_segment<T, CT, RECON, true>(shape.R, shape.grid, R[0].a, pc, k, ke, sv, pending ? 1 : 0, sl, stream); launched = true; }
        }
        if (!launched) launch_slab_segment<T, CT, RECON, false>(shape.R, shape.grid, R[0].a, pc, k, ke, sv, pending ? 1 : 0, sl, stream);
        PISO_LAUNCH_CHECK();
        // did the segment fail anywhere?  (g[12] = my error flag, summed over the ranks)
        slab_err_to_sum<T><<<1, 64, 0, stream>>>(pc.err, c->err, R[0].g + 12);
        rc = comm.allreduce(R, 12, 1, stream);
        if (rc != PISO_OK) return rc;
        if constexpr (sizeof(T) == 8) PISO_HIP_CHECK(hipMemcpyAsync(&tl_slab_pinned->errsum, R[0].g + 12, sizeof(double), hipMemcpyDeviceToHost, stream));
        PISO_HIP_CHECK(hipMemcpyAsync(&tl_slab_pinned->st, &R[0].a.state[0], sizeof(CgState), hipMemcpyDeviceToHost, stream));
        PISO_HIP_CHECK(hipStreamSynchronize(stream));
        if (tl_slab_pinned->errsum != 0) {
          ++c->persist_fallbacks;
          PISO_HIP_CHECK(hipMemsetAsync(c->err, 0, sizeof(int), stream));
          return kSlabRetry;
        }
        c->persist_iterations += ke - k;
        ++segments_run;
        if (tl_slab_pinned->st.done) { finished = true; stop_it = tl_slab_pinned->st.iterations; }
        k_last = ke - 1;
        pending = false;                                   // the segment applies every x += alpha p itself
        k = ke - 1;
        continue;
      }
    }
    if (is_reset) {
      if (pending) { for (int q = 0; q < nloc; ++q) cg_flush_x<T><<<gflat[q], kBlock, 0, stream>>>(R[q].a, k - 1, sv); pending = false; }
      rc = comm.exchange(R, 1, stream);                  // halo rows of x for L x
      if (rc == PISO_OK) rc = k1(k, MODE_RESET, sv, k > 0 ? 1 : 0, 0);
      ++sv;
      for (int q = 0; q < nloc; ++q) cg_reset_residual<T><<<gflat[q], kBlock, 0, stream>>>(R[q].a, sv);
      if (rc == PISO_OK) rc = comm.exchange(R, 0, stream);
      if (rc == PISO_OK) rc = k1(k, MODE_INIT, sv, 0, 0);
    } else if (k == 0) {
      rc = k1(k, MODE_INIT, sv, 0, 0);
    } else {
      rc = k1(k, MODE_NORMAL, sv, 1, pending ? 1 : 0);
      ++sv;
    }
    if (rc == PISO_OK) rc = k2(k, sv);
    if (rc != PISO_OK) return rc;
    PISO_LAUNCH_CHECK();
    pending = true;
    k_last = k;
    if ((k + 1) % batch == 0 || k + 1 == max_iterations) {
      PISO_HIP_CHECK(hipMemcpyAsync(&tl_slab_pinned->st, &R[0].a.state[sv & 1], sizeof(CgState), hipMemcpyDeviceToHost, stream));
      PISO_HIP_CHECK(hipStreamSynchronize(stream));
      if (tl_slab_pinned->st.done) { finished = true; stop_it = tl_slab_pinned->st.iterations; }
    }
  }
  if (pending && k_last >= 0)
    for (int q = 0; q < nloc; ++q) cg_flush_x<T><<<gflat[q], kBlock, 0, stream>>>(R[q].a, k_last, sv);
  // ---- run-time verification of a solve that used the persistent slab kernel (as cg.hip does on one GPU; here it also covers
  // what crossed xGMI): r must still be b - A^ x for the x this rank returns.  Needs the neighbours' edge rows of x and the
  // global sum(x); the verdicts of all ranks are summed, so all ranks accept or all restart on the two-kernel iteration.
  if constexpr (sizeof(T) == 8) {
    if (segments_run > 0 && opt(OPT_CG_VERIFY) != 0) {
      CgArgs<T>& a0 = R[0].a;
      unsigned* out2 = reinterpret_cast<unsigned*>(pc.err) + 4;
      PISO_HIP_CHECK(hipMemsetAsync(out2, 0, 2 * sizeof(unsigned), stream));
      int rc = comm.exchange(R, 1, stream);
      const int gvf = grid_for((long long)a0.nx * a0.ny, kBlock * 4, 1024);
      cg_verify_sum_x<T><<<gvf, kBlock, 0, stream>>>(a0, a0.partsA);
      slab_collapse<T><<<1, kBlock, 0, stream>>>(a0.partsA, gvf, 1, R[0].g + 12);
      if (rc == PISO_OK) rc = comm.allreduce(R, 12, 1, stream);
      cg_verify_gap<T, CT><<<gvf, kBlock, 0, stream>>>(a0, a0.partsA, 0, out2, R[0].g + 12);
      slab_gap_to_sum<T><<<1, 64, 0, stream>>>(out2, R[0].g + 13, opt(OPT_CG_VERIFY) == 2 ? 1 : 0);
      if (rc == PISO_OK) rc = comm.allreduce(R, 13, 1, stream);
      if (rc != PISO_OK) return rc;
      PISO_LAUNCH_CHECK();
      PISO_HIP_CHECK(hipMemcpyAsync(&tl_slab_pinned->errsum, R[0].g + 13, sizeof(double), hipMemcpyDeviceToHost, stream));
      PISO_HIP_CHECK(hipStreamSynchronize(stream));
      ++comm.rccl->verify_runs;
      if (tl_slab_pinned->errsum != 0) {
        ++comm.rccl->verify_failures;
        ++comm.rccl->persist_fallbacks;
        return kSlabRetry;
      }
    }
  }
  for (int q = 0; q < nloc; ++q)
    slab_copy_rows<T><<<gflat[q], kBlock, 0, stream>>>(R[q].a.x, R[q].x_out, (size_t)R[q].a.nx * R[q].a.ny);
  PISO_LAUNCH_CHECK();
  if (comm.peer()) {
    if (comm.world > 1)       // every rank returns the same status (a wait may have given up on one rank only)
      peer_agree_on_error<><<<1, 64, 0, stream>>>(make_view(comm.rccl, comm.periodic_y), comm.rccl->err, ++comm.rccl->seq_ar);
    PISO_HIP_CHECK(hipMemcpyAsync(&tl_slab_pinned->err, comm.rccl->err, sizeof(int), hipMemcpyDeviceToHost, stream));
  }
  PISO_HIP_CHECK(hipStreamSynchronize(stream));
  if (comm.peer() && tl_slab_pinned->err) {
    PISO_HIP_CHECK(hipMemsetAsync(comm.rccl->err, 0, sizeof(int), stream));
    set_error_msg("slab CG: a wait on a peer's mailbox gave up (peer process gone or not running?)");
    return PISO_ERR_HIP;
  }
  if (iterations_out) *iterations_out = finished ? stop_it : max_iterations;
  return PISO_OK;
}

// Set up the ranks found in `R` (L, b, x_out, rank already filled in) inside `ws`, then iterate.
template <typename T>
static int slab_solve(std::vector<SlabRank<T>>& R, Comm<T>& comm, int nx, int nyl, int per_x, const T* const* b,
                      double global_cells, float accuracy, int max_iterations, int rank_deficient, int reset,
                      int* iterations_out, char* ws, size_t ws_per_rank, hipStream_t stream) {
  const int nloc = (int)R.size();
  const size_t n = (size_t)nx * nyl, nh = (size_t)nx * (nyl + 2);
  // the g buffers of all local ranks are contiguous (loopback all-reduce walks them with a fixed stride)
  T* gall = reinterpret_cast<T*>(ws);
  const size_t gbytes = align_up((size_t)nloc * 16 * sizeof(T), 256);
  comm.g_stride = 16;
  PISO_HIP_CHECK(hipMemsetAsync(gall, 0, gbytes, stream));
  for (int q = 0; q < nloc; ++q) {
    SlabRank<T>& k = R[q];
    Arena ar(ws + gbytes + (size_t)q * ws_per_rank, ws_per_rank);
    k.g = gall + (size_t)q * 16;
    k.cC = ar.take<T>(n); k.oT = ar.take<T>(4 * n); k.oF = ar.take<float>(4 * n);
    k.flags = ar.take<int>(4);
    k.rbase = ar.take<T>(nh); k.pbase[0] = ar.take<T>(nh); k.pbase[1] = ar.take<T>(nh); k.xbase = ar.take<T>(nh);
    CgArgs<T>& a = k.a;
    a.z = ar.take<T>(n);
    a.zp[0] = ar.take<T>(n); a.zp[1] = ar.take<T>(n);   // z' perimeters of the persistent kernel (agent-scope accesses only)
    k.persist_ws = ar.take<unsigned>(kSlabPersistWsWords);
    a.partsA = ar.take<T>(3 * kMaxPartials); a.partsB = ar.take<T>(3 * kMaxPartials); a.partsS = ar.take<T>(kMaxPartials);
    a.scal = ar.take<T>(SC_COUNT);
    a.state = ar.take<CgState>(2);
    if (!ar.ok()) { set_error_msg("piso_cg_solve_slab: workspace too small"); return PISO_ERR_INVALID_ARG; }
    a.cC = k.cC; a.b = b[q];
    a.r = k.rbase + nx; a.p[0] = k.pbase[0] + nx; a.p[1] = k.pbase[1] + nx; a.x = k.xbase + nx;
    a.nx = nx; a.ny = nyl; a.per_x = per_x; a.per_y = 2;
    a.gA = nullptr; a.gB = nullptr; a.nt = 0;
    a.nx_true = 0; a.ny_true = 0; a.ncells = 0.0;
    PISO_HIP_CHECK(hipMemsetAsync(k.flags, 0, 4 * sizeof(int), stream));
    PISO_HIP_CHECK(hipMemsetAsync(k.rbase, 0, nh * sizeof(T), stream));
    PISO_HIP_CHECK(hipMemsetAsync(k.pbase[0], 0, nh * sizeof(T), stream));
    PISO_HIP_CHECK(hipMemsetAsync(k.pbase[1], 0, nh * sizeof(T), stream));
    PISO_HIP_CHECK(hipMemsetAsync(k.xbase, 0, nh * sizeof(T), stream));
    cg_zero_partials<T><<<(3 * kMaxPartials + 255) / 256, 256, 0, stream>>>(a.partsA, a.partsB, a.partsS);
    const int gs = grid_for((long long)n, kBlock * 4);
    // symmetry is checked inside the slab (per_y = 2: the N entries of its last row pair with S entries on the neighbour - the
    // persistent kernel reads them from the N array, so nothing is assumed about that pair)
    cg_setup_coeffs<T><<<gs, kBlock, 0, stream>>>(k.L, k.cC, k.oT, k.oF, a.partsS, k.flags, n, nx, nyl, per_x, 2);
    slab_collapse<T><<<1, kBlock, 0, stream>>>(a.partsS, gs, 1, k.g + 8);
    slab_flags_to_sums<T><<<1, 64, 0, stream>>>(k.flags, k.g + 8);
  }
  PISO_LAUNCH_CHECK();
  { const int rc = comm.allreduce(R, 8, 4, stream); if (rc != PISO_OK) return rc; }
  T hg[4];
  PISO_HIP_CHECK(hipMemcpyAsync(hg, R[0].g + 8, 4 * sizeof(T), hipMemcpyDeviceToHost, stream));
  PISO_HIP_CHECK(hipStreamSynchronize(stream));
  const bool f32ok = sizeof(T) == 8 && hg[1] == 0 && !opt_on(OPT_CG_NO_COMPACT);
  const bool recon = f32ok && hg[2] == 0 && !opt_on(OPT_CG_NO_RECON);
  const bool symmetric = hg[3] == 0 && !opt_on(OPT_CG_NO_SYM);
  constexpr int VMID = 16 / sizeof(T);
  const bool vec = (nx % VMID == 0);
  for (int attempt = 0; attempt < 2; ++attempt) {
    // (second attempt: a persistent segment failed on some rank - every rank restarts the solve on the two-kernel iteration)
    for (int q = 0; q < nloc; ++q) {
      SlabRank<T>& k = R[q];
      const int gflat = grid_for((long long)n, kBlock * 4);
      if (attempt > 0) {
        PISO_HIP_CHECK(hipMemsetAsync(k.rbase, 0, nh * sizeof(T), stream));
        PISO_HIP_CHECK(hipMemsetAsync(k.pbase[0], 0, nh * sizeof(T), stream));
        PISO_HIP_CHECK(hipMemsetAsync(k.pbase[1], 0, nh * sizeof(T), stream));
        PISO_HIP_CHECK(hipMemsetAsync(k.xbase, 0, nh * sizeof(T), stream));
        cg_zero_partials<T><<<(3 * kMaxPartials + 255) / 256, 256, 0, stream>>>(k.a.partsA, k.a.partsB, k.a.partsS);
        k.a.gA = nullptr; k.a.gB = nullptr;
      }
      cg_init<T><<<gflat, kBlock, 0, stream>>>(k.a, rank_deficient, k.g + 8, global_cells);
      if (f32ok) { k.a.oS = k.oF; k.a.oW = k.oF + n; k.a.oE = k.oF + 2 * n; k.a.oN = k.oF + 3 * n; }
      else { k.a.oS = k.oT; k.a.oW = k.oT + n; k.a.oE = k.oT + 2 * n; k.a.oN = k.oT + 3 * n; }
    }
    { const int rc = comm.exchange(R, 0, stream); if (rc != PISO_OK) return rc; }     // halo rows of r0 = b
    int rc = PISO_OK;
#define PISO_SLAB_RUN(CT, V, RC) \
    rc = slab_iterate<T, CT, V, RC>(R, comm, accuracy, max_iterations, reset, iterations_out, stream, symmetric, attempt == 0, global_cells, R[0].persist_ws)
    if (f32ok && recon && vec) PISO_SLAB_RUN(float, VMID, true);
    else if (f32ok && recon) PISO_SLAB_RUN(float, 1, true);
    else if (f32ok && vec) PISO_SLAB_RUN(float, VMID, false);
    else if (f32ok) PISO_SLAB_RUN(float, 1, false);
    else if (vec) PISO_SLAB_RUN(T, VMID, false);
    else PISO_SLAB_RUN(T, 1, false);
#undef PISO_SLAB_RUN
    if (rc != kSlabRetry) return rc;
  }
  set_error_msg("slab CG: persistent segment failed twice");
  return PISO_ERR_HIP;
}

}  // namespace piso

using namespace piso;

extern "C" {

int piso_comm_unique_id(void* id128) {
  { const int rc = load_rccl(); if (rc != PISO_OK) return rc; }
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  PISO_NCCL_CHECK(g_rccl.GetUniqueId(static_cast<ncclUniqueId*>(id128)));
  return PISO_OK;
}

int piso_comm_create(const void* id128, int rank, int world, void** comm_out) {
  { const int rc = load_rccl(); if (rc != PISO_OK) return rc; }
  if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) { set_error_msg("piso_comm_create: invalid argument"); return PISO_ERR_INVALID_ARG; }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  PisoComm* c = new PisoComm;
  c->rank = rank; c->world = world; c->transport = TRANSPORT_RCCL;
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { set_error_msg(g_rccl.GetErrorString(r)); delete c; return PISO_ERR_HIP; }
  *comm_out = c;
  return PISO_OK;
}

int piso_comm_destroy(void* comm) {
  if (!comm) return PISO_OK;
  PisoComm* c = static_cast<PisoComm*>(comm);
  if (c->transport == TRANSPORT_RCCL) {
    if (g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  } else {
    (void)hipDeviceSynchronize();
    if (c->vmm) {
      for (int r = 0; r < c->world; ++r) {
        if (!c->mbox[r]) continue;
        (void)hipMemUnmap(c->mbox[r], c->vmm_bytes);
        (void)hipMemAddressFree(c->mbox[r], c->vmm_bytes);
        if (c->vmm_handle[r]) (void)hipMemRelease(c->vmm_handle[r]);
      }
    } else {
      for (int r = 0; r < c->world; ++r)
        if (r != c->rank && c->mbox[r]) (void)hipIpcCloseMemHandle(c->mbox[r]);
      if (c->mbox[c->rank]) (void)hipFree(c->mbox[c->rank]);
    }
    if (c->err) (void)hipFree(c->err);
  }
  delete c;
  return PISO_OK;
}

// ---- peer transport: create my mailbox (step 1), exchange the 64-byte handles by any means, connect (step 2)
int piso_comm_peer_create(int rank, int world, int row_capacity, void** comm_out, void* ipc_handle64_out) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  if (!comm_out || !ipc_handle64_out || world < 1 || world > kMaxRanks || rank < 0 || rank >= world || row_capacity < 1) {
    set_error_msg("piso_comm_peer_create: invalid argument (at most 8 ranks: the GPUs of one node)");
    return PISO_ERR_INVALID_ARG;
  }
  PisoComm* c = new PisoComm;
  c->rank = rank; c->world = world; c->transport = TRANSPORT_PEER;
  c->row_cap = align_up((size_t)row_capacity, 32);
  c->mbox_bytes = PeerLayout::bytes(c->row_cap);
  void* mb = nullptr;
  // uncached + fine-grained: a peer's write over xGMI is visible to my system-scope loads without any cache maintenance
  hipError_t e = hipExtMallocWithFlags(&mb, c->mbox_bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) { set_error("hipExtMallocWithFlags(mailbox)", e); delete c; return PISO_ERR_HIP; }
  c->mbox[rank] = static_cast<char*>(mb);
  e = hipMemset(mb, 0, c->mbox_bytes);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->err), sizeof(int));
  if (e == hipSuccess) e = hipMemset(c->err, 0, sizeof(int));
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipIpcGetMemHandle(static_cast<hipIpcMemHandle_t*>(ipc_handle64_out), mb);
  if (e != hipSuccess) { set_error("piso_comm_peer_create", e); (void)hipFree(mb); if (c->err) (void)hipFree(c->err); delete c; return PISO_ERR_HIP; }
  c->connected = (world == 1);
  *comm_out = c;
  return PISO_OK;
}

int piso_comm_peer_connect(void* comm, const void* ipc_handles64_all_ranks) {
  PisoComm* c = static_cast<PisoComm*>(comm);
  if (!c || c->transport != TRANSPORT_PEER || !ipc_handles64_all_ranks) { set_error_msg("piso_comm_peer_connect: invalid argument"); return PISO_ERR_INVALID_ARG; }
  const char* h = static_cast<const char*>(ipc_handles64_all_ranks);
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank || c->mbox[r]) continue;
    hipIpcMemHandle_t handle;
    memcpy(&handle, h + (size_t)r * 64, 64);
    void* p = nullptr;
    PISO_HIP_CHECK(hipIpcOpenMemHandle(&p, handle, hipIpcMemLazyEnablePeerAccess));
    c->mbox[r] = static_cast<char*>(p);
  }
  c->connected = true;
  return PISO_OK;
}

// ---- the same mailboxes through the virtual-memory API, for nodes whose driver refuses hipIpcGetMemHandle across ranks: the allocation
// is created exportable (hipMemCreate, uncached type), exported as a POSIX file descriptor, handed to the other ranks by the caller
// (a Unix socket with SCM_RIGHTS: diffpiso/distributed.py) and imported + mapped there.  Everything else of the transport is unchanged.
static int vmm_map(PisoComm* c, int r, hipMemGenericAllocationHandle_t h, int dev) {
  void* p = nullptr;
  hipError_t e = hipMemAddressReserve(&p, c->vmm_bytes, 0, nullptr, 0);
  if (e != hipSuccess) { set_error("hipMemAddressReserve(mailbox)", e); return PISO_ERR_HIP; }
  e = hipMemMap(p, c->vmm_bytes, 0, h, 0);
  if (e != hipSuccess) { set_error("hipMemMap(mailbox)", e); (void)hipMemAddressFree(p, c->vmm_bytes); return PISO_ERR_HIP; }
  hipMemAccessDesc acc{};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(p, c->vmm_bytes, &acc, 1);
  if (e != hipSuccess) { set_error("hipMemSetAccess(mailbox)", e); (void)hipMemUnmap(p, c->vmm_bytes); (void)hipMemAddressFree(p, c->vmm_bytes); return PISO_ERR_HIP; }
  c->mbox[r] = static_cast<char*>(p);
  c->vmm_handle[r] = h;
  return PISO_OK;
}

int piso_comm_peer_create_fd(int rank, int world, int row_capacity, void** comm_out, int* fd_out) {
  if (!comm_out || !fd_out || world < 1 || world > kMaxRanks || rank < 0 || rank >= world || row_capacity < 1) {
    set_error_msg("piso_comm_peer_create_fd: invalid argument (at most 8 ranks: the GPUs of one node)");
    return PISO_ERR_INVALID_ARG;
  }
  int dev = 0;
  PISO_HIP_CHECK(hipGetDevice(&dev));
  PisoComm* c = new PisoComm;
  c->rank = rank; c->world = world; c->transport = TRANSPORT_PEER; c->vmm = 1;
  c->row_cap = align_up((size_t)row_capacity, 32);
  c->mbox_bytes = PeerLayout::bytes(c->row_cap);
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypeUncached;               // (as hipDeviceMallocUncached: a peer's write is visible to my system-scope loads)
  prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
  if (e != hipSuccess || gran == 0) { set_error("hipMemGetAllocationGranularity(mailbox)", e); delete c; return PISO_ERR_HIP; }
  c->vmm_bytes = align_up(c->mbox_bytes, gran);
  hipMemGenericAllocationHandle_t h{};
  e = hipMemCreate(&h, c->vmm_bytes, &prop, 0);
  if (e != hipSuccess) { set_error("hipMemCreate(mailbox, uncached, exportable)", e); delete c; return PISO_ERR_HIP; }
  int rc = vmm_map(c, rank, h, dev);
  if (rc != PISO_OK) { (void)hipMemRelease(h); delete c; return rc; }
  int fd = -1;
  e = hipMemset(c->mbox[rank], 0, c->mbox_bytes);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->err), sizeof(int));
  if (e == hipSuccess) e = hipMemset(c->err, 0, sizeof(int));
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0);
  if (e != hipSuccess) { set_error("piso_comm_peer_create_fd", e); (void)piso_comm_destroy(c); return PISO_ERR_HIP; }
  *fd_out = fd;                                           // the caller closes it once every peer has received its copy
  c->connected = (world == 1);
  *comm_out = c;
  return PISO_OK;
}

int piso_comm_peer_connect_fd(void* comm, const int* fds_all_ranks) {
  PisoComm* c = static_cast<PisoComm*>(comm);
  if (!c || c->transport != TRANSPORT_PEER || !c->vmm || !fds_all_ranks) { set_error_msg("piso_comm_peer_connect_fd: invalid argument"); return PISO_ERR_INVALID_ARG; }
  int dev = 0;
  PISO_HIP_CHECK(hipGetDevice(&dev));
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank || c->mbox[r]) continue;
    hipMemGenericAllocationHandle_t h{};
    // (this runtime reads the descriptor THROUGH the pointer - handing the integer over as the pointer's value, as the CUDA driver API
    // takes it, makes it dereference address `fd`)
    int fd = fds_all_ranks[r];
    hipError_t e = hipMemImportFromShareableHandle(&h, static_cast<void*>(&fd), hipMemHandleTypePosixFileDescriptor);
    if (e != hipSuccess) { set_error("hipMemImportFromShareableHandle(mailbox)", e); return PISO_ERR_HIP; }
    const int rc = vmm_map(c, r, h, dev);
    if (rc != PISO_OK) { (void)hipMemRelease(h); return rc; }
  }
  c->connected = true;
  return PISO_OK;
}

// Round-trip time of one tagged word between ranks a and b through the mailboxes (`iters` round trips; a == b: a rank's own mailbox).
// EVERY rank calls it with the same arguments; ranks other than a and b return at once.  us_out (host float, written on rank a only):
// microseconds per round trip, timed with events around the initiator's kernel; the one-way hop is half of it.
int piso_comm_pingpong(void* comm, int a, int b, int iters, float* us_out, piso_stream_t stream_) {
  PisoComm* pc = static_cast<PisoComm*>(comm);
  if (!pc || pc->transport != TRANSPORT_PEER || !pc->connected || a < 0 || b < 0 || a >= pc->world || b >= pc->world || iters < 1) {
    set_error_msg("piso_comm_pingpong: needs a connected peer communicator and two of its ranks");
    return PISO_ERR_INVALID_ARG;
  }
  const unsigned seq0 = pc->seq_pp + 1;
  pc->seq_pp += (unsigned)iters + 1;                      // (advances identically on every rank: all of them make every call)
  if (pc->rank != a && pc->rank != b) return PISO_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const PeerView pv = make_view(pc, true);
  hipEvent_t e0, e1;
  PISO_HIP_CHECK(hipEventCreate(&e0));
  PISO_HIP_CHECK(hipEventCreate(&e1));
  PISO_HIP_CHECK(hipEventRecord(e0, stream));
  peer_pingpong<<<1, 64, 0, stream>>>(pv, a, b, iters, seq0, pc->err);
  PISO_HIP_CHECK(hipEventRecord(e1, stream));
  PISO_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0;
  PISO_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (us_out && pc->rank == a) *us_out = 1e3f * ms / (float)iters;
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}

// what the communicator did so far: [0] transport (1 RCCL, 2 peer mailboxes), [1] CG iterations executed inside persistent slab
// segments, [2] solves restarted on the two-kernel iteration after a segment failed, [3] persistent launches
int piso_comm_stats(void* comm, long long* out4) {      // (six values: see include/piso_hip.h)
  PisoComm* c = static_cast<PisoComm*>(comm);
  if (!c || !out4) { set_error_msg("piso_comm_stats: invalid argument"); return PISO_ERR_INVALID_ARG; }
  out4[0] = c->transport; out4[1] = c->persist_iterations; out4[2] = c->persist_fallbacks; out4[3] = c->launches;
  out4[4] = c->verify_runs; out4[5] = c->verify_failures;
  return PISO_OK;
}

// Halo rows of ANY globally indexed vector of the slab-decomposed step (faces, cells, CSR values): four messages of up to three
// element segments each, in the order {to the upper neighbour, to the lower neighbour, from the lower, from the upper};
// msgs28 = 4 x {count, off[3], len[3]} (element offsets into `vec`).  Ring neighbours always (without a periodic y axis the wrap
// rows travel and nobody reads them).  One launch; the elements cross xGMI as 8-byte words written into the consumer's mailbox.
int piso_comm_exchange(void* comm, void* vec, int dtype, const int* msgs28, piso_stream_t stream_) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  PisoComm* pc = static_cast<PisoComm*>(comm);
  if (!pc || !vec || !msgs28) { set_error_msg("piso_comm_exchange: invalid argument"); return PISO_ERR_INVALID_ARG; }
  if (pc->world == 1 && opt(OPT_SLAB_FORCE) <= 0) return PISO_OK;        // (slab_force: test knob - a ring of one rank exchanges with itself)
  if (pc->transport == TRANSPORT_PEER && !pc->connected) { set_error_msg("piso_comm_exchange: the peer communicator is not connected"); return PISO_ERR_INVALID_ARG; }
  HaloMsg m[4];
  for (int q = 0; q < 4; ++q) {
    m[q].count = msgs28[7 * q];
    size_t total = 0;
    if (m[q].count < 0 || m[q].count > 3) { set_error_msg("piso_comm_exchange: at most three segments per message"); return PISO_ERR_INVALID_ARG; }
    for (int k = 0; k < 3; ++k) {
      m[q].off[k] = msgs28[7 * q + 1 + k]; m[q].len[k] = msgs28[7 * q + 4 + k];
      if (k < m[q].count) { if (m[q].off[k] < 0 || m[q].len[k] < 0) { set_error_msg("piso_comm_exchange: negative segment"); return PISO_ERR_INVALID_ARG; } total += (size_t)m[q].len[k]; }
    }
    if (pc->transport == TRANSPORT_PEER && total > pc->row_cap) { set_error_msg("piso_comm_exchange: message longer than the communicator's row_capacity"); return PISO_ERR_INVALID_ARG; }
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (dtype < 0 || dtype > 2) { set_error_msg("piso_comm_exchange: dtype must be 0 (float), 1 (double) or 2 (int32)"); return PISO_ERR_INVALID_ARG; }
  if (pc->transport == TRANSPORT_RCCL) return comm_rccl_exchange_segments(pc, vec, dtype, m, stream);
  const PeerView pv = make_view(pc, true);
  const unsigned seq = ++pc->seq_ex;
  if (dtype == 0) peer_exchange_segments<float><<<2, 256, 0, stream>>>(pv, static_cast<float*>(vec), m[0], m[1], m[2], m[3], seq, pc->err);
  else if (dtype == 1) peer_exchange_segments<double><<<2, 256, 0, stream>>>(pv, static_cast<double*>(vec), m[0], m[1], m[2], m[3], seq, pc->err);
  else if (dtype == 2) peer_exchange_segments<int><<<2, 256, 0, stream>>>(pv, static_cast<int*>(vec), m[0], m[1], m[2], m[3], seq, pc->err);
  else { set_error_msg("piso_comm_exchange: dtype must be 0 (float), 1 (double) or 2 (int32)"); return PISO_ERR_INVALID_ARG; }
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}
// did any wait on a peer give up since the last call?  (agreed over the ranks; synchronises the stream)
int piso_comm_check(void* comm, piso_stream_t stream_) {
  PisoComm* pc = static_cast<PisoComm*>(comm);
  if (!pc) { set_error_msg("piso_comm_check: NULL communicator"); return PISO_ERR_INVALID_ARG; }
  if (pc->transport != TRANSPORT_PEER || pc->world == 1) return PISO_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int herr = 0;
  peer_agree_on_error<><<<1, 64, 0, stream>>>(make_view(pc, true), pc->err, ++pc->seq_ar);
  PISO_HIP_CHECK(hipMemcpyAsync(&herr, pc->err, sizeof(int), hipMemcpyDeviceToHost, stream));
  PISO_HIP_CHECK(hipStreamSynchronize(stream));
  if (herr) {
    PISO_HIP_CHECK(hipMemsetAsync(pc->err, 0, sizeof(int), stream));
    set_error_msg("piso_comm_check: a wait on a peer's mailbox gave up (peer process gone or not running?)");
    return PISO_ERR_HIP;
  }
  return PISO_OK;
}

size_t piso_cg_slab_workspace_bytes(int nx, int ny_local, int local_ranks) {
  return (size_t)local_ranks * slab_rank_bytes<double>(nx, ny_local) + align_up((size_t)local_ranks * 16 * sizeof(double), 256) + 4096;
}

int piso_cg_solve_slab_f64(void* comm, int nx, int ny_local, int periodic_x, int periodic_y, const double* laplace_local,
                           const double* divergence_local, double* x_out_local, double* x_out_global, float accuracy,
                           int max_iterations, int rank_deficient, int residual_reset, int* iterations_out, void* workspace,
                           size_t workspace_bytes, piso_stream_t stream_) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  if (!comm || nx < 1 || ny_local < 1 || !laplace_local || !divergence_local || !x_out_local || !workspace || residual_reset < 1) {
    set_error_msg("piso_cg_solve_slab_f64: invalid argument");
    return PISO_ERR_INVALID_ARG;
  }
  if (workspace_bytes < piso_cg_slab_workspace_bytes(nx, ny_local, 1)) { set_error_msg("piso_cg_solve_slab_f64: workspace too small"); return PISO_ERR_INVALID_ARG; }
  PisoComm* pc = static_cast<PisoComm*>(comm);
  if (pc->transport == TRANSPORT_PEER && !pc->connected) { set_error_msg("piso_cg_solve_slab_f64: peer communicator not connected"); return PISO_ERR_INVALID_ARG; }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  std::vector<SlabRank<double>> R(1);
  R[0].L = laplace_local; R[0].x_out = x_out_local; R[0].rank = pc->rank;
  Comm<double> cm;
  cm.world = pc->world; cm.periodic_y = periodic_y != 0; cm.rccl = pc; cm.g_stride = 16;
  const double* b[1] = {divergence_local};
  const int rc = slab_solve<double>(R, cm, nx, ny_local, periodic_x, b, (double)nx * ny_local * pc->world, accuracy, max_iterations,
                                    rank_deficient, residual_reset, iterations_out, static_cast<char*>(workspace),
                                    slab_rank_bytes<double>(nx, ny_local), stream);
  if (rc != PISO_OK) return rc;
  if (x_out_global) {                                       // every rank receives the whole field (replicated PISO step)
    if (pc->world == 1) {
      PISO_HIP_CHECK(hipMemcpyAsync(x_out_global, x_out_local, (size_t)nx * ny_local * sizeof(double), hipMemcpyDeviceToDevice, stream));
    } else if (pc->transport == TRANSPORT_PEER) {
      set_error_msg("piso_cg_solve_slab_f64: x_out_global is an RCCL all-gather; with the peer transport pass NULL and gather the slabs yourself");
      return PISO_ERR_INVALID_ARG;
    } else {
      PISO_NCCL_CHECK(g_rccl.AllGather(x_out_local, x_out_global, (size_t)nx * ny_local, ncclDouble, pc->comm, stream));
    }
    PISO_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return PISO_OK;
}

int piso_cg_solve_slab_emulated_f64(int slabs, int nx, int ny, int periodic_x, int periodic_y, const double* laplace,
                                    const double* divergence, double* x_out, float accuracy, int max_iterations,
                                    int rank_deficient, int residual_reset, int* iterations_out, void* workspace,
                                    size_t workspace_bytes, piso_stream_t stream_) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  if (slabs < 1 || nx < 1 || ny < slabs || ny % slabs != 0 || !laplace || !divergence || !x_out || !workspace || residual_reset < 1) {
    set_error_msg("piso_cg_solve_slab_emulated_f64: invalid argument (ny must be a multiple of slabs)");
    return PISO_ERR_INVALID_ARG;
  }
  const int nyl = ny / slabs;
  if (workspace_bytes < piso_cg_slab_workspace_bytes(nx, nyl, slabs)) { set_error_msg("piso_cg_solve_slab_emulated_f64: workspace too small"); return PISO_ERR_INVALID_ARG; }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  std::vector<SlabRank<double>> R(slabs);
  std::vector<const double*> b(slabs);
  for (int r = 0; r < slabs; ++r) {
    const size_t off = (size_t)r * nyl * nx;
    R[r].L = laplace + off * 5; R[r].x_out = x_out + off; R[r].rank = r;
    b[r] = divergence + off;
  }
  Comm<double> cm;
  cm.world = slabs; cm.periodic_y = periodic_y != 0; cm.rccl = nullptr; cm.g_stride = 16;
  return slab_solve<double>(R, cm, nx, nyl, periodic_x, b.data(), (double)nx * ny, accuracy, max_iterations, rank_deficient,
                            residual_reset, iterations_out, static_cast<char*>(workspace), slab_rank_bytes<double>(nx, nyl), stream);
}

}  // extern "C"
