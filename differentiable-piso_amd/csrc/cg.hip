// Host driver of the single-GPU pressure CG (kernels: cg_kernels.h).  See cg_kernels.h for the design.
#include <atomic>
#include "cg_kernels.h"
#include "cg_persist.h"
#include "cg_persist1.h"
#include "cg_tiny.h"
#include "options.h"
#include <cstdio>
#include <vector>

namespace piso {

// ---------------------------------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------------------------------
struct CgProfile {
  int enabled = 0, stride = 8;
  double ms[4] = {0, 0, 0, 0};          // K1, K2, persistent segments, (unused)
  long long count[4] = {0, 0, 0, 0};    // launches of K1, K2; ITERATIONS executed inside persistent segments; segment LAUNCHES
};
static CgProfile g_prof;
constexpr size_t kPersistWsWords = kPersistWsWordsAll;   // exchange records + control words (cg_persist.h)

struct HostPoll {
  CgState* pinned = nullptr;   // [2]
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipEvent_t seg_ev[2] = {nullptr, nullptr};   // timing events around persistent segments (profiling only; created once)
};
static std::atomic<unsigned> g_persist_launches{0};   // persistent launches so far: the high half of their exchange tags
static int g_persist_fallbacks = 0;            // solves that were restarted on the two-kernel path after an exchange timed out
// option cg_xcd_map 1 (tests): the XCD every workgroup of the solve's LAST chip-wide persistent launch ran on (cg_persist1.h: hier_enter)
static thread_local int tl_xcd_map[kPersistMaxGrid];
static thread_local int tl_xcd_map_n = 0;
static long long g_tiny_solves = 0;            // solves that ran inside one workgroup (cg_tiny.h)
static bool g_xcd_local_failed = false;        // an XCD-local launch gave up once (the device does not behave as assumed): not tried again
static long long g_verify_runs = 0;            // solves whose final state was checked against the true residual (cg_verify_gap)
static int g_verify_failures = 0;              // ... and failed: restarted on the two-kernel path
// one per device and thread (events belong to the device that was current when they were created); ensure_poll() selects
constexpr int kPollDevices = 16;
static thread_local HostPoll tl_poll_dev[kPollDevices];
static thread_local HostPoll* tl_poll_cur = &tl_poll_dev[0];
#define tl_poll (*tl_poll_cur)

static int ensure_poll() {
  int dev = 0;
  PISO_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= kPollDevices) { set_error_msg("piso_cg_solve: device ordinal out of range"); return PISO_ERR_INVALID_ARG; }
  tl_poll_cur = &tl_poll_dev[dev];
  if (!tl_poll.pinned) {
    PISO_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&tl_poll.pinned), 2 * sizeof(CgState), hipHostMallocDefault));
    PISO_HIP_CHECK(hipEventCreateWithFlags(&tl_poll.ev[0], hipEventDisableTiming));
    PISO_HIP_CHECK(hipEventCreateWithFlags(&tl_poll.ev[1], hipEventDisableTiming));
  }
  return PISO_OK;
}

// Padded-grid mode: a wall-bounded grid the persistent kernel cannot tile (row length not a multiple of 128 cells, rows not a
// multiple of the region height - e.g. the lid-driven cavity's 64 x 65) is embedded in the next grid it can: zero coefficients
// and a zero right-hand side keep the padding at zero (the kernels that add the rank-1 shift skip it, cg_kernels.h).  Only for
// small grids, where two dependent launches per iteration cost ~14 us against ~4 us of a persistent iteration; periodic axes
// cannot be padded (the wrap partner would move).
constexpr size_t kPadMaxCells = (size_t)1 << 19;
static bool padded_dims(int nx, int ny, int per_x, int per_y, int elem, int* nxp, int* nyp) {
  *nxp = nx; *nyp = ny;
  if (elem != 8 || nx % 2 != 0 || nx < 8 || ny < 8) return false;
  // grids the kernel tiles as they are stay as they are: rows of whole 128-cell strips, an even number of rows (regions of 2)
  const bool pad_x = nx % 128 != 0, pad_y = ny % 2 != 0;
  if (!pad_x && !pad_y) return false;
  const int px = (nx + 127) / 128 * 128;
  int py = (ny + 3) / 4 * 4;                      // rows a multiple of 4: the number of 2-row regions comes out even (two per wave)
  if (px != nx && per_x) return false;
  if (py != ny && per_y) {
    py = ny;                                      // periodic in y: only x is padded, if the region count still works out
    if (ny % 2 != 0 || ((long long)(px / 128) * (ny / 2)) % 2 != 0) return false;
  }
  if ((size_t)px * py > kPadMaxCells) return false;
  *nxp = px; *nyp = py;
  return true;
}

template <typename T>
static size_t cg_workspace_bytes(int nx_in, int ny_in) {
  int nx = nx_in, ny = ny_in;
  const bool padded = padded_dims(nx_in, ny_in, 0, 0, (int)sizeof(T), &nx, &ny);   // (an upper bound: periodic grids are never padded)
  const size_t n = (size_t)nx * ny + (padded ? 2 * (size_t)nx * ny : 0);             // + padded copies of b and x
  size_t b = 0;
  b += 11 * align_up(n * sizeof(T), 256);                // diag + 4 off-diagonal arrays (T) + r, z, p0, p1 + the two z' perimeter buffers
  b += align_up(4 * n * sizeof(float), 256) + 256;        // float copy of the off-diagonals + flag
  b += 3 * align_up(3 * kMaxPartials * sizeof(T), 256);
  b += align_up(SC_COUNT * sizeof(T), 256) + align_up(2 * sizeof(CgState), 256) + 512;
  b += align_up(kPersistWsWords * sizeof(unsigned), 256);  // exchange records of the persistent kernel
  return b + 4096;
}

// Which (state, coefficient, matrix) combinations have a 16-row instance at all: the ones that keep their registers.  fp32 state
// without a symmetric matrix with rebuilt diagonals (26-84 spilled vector registers) and fp64 COEFFICIENTS (a general matrix: 8
// spilled vector registers) are tiled with regions of 4 / 2 rows instead - those instances spill nothing - or iterate on the
// two-kernel path; the spilling instances are not compiled.
template <typename T, typename CT, bool RECON, bool SYMV>
constexpr bool kHas16 = (sizeof(T) == 8 && sizeof(CT) == 4) || (sizeof(T) == 4 && sizeof(CT) == 4 && RECON && SYMV);

template <typename T, typename CT, bool RECON, bool SYMV>
static const void* persist_kernel(int R, bool ragged = false, int NQ = 0) {
  if constexpr (sizeof(T) == 8 && RECON && SYMV) {
    if (ragged) {
      switch (R) {
        case 2: return reinterpret_cast<const void*>(&cg_persist1<T, CT, 2, 2, RECON, SYMV, false, true>);
        case 4: return reinterpret_cast<const void*>(&cg_persist1<T, CT, 4, 2, RECON, SYMV, false, true>);
        default: return reinterpret_cast<const void*>(&cg_persist1<T, CT, 16, 1, RECON, SYMV, false, true>);
      }
    }
  }
  if constexpr (sizeof(T) == 8 && sizeof(CT) == 4 && SYMV) {
    if (R == 2 && NQ == 1) return reinterpret_cast<const void*>(&cg_persist1<T, CT, 2, 1, RECON, SYMV>);
  }
  switch (R) {
    case 2: return reinterpret_cast<const void*>(&cg_persist1<T, CT, 2, 2, RECON, SYMV>);
    case 4: return reinterpret_cast<const void*>(&cg_persist1<T, CT, 4, 2, RECON, SYMV>);
    default:
      if constexpr (kHas16<T, CT, RECON, SYMV>) return reinterpret_cast<const void*>(&cg_persist1<T, CT, 16, 1, RECON, SYMV>);
      else return nullptr;
  }
}

struct EventPool {
  static constexpr int kMax = 64;
  hipEvent_t start[2][kMax], stop[2][kMax];
  int used[2] = {0, 0};
  bool created = false;
};
static thread_local EventPool tl_events;

template <typename T, typename CT, int V, bool RECON>
static int cg_run(CgArgs<T> a, unsigned* persist_ws, bool symmetric, float accuracy, int max_iterations, int rank_deficient, int reset, int fixed,
                  int* iterations_out, float* kernel_ms_out, hipStream_t stream, bool allow_persist = true) {
  const int nx = a.nx, ny = a.ny;
  const size_t n = (size_t)nx * ny;
  a.ntx = (nx + 64 * V - 1) / (64 * V);
  int rpw = (int)(((long long)ny * a.ntx) / (4 * 1024));
  rpw = rpw < 2 ? 2 : (rpw > 16 ? 16 : rpw);
  if (opt(OPT_CG_RPW) > 0) rpw = opt(OPT_CG_RPW);                                             // tuning knob
  a.rows_per_wave = rpw;
  a.nty = (ny + 4 * rpw - 1) / (4 * rpw);
  a.accuracy = fixed ? -1.0f : accuracy;                 // fixed-work mode: the test can never succeed
  int cap = 1024;
  if (opt(OPT_CG_MAXBLOCKS) >= 8 && opt(OPT_CG_MAXBLOCKS) <= kMaxPartials) cap = opt(OPT_CG_MAXBLOCKS);
  const int g1 = grid_for((long long)a.ntx * a.nty, 1, cap);
  const int g2 = grid_for((long long)((n / V + kBlock - 1) / kBlock), 4);
  a.nA = g1; a.nB = g2;
  const int gflat = grid_for((long long)n, kBlock * 4);

  { const int rc = ensure_poll(); if (rc != PISO_OK) return rc; }
  const bool prof = (kernel_ms_out != nullptr) || g_prof.enabled;
  EventPool& ep = tl_events;
  if (prof && !ep.created) {
    for (int q = 0; q < 2; ++q)
      for (int i = 0; i < EventPool::kMax; ++i) {
        PISO_HIP_CHECK(hipEventCreate(&ep.start[q][i]));
        PISO_HIP_CHECK(hipEventCreate(&ep.stop[q][i]));
      }
    ep.created = true;
  }
  ep.used[0] = ep.used[1] = 0;
  const int prof_stride = g_prof.stride > 0 ? g_prof.stride : 8;

  cg_init<T><<<gflat, kBlock, 0, stream>>>(a, rank_deficient);
  PISO_LAUNCH_CHECK();

  // poll cadence: about 1 ms of work between host looks, never fewer than 10 iterations
  double t_iter_us = (double)n * 120.0 / 4.0e6;          // ~4 TB/s
  if (t_iter_us < 8.0) t_iter_us = 8.0;
  int batch = (int)(1000.0 / t_iter_us);
  batch = batch < 10 ? 10 : (batch > 200 ? 200 : batch);

  int sv = 0, polls = 0, stop_it = -1;
  const int total = fixed ? fixed : max_iterations;
  bool finished = false;
  auto inspect = [&](int slot) -> int {                  // wait for poll `slot`, return 1 if the solver reported done
    hipError_t e = hipEventSynchronize(tl_poll.ev[slot]);
    if (e != hipSuccess) { set_error("hipEventSynchronize", e); return -1; }
    if (tl_poll.pinned[slot].done) { stop_it = tl_poll.pinned[slot].iterations; return 1; }
    return 0;
  };
  bool pending = false;                                  // x still lacks alpha_k p_k of the last executed iteration
  int k_last = -1;
  // ---- persistent segments (cg_persist1.h): applicable when every wave's region fits on chip
  int persist_R = 0, persist_NQ = 0, persist_grid = 0;
  PersistCtl pc;
  pc.rec = nullptr; pc.err = nullptr; pc.nreg = 0; pc.ntx = 0; pc.timing = nullptr; pc.epoch0 = 0; pc.xcd = nullptr; pc.local_n = 0; pc.waves = kPersistWaves;
  bool xcd_local = false;                                // the solve runs on the workgroups of one XCD (cg_persist1<..., LOCAL>)
  constexpr int kXcdCus = 32;                            // CUs of one MI355X XCD
  const int force = opt(OPT_CG_PERSIST), force_r = opt(OPT_CG_PERSIST_R);   // -1: automatic
  // fp32 state: the 16-row instance keeps its registers only for a symmetric matrix with rebuilt diagonals (the others spill
  // 26-84 VGPRs); any other fp32 system is tiled with regions of 4 / 2 rows (no spills), or iterates on the two-kernel path
  const bool f32_small_regions = (sizeof(T) != 8 && !(symmetric && RECON)) || sizeof(CT) == 8;     // (see kHas16)
  if (V == 16 / (int)sizeof(T) && a.per_y != 2 && allow_persist && force != 0) {
    int dev = 0, cus = 0;
    PISO_HIP_CHECK(hipGetDevice(&dev));
    PISO_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    PersistShape shape = persist_shape(nx, ny, V, cus, force_r);
    if (f32_small_regions && shape.R == 16) {
      shape = PersistShape();
      if (force_r <= 0) { shape = persist_shape(nx, ny, V, cus, 4); if (!shape.R) shape = persist_shape(nx, ny, V, cus, 2); }
    }
    // XCD-local mode: symmetric compact coefficients, fp64, one exchange, at most one XCD's worth of workgroups
    constexpr bool kLocalKernel = RECON && sizeof(CT) == 4 && sizeof(T) == 8;
    const bool local_ok = kLocalKernel && symmetric && opt(OPT_CG_XCD_LOCAL) != 0 && !g_xcd_local_failed &&
                          cus == kXcds * kXcdCus;
    // (measured at 2048^2-class work per workgroup: regions of 4 rows to make a 64-workgroup grid fit one XCD lose more in the row
    // loops than the shorter exchange wins - 512^2: 6.2 against 4.5 us per iteration; 256^2, 16 workgroups either way: 3.8 against 4.3)
    persist_R = shape.R; persist_NQ = shape.NQ; persist_grid = shape.grid; pc.nreg = shape.nreg; pc.ntx = shape.ntx;

    // Small regions: ONE wave with work per SIMD instead of two (waves 4-7 of a workgroup own nothing), twice the workgroups, wherever
    // the doubled grid still fits the chip: a wave then never waits at the exchange's first barrier for the wave it shares a SIMD
    // with (0.6 us of a ~4 us iteration).  Measured: 256^2 3.80 -> 3.47 us per iteration, 512^2 4.45 -> 4.08, 1024 x 256 4.48 -> 4.08,
    // 1024 x 512 unchanged.  Option cg_persist_half 0: never, 1: wherever it fits.  Automatic (-1) leaves out the one case where the
    // doubling would push a grid that fits ONE XCD (17-32 workgroups) out of it: since the XCD-local exchange polls its own XCD's
    // records only, 32 full workgroups there beat 64 half ones chip-wide (512 x 256, round 4: 3.98 against 4.19 us per iteration).
    {
      const int half = opt(OPT_CG_PERSIST_HALF);
      const bool fits = (persist_R == 2 || persist_R == 4) && persist_NQ == 2 && 2 * persist_grid <= cus;
      const bool leaves_xcd = local_ok && persist_grid <= kXcdCus && 2 * persist_grid > kXcdCus;
      if (fits && half != 0 && (half == 1 || !leaves_xcd)) {
        pc.waves = kPersistWaves / 2;
        persist_grid = (shape.nreg + pc.waves * persist_NQ - 1) / (pc.waves * persist_NQ);
      }
    }
    xcd_local = local_ok && (persist_R == 2 || persist_R == 4) && persist_grid <= kXcdCus;
    // Regions of 2 rows on a grid that runs chip-wide anyway (more than one XCD's worth of workgroups): ONE region per wave, all eight
    // waves of a workgroup at work - the row work per SIMD of the half-occupancy shape (two waves x one region instead of one wave
    // x two) with half its workgroups in the exchange.  Round 5, A/B on one box: 1024 x 256 (config 4) 3.73 -> 3.57 us per iteration,
    // 512^2 3.72 -> 3.60; 256^2 stays on its XCD (2.69 against 3.41).  Option cg_persist_nq: 0 never, 1 wherever the chip holds it.
    {
      constexpr bool kHasNq1 = sizeof(T) == 8 && sizeof(CT) == 4;
      const int nq = opt(OPT_CG_PERSIST_NQ);
      if (kHasNq1 && symmetric && nq != 0 && persist_R == 2 && !a.nx_true && (!xcd_local || nq == 1) &&
          (shape.nreg + kPersistWaves - 1) / kPersistWaves <= cus) {
        persist_NQ = 1; pc.waves = kPersistWaves; xcd_local = false;
        persist_grid = (shape.nreg + kPersistWaves - 1) / kPersistWaves;
      }
    }
    {
      // ... and inside ONE XCD as well, where that needs no more than its 32 workgroups: 256^2 (config 2) 2.74 -> 2.59 us per iteration
      // (eight working waves x one region instead of four x two; 512 x 256 would need 64 workgroups and keeps two regions per wave)
      constexpr bool kHasNq1L = sizeof(T) == 8 && sizeof(CT) == 4 && RECON;
      if (kHasNq1L && opt(OPT_CG_PERSIST_NQ) != 0 && xcd_local && persist_R == 2 && persist_NQ == 2 && !a.nx_true &&
          (shape.nreg + kPersistWaves - 1) / kPersistWaves <= kXcdCus) {
        persist_NQ = 1; pc.waves = kPersistWaves;
        persist_grid = (shape.nreg + kPersistWaves - 1) / kPersistWaves;
      }
    }
    if (persist_R && n < 16384 && force != 1 && !a.nx_true) persist_R = 0;    // tiny grids: two-kernel path (a padded grid is here BECAUSE it is small)
  }
  const bool ragged = a.nx_true != 0;
  if (ragged && persist_R) {
    // the padded-grid variant exists for the common case only: fp64 state, one exchange, exact-float symmetric coefficients
    constexpr bool kRaggedKernel = RECON && sizeof(CT) == 4 && sizeof(T) == 8;
    if (!kRaggedKernel || !symmetric) persist_R = 0;
  }
  if (!persist_R) xcd_local = false;
  const int launch_grid = xcd_local ? kXcds * persist_grid : persist_grid;   // (XCD-local: some XCD is dealt a full group)
  if (persist_R) {
    // the exchanges spin: EVERY workgroup must be resident at the same time.  What the occupancy calculator says one CU can
    // hold (LDS, registers) times the CUs of the device must cover the grid; what it cannot see (another process, a CU mask)
    // is caught by the spin bound -> restart on the two-kernel path (below).
    // (the symmetric variant - S and W streamed, N and E taken from the neighbours' S and W - also serves systems whose diagonal cannot
    // be rebuilt from the off-diagonals: open boundaries, where the diagonal carries the face to the outside - BASELINE config 4.  It
    // then streams the diagonal beside S and W: 16 instead of 24 bytes per cell and pass.  fp64 state only: the fp32 instances keep
    // their registers only with rebuilt diagonals.)
    constexpr bool kCanSymO = sizeof(CT) == 4 && (RECON || sizeof(T) == 8);
    const void* kfn = persist_kernel<T, CT, RECON, false>(persist_R);
    if constexpr (kCanSymO) { if (symmetric) kfn = persist_kernel<T, CT, RECON, true>(persist_R, ragged, persist_NQ); }
    int per_cu = 0, dev = 0, cus = 0;
    PISO_HIP_CHECK(hipGetDevice(&dev));
    PISO_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (!kfn) persist_R = 0;                                 // (a forced 16-row shape for a combination that has no such instance: kHas16)
    else {
      PISO_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, kPersistThreads, 0));
      if ((long long)per_cu * cus < launch_grid) persist_R = 0;
    }
  }
  if (persist_R) {
    pc.rec = reinterpret_cast<unsigned long long*>(persist_ws);
    pc.err = reinterpret_cast<int*>(persist_ws + kPersistWsWords - 16);
    PISO_HIP_CHECK(hipMemsetAsync(persist_ws, 0, kPersistWsWords * sizeof(unsigned), stream));
    pc.xcd = reinterpret_cast<int*>(persist_ws + kPersistRecWords);   // 10 words behind the records, before the error flag
    pc.local_n = xcd_local ? persist_grid : 0;
    if (launch_grid > kPersistMaxGrid) persist_R = 0;
    if (kPersistDiag && opt_on(OPT_CG_PERSIST_TIMING)) {   // diagnostic builds only: per-phase clocks of every workgroup
      PISO_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&pc.timing), 12 * launch_grid * sizeof(unsigned long long)));
      PISO_HIP_CHECK(hipMemsetAsync(pc.timing, 0, 12 * launch_grid * sizeof(unsigned long long), stream));
    }
  }
  auto launch_segment = [&](int kb, int ke) -> int {
    // Tags are unique per launch (a 16-bit launch counter above a 16-bit exchange counter; a segment has < 2^15 exchanges): a
    // record left by an earlier launch - in memory or in some XCD's L2 - can never pass for one of this launch.  The records are
    // zeroed as well, which covers the counter's wrap.
    pc.epoch0 = (g_persist_launches.fetch_add(1, std::memory_order_relaxed) & 0xffffu) << 16;
    PISO_HIP_CHECK(hipMemsetAsync(pc.rec, 0, kPersistZeroBytes, stream));   // records (both levels) + XCD arrivals
    constexpr bool kCanSym = RECON && sizeof(CT) == 4;     // the symmetric variant exists for the compact coefficient path
    if constexpr (kCanSym && sizeof(T) == 8) {
      if (xcd_local) {                                       // (symmetric, one exchange, regions of 2 / 4 rows: checked above)
        if (ragged) {
          if (persist_R == 2) cg_persist1<T, CT, 2, 2, RECON, true, false, true, true><<<launch_grid, kPersistThreads, 0, stream>>>(a, pc, kb, ke, sv, pending ? 1 : 0);
          else cg_persist1<T, CT, 4, 2, RECON, true, false, true, true><<<launch_grid, kPersistThreads, 0, stream>>>(a, pc, kb, ke, sv, pending ? 1 : 0);
        } else {
          if (persist_R == 2 && persist_NQ == 1) cg_persist1<T, CT, 2, 1, RECON, true, false, false, true><<<launch_grid, kPersistThreads, 0, stream>>>(a, pc, kb, ke, sv, pending ? 1 : 0);
          else if (persist_R == 2) cg_persist1<T, CT, 2, 2, RECON, true, false, false, true><<<launch_grid, kPersistThreads, 0, stream>>>(a, pc, kb, ke, sv, pending ? 1 : 0);
          else cg_persist1<T, CT, 4, 2, RECON, true, false, false, true><<<launch_grid, kPersistThreads, 0, stream>>>(a, pc, kb, ke, sv, pending ? 1 : 0);
        }
        PISO_LAUNCH_CHECK();
        return PISO_OK;
      }
      if (ragged) {                                          // padded-grid mode (symmetric, one exchange: checked above)
        if (persist_R == 2) cg_persist1<T, CT, 2, 2, RECON, true, false, true><<<persist_grid, kPersistThreads, 0, stream>>>(a, pc, kb, ke, sv, pending ? 1 : 0);
        else if (persist_R == 4) cg_persist1<T, CT, 4, 2, RECON, true, false, true><<<persist_grid, kPersistThreads, 0, stream>>>(a, pc, kb, ke, sv, pending ? 1 : 0);
        else cg_persist1<T, CT, 16, 1, RECON, true, false, true><<<persist_grid, kPersistThreads, 0, stream>>>(a, pc, kb, ke, sv, pending ? 1 : 0);
        PISO_LAUNCH_CHECK();
        return PISO_OK;
      }
    }
#define PISO_PERSIST_LAUNCH(SYMV)                                                                                            \
    do {                                                                                                                     \
      if constexpr (sizeof(T) == 8 && sizeof(CT) == 4 && SYMV) {                                                             \
        if (persist_R == 2 && persist_NQ == 1) { cg_persist1<T, CT, 2, 1, RECON, SYMV><<<persist_grid, kPersistThreads, 0, stream>>>(a, pc, kb, ke, sv, pending ? 1 : 0); break; } \
      }                                                                                                                      \
      if (persist_R == 2) cg_persist1<T, CT, 2, 2, RECON, SYMV><<<persist_grid, kPersistThreads, 0, stream>>>(a, pc, kb, ke, sv, pending ? 1 : 0);        \
      else if (persist_R == 4) cg_persist1<T, CT, 4, 2, RECON, SYMV><<<persist_grid, kPersistThreads, 0, stream>>>(a, pc, kb, ke, sv, pending ? 1 : 0);   \
      else if constexpr (kHas16<T, CT, RECON, SYMV>) cg_persist1<T, CT, 16, 1, RECON, SYMV><<<persist_grid, kPersistThreads, 0, stream>>>(a, pc, kb, ke, sv, pending ? 1 : 0);  \
    } while (0)
    constexpr bool kCanSymPlain = sizeof(CT) == 4 && (RECON || sizeof(T) == 8);     // (see kCanSymO above)
    if constexpr (kCanSymPlain) {
      if (symmetric) {
        PISO_PERSIST_LAUNCH(true);
        PISO_LAUNCH_CHECK();
        return PISO_OK;
      }
    }
    PISO_PERSIST_LAUNCH(false);
#undef PISO_PERSIST_LAUNCH
    PISO_LAUNCH_CHECK();
    return PISO_OK;
  };
  // ~10 ms of work per segment at 2048^2 (1 000 iterations; one host look per segment - a converged solve leaves its segment by
  // itself).  Measured in the bench: segments of 500 / 1 000 / 2 000 iterations 4.41 / 4.44 / 4.46 steps/s - every launch pays its
  // prologue, the state's trip from and to memory and a cold first iteration
  int seg_len = (int)(40000.0 / ((double)n * 8.5e-6 + 4.0));
  seg_len = seg_len < 50 ? 50 : (seg_len > 2000 ? 2000 : seg_len);
  if (opt(OPT_CG_SEGMENT) > 0) seg_len = opt(OPT_CG_SEGMENT);
  hipEvent_t* seg_ev = tl_poll.seg_ev;
  if (persist_R && prof && !seg_ev[0]) { PISO_HIP_CHECK(hipEventCreate(&seg_ev[0])); PISO_HIP_CHECK(hipEventCreate(&seg_ev[1])); }
  double seg_ms = 0; long long seg_iters = 0, seg_launches = 0;
  int segments_run = 0, unsynced = 0;
  for (int k = 0; k < total && !finished; ++k) {
    const bool is_reset = !fixed && ((k + 1) % reset == 0);
    if (persist_R && k > 0 && !is_reset) {
      // run NORMAL iterations [k, ke) in one launch: up to the next reset iteration / the end / one segment length
      int ke = total;
      if (!fixed) { const int next_reset = ((k + 1 + reset - 1) / reset) * reset - 1; if (next_reset < ke) ke = next_reset; }
      if (ke > k + seg_len) ke = k + seg_len;
      if (ke > k) {
        if (prof) PISO_HIP_CHECK(hipEventRecord(seg_ev[0], stream));
        { const int rc = launch_segment(k, ke); if (rc != PISO_OK) return rc; }
        if (prof) PISO_HIP_CHECK(hipEventRecord(seg_ev[1], stream));
        // Short segments (frequent residual resets: the reference's default residual_reset = 10 leaves 9 iterations between two
        // resets) are not worth a host round trip each: the host looks again after ~250 iterations.  Everything queued behind a
        // converged or failed segment returns at once (every kernel checks the state record first), the error flag is sticky.
        if (!prof && ke - k <= 32 && unsynced + (ke - k) <= 256 && ke < total) {
          unsynced += ke - k;
          ++segments_run;
          k_last = ke - 1;
          pending = false;
          k = ke - 1;
          continue;
        }
        unsynced = 0;
        PISO_HIP_CHECK(hipMemcpyAsync(&tl_poll.pinned[0], &a.state[0], sizeof(CgState), hipMemcpyDeviceToHost, stream));
        int herr = 0;
        PISO_HIP_CHECK(hipMemcpyAsync(&herr, pc.err, sizeof(int), hipMemcpyDeviceToHost, stream));
        PISO_HIP_CHECK(hipStreamSynchronize(stream));
        if (herr) {
          // A grid-wide exchange gave up: some workgroups were not resident (another kernel or process holds CUs).  The
          // segment's state is unusable; the two-kernel path needs no co-residency: run the whole solve again on it.
          ++g_persist_fallbacks;
          if (xcd_local) g_xcd_local_failed = true;
          if (pc.timing) { PISO_HIP_CHECK(hipFree(pc.timing)); pc.timing = nullptr; }
          return cg_run<T, CT, V, RECON>(a, persist_ws, symmetric, accuracy, max_iterations, rank_deficient, reset, fixed, iterations_out,
                                         kernel_ms_out, stream, false);
        }
        if (prof) {
          // (a solve that converges inside the launch leaves it there: the iterations it RAN count, not the segment's length)
          const int ran = tl_poll.pinned[0].done ? (tl_poll.pinned[0].iterations - k > 0 ? tl_poll.pinned[0].iterations - k : 0) : ke - k;
          float t = 0; PISO_HIP_CHECK(hipEventElapsedTime(&t, seg_ev[0], seg_ev[1])); seg_ms += t; seg_iters += ran < ke - k ? ran : ke - k; ++seg_launches;
        }
        if (tl_poll.pinned[0].done) { finished = true; stop_it = tl_poll.pinned[0].iterations; }
        ++segments_run;
        k_last = ke - 1;
        pending = false;                                   // the segment applies every x += alpha p itself
        k = ke - 1;                                        // the loop increment moves to ke
        continue;
      }
    }
    const bool sample = prof && (k % prof_stride == prof_stride - 1) && ep.used[0] < EventPool::kMax && !is_reset && k > 0;
    if (is_reset) {
      if (pending) { cg_flush_x<T><<<gflat, kBlock, 0, stream>>>(a, k - 1, sv); pending = false; }
      cg_k1<T, CT, V, RECON><<<g1, kBlock, 0, stream>>>(a, k, MODE_RESET, sv, k > 0 ? 1 : 0, 0);
      ++sv;
      cg_reset_residual<T><<<gflat, kBlock, 0, stream>>>(a, sv);
      cg_k1<T, CT, V, RECON><<<g1, kBlock, 0, stream>>>(a, k, MODE_INIT, sv, 0, 0);
    } else if (k == 0) {
      cg_k1<T, CT, V, RECON><<<g1, kBlock, 0, stream>>>(a, k, MODE_INIT, sv, 0, 0);
    } else {
      if (sample) PISO_HIP_CHECK(hipEventRecord(ep.start[0][ep.used[0]], stream));
      cg_k1<T, CT, V, RECON><<<g1, kBlock, 0, stream>>>(a, k, MODE_NORMAL, sv, 1, pending ? 1 : 0);
      if (sample) PISO_HIP_CHECK(hipEventRecord(ep.stop[0][ep.used[0]++], stream));
      ++sv;
    }
    if (sample) PISO_HIP_CHECK(hipEventRecord(ep.start[1][ep.used[1]], stream));
    cg_k2<T, V><<<g2, kBlock, 0, stream>>>(a, k, sv);
    if (sample) PISO_HIP_CHECK(hipEventRecord(ep.stop[1][ep.used[1]++], stream));
    PISO_LAUNCH_CHECK();
    pending = true;
    k_last = k;
    if (!fixed && ((k + 1) % batch == 0) && k + 1 < total) {
      const int slot = polls & 1;
      PISO_HIP_CHECK(hipMemcpyAsync(&tl_poll.pinned[slot], &a.state[sv & 1], sizeof(CgState), hipMemcpyDeviceToHost, stream));
      PISO_HIP_CHECK(hipEventRecord(tl_poll.ev[slot], stream));
      if (polls > 0) {                                   // look at the PREVIOUS poll while this batch is already queued
        const int r = inspect((polls - 1) & 1);
        if (r < 0) return PISO_ERR_HIP;
        if (r > 0) finished = true;
      }
      ++polls;
    }
  }
  // the direction of the last executed iteration (a converged solve was already flushed by the K1 that detected it)
  if (pending && k_last >= 0) cg_flush_x<T><<<gflat, kBlock, 0, stream>>>(a, k_last, sv);
  // Final look.  (A success of the test that belongs to the very last iteration is not evaluated: the reference would
  // report iterations == total for it, which is what an unfinished loop reports as well.)
  if (!fixed && !finished) {
    const int slot = polls & 1;
    PISO_HIP_CHECK(hipMemcpyAsync(&tl_poll.pinned[slot], &a.state[sv & 1], sizeof(CgState), hipMemcpyDeviceToHost, stream));
    PISO_HIP_CHECK(hipEventRecord(tl_poll.ev[slot], stream));
    const int r = inspect(slot);
    if (r < 0) return PISO_ERR_HIP;
    if (r > 0) finished = true;
  }
  PISO_HIP_CHECK(hipStreamSynchronize(stream));
  if (opt(OPT_CG_XCD_MAP) == 1) {
    tl_xcd_map_n = 0;
    if (segments_run > 0 && persist_R && !xcd_local && persist_grid <= kPersistMaxGrid) {
      PISO_HIP_CHECK(hipMemcpy(tl_xcd_map, pc.xcd + kPersistXcdTable, (size_t)persist_grid * sizeof(int), hipMemcpyDeviceToHost));
      tl_xcd_map_n = persist_grid;
    }
  }
  if (segments_run > 0 && allow_persist) {                 // (segments whose host look was deferred: did one of them give up?)
    int herr = 0;
    PISO_HIP_CHECK(hipMemcpy(&herr, pc.err, sizeof(int), hipMemcpyDeviceToHost));
    if (herr) {
      ++g_persist_fallbacks;
      if (pc.timing) { PISO_HIP_CHECK(hipFree(pc.timing)); pc.timing = nullptr; }
      return cg_run<T, CT, V, RECON>(a, persist_ws, symmetric, accuracy, max_iterations, rank_deficient, reset, fixed, iterations_out,
                                     kernel_ms_out, stream, false);
    }
  }
  // ---- The persistent kernel lets workgroups read what others published without release / acquire fences (cg_persist1.h).  That
  // is checked here at run time instead of being trusted: r - the CG recurrence - must still equal b - A^ x for the x the solve
  // returns (to eps * condition * |b|; a stale perimeter value would leave an O(alpha |z'|) gap that nothing removes before the
  // next residual reset).  One stencil pass per solve; a failure restarts the solve on the two-kernel path and is counted.
  if (segments_run > 0 && sizeof(T) == 8 && !fixed && opt(OPT_CG_VERIFY) != 0) {
    unsigned* out2 = reinterpret_cast<unsigned*>(pc.err) + 4;
    PISO_HIP_CHECK(hipMemsetAsync(out2, 0, 2 * sizeof(unsigned), stream));
    const int gvf = grid_for((long long)n, kBlock * 4, 1024);
    cg_verify_sum_x<T><<<gvf, kBlock, 0, stream>>>(a, a.partsA);
    cg_verify_gap<T, CT><<<gvf, kBlock, 0, stream>>>(a, a.partsA, gvf, out2);
    PISO_LAUNCH_CHECK();
    unsigned h2[2] = {0, 0};
    PISO_HIP_CHECK(hipMemcpyAsync(h2, out2, sizeof(h2), hipMemcpyDeviceToHost, stream));
    PISO_HIP_CHECK(hipStreamSynchronize(stream));
    float gap, scale;
    memcpy(&gap, &h2[0], 4); memcpy(&scale, &h2[1], 4);
    ++g_verify_runs;
    if ((gap > 1e-5f * scale && gap > 1e-30f) || opt(OPT_CG_VERIFY) == 2) {     // (2: test knob - treat the check as failed)
      ++g_verify_failures; ++g_persist_fallbacks;
      if (pc.timing) { PISO_HIP_CHECK(hipFree(pc.timing)); pc.timing = nullptr; }
      return cg_run<T, CT, V, RECON>(a, persist_ws, symmetric, accuracy, max_iterations, rank_deficient, reset, fixed, iterations_out,
                                     kernel_ms_out, stream, false);
    }
  }
  if (pc.timing) {
    std::vector<unsigned long long> h(12 * persist_grid);
    PISO_HIP_CHECK(hipMemcpy(h.data(), pc.timing, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    PISO_HIP_CHECK(hipFree(pc.timing));
    const char* names[9] = {"D (p update, stencil, sums, publish)", "exchange", "U (stencil, x / r update, ring)", "-", "-",
                             "  exchange: wave sums + drain of the perimeter stores", "  exchange: first barrier", "  exchange: publish + polling",
                             "  exchange: record sums + second barrier"};
    if (opt(OPT_CG_PERSIST_TIMING) >= 2) {                  // the whole table: one line per workgroup (us per iteration)
      const double f = 0.01 / (double)(k_last > 0 ? k_last : 1);
      for (int b = 0; b < persist_grid; ++b)
        fprintf(stderr, "cg_persist_wg %3d xcd %d band %3d  D %.2f  exchange %.2f  U %.2f  | drain %.2f  barrier1 %.2f  publish+poll %.2f  sums %.2f\n", b, (int)h[9 * persist_grid + b], (int)h[10 * persist_grid + b],
                f * (double)h[0 * persist_grid + b], f * (double)h[1 * persist_grid + b], f * (double)h[2 * persist_grid + b],
                f * (double)h[5 * persist_grid + b], f * (double)h[6 * persist_grid + b], f * (double)h[7 * persist_grid + b], f * (double)h[8 * persist_grid + b]);
    }
    for (int q = 0; q < 9; ++q) {
      double s = 0, mn = 1e300, mx = 0;
      for (int b = 0; b < persist_grid; ++b) { const double v = (double)h[q * persist_grid + b]; s += v; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
      fprintf(stderr, "cg_persist %s: avg %.2f us/iter  min %.2f  max %.2f\n", names[q], 0.01 * s / persist_grid / (double)(k_last > 0 ? k_last : 1),
              0.01 * mn / (double)(k_last > 0 ? k_last : 1), 0.01 * mx / (double)(k_last > 0 ? k_last : 1));
    }
  }
  if (iterations_out) *iterations_out = finished ? stop_it : total;
  if (prof) {
    double ms[2] = {0, 0};
    for (int q = 0; q < 2; ++q)
      for (int i = 0; i < ep.used[q]; ++i) {
        float t = 0;
        PISO_HIP_CHECK(hipEventElapsedTime(&t, ep.start[q][i], ep.stop[q][i]));
        ms[q] += t;
      }
    if (kernel_ms_out) {
      kernel_ms_out[0] = ep.used[0] ? (float)(ms[0] / ep.used[0]) : 0.f;
      kernel_ms_out[1] = ep.used[1] ? (float)(ms[1] / ep.used[1]) : 0.f;
    }
    if (g_prof.enabled) {
      for (int q = 0; q < 2; ++q) { g_prof.ms[q] += ms[q]; g_prof.count[q] += ep.used[q]; }
      g_prof.ms[2] += seg_ms; g_prof.count[2] += seg_iters; g_prof.count[3] += seg_launches;
    }
    if (kernel_ms_out && seg_iters > 0) { kernel_ms_out[0] = (float)(seg_ms / seg_iters); kernel_ms_out[1] = 0.f; }
  }
  return PISO_OK;
}

// rows of nx elements between arrays of different leading dimensions (padded-grid mode: b in, x out)
template <typename T>
__global__ __launch_bounds__(kBlock) void cg_copy_rows(const T* __restrict__ src, T* __restrict__ dst, int nx, int ny, int ld_src, int ld_dst) {
  const size_t n = (size_t)nx * ny;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const size_t j = i / (size_t)nx, c = i % (size_t)nx;
    dst[j * (size_t)ld_dst + c] = src[j * (size_t)ld_src + c];
  }
}

template <typename T>
static int cg_solve(int nx, int ny, int per_x, int per_y, const T* L, const T* b, T* x_out, float accuracy,
                    int max_iterations, int rank_deficient, int reset, int fixed, int* iterations_out,
                    float* kernel_ms_out, void* ws, size_t ws_bytes, piso_stream_t stream_, int* iterations_dev = nullptr) {
  if (nx < 1 || ny < 1 || !L || !b || !x_out || !ws || max_iterations < 0 || reset < 1) {
    set_error_msg("piso_cg_solve: invalid argument");
    return PISO_ERR_INVALID_ARG;
  }
  if (ws_bytes < cg_workspace_bytes<T>(nx, ny)) {
    set_error_msg("piso_cg_solve: workspace too small");
    return PISO_ERR_INVALID_ARG;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const size_t n_true = (size_t)nx * ny;
  if (n_true <= (size_t)kTinyMaxCells && opt(OPT_CG_TINY) != 0 && opt(OPT_CG_PERSIST) < 0) {   // (a forced / forbidden persistent path: tests)
    // a tiny grid (the lid-driven cavity): the whole solve in ONE workgroup, one launch (cg_tiny.h)
    CgState* st_dev = reinterpret_cast<CgState*>(ws);
    const int total = fixed ? fixed : max_iterations;
    hipEvent_t* ev = nullptr;
    if (kernel_ms_out) {
      { const int rc = ensure_poll(); if (rc != PISO_OK) return rc; }
      ev = tl_poll.seg_ev;
      if (!ev[0]) { PISO_HIP_CHECK(hipEventCreate(&ev[0])); PISO_HIP_CHECK(hipEventCreate(&ev[1])); }
      PISO_HIP_CHECK(hipEventRecord(ev[0], stream));
    }
    const bool async = iterations_dev != nullptr;         // the iteration count stays on the device: nothing waits for the solve
    const bool cols = nx <= 64 && ny <= kColsMaxNy && (!per_x || nx == 64) && opt(OPT_CG_TINY) != 2;    // (cg_tiny = 2: the general kernel, tests)
    if (cols && per_x)
      cg_tiny_cols<T, true><<<1, kTinyThreads, 0, stream>>>(L, b, x_out, nx, ny, per_y, fixed ? -1.0f : accuracy, total, fixed ? 0 : reset,
                                                            rank_deficient, st_dev, iterations_dev);
    else if (cols)
      cg_tiny_cols<T, false><<<1, kTinyThreads, 0, stream>>>(L, b, x_out, nx, ny, per_y, fixed ? -1.0f : accuracy, total, fixed ? 0 : reset,
                                                             rank_deficient, st_dev, iterations_dev);
    else
      cg_tiny<T><<<1, kTinyThreads, 0, stream>>>(L, b, x_out, nx, ny, per_x, per_y, fixed ? -1.0f : accuracy, total, fixed ? 0 : reset,
                                                 rank_deficient, st_dev, iterations_dev);
    PISO_LAUNCH_CHECK();
    ++g_tiny_solves;
    if (async) return PISO_OK;
    if (ev) PISO_HIP_CHECK(hipEventRecord(ev[1], stream));
    CgState hst;
    PISO_HIP_CHECK(hipMemcpyAsync(&hst, st_dev, sizeof(CgState), hipMemcpyDeviceToHost, stream));
    PISO_HIP_CHECK(hipStreamSynchronize(stream));
    if (iterations_out) *iterations_out = (!fixed && hst.done) ? hst.iterations : total;
    if (ev) { float ms = 0; PISO_HIP_CHECK(hipEventElapsedTime(&ms, ev[0], ev[1])); kernel_ms_out[0] = total > 0 ? ms / (float)total : 0.f; kernel_ms_out[1] = 0.f; }
    return PISO_OK;
  }
  if (iterations_dev) {
    set_error_msg("piso_cg_solve_async: this grid is solved with the host in the loop (segments, stopping test): call piso_cg_solve");
    return PISO_ERR_NEEDS_HOST;
  }
  int nxp = nx, nyp = ny;
  const bool padded = opt(OPT_CG_PERSIST) != 0 && opt(OPT_CG_PAD) != 0 &&
                      padded_dims(nx, ny, per_x, per_y, (int)sizeof(T), &nxp, &nyp);      // (see padded_dims)
  const size_t n = (size_t)nxp * nyp;
  Arena ar(ws, ws_bytes);
  CgArgs<T> a;
  T* cC = ar.take<T>(n);
  T* oT = ar.take<T>(4 * n);
  float* oF = ar.take<float>(4 * n);
  int* flags = ar.take<int>(4);
  a.cC = cC;
  a.b = b; a.x = x_out;
  T *b_pad = nullptr, *x_pad = nullptr;
  if (padded) { b_pad = ar.take<T>(n); x_pad = ar.take<T>(n); a.b = b_pad; a.x = x_pad; }
  a.r = ar.take<T>(n); a.z = ar.take<T>(n); a.p[0] = ar.take<T>(n); a.p[1] = ar.take<T>(n);
  a.zp[0] = ar.take<T>(n); a.zp[1] = ar.take<T>(n);
  a.partsA = ar.take<T>(3 * kMaxPartials); a.partsB = ar.take<T>(3 * kMaxPartials); a.partsS = ar.take<T>(kMaxPartials);
  a.scal = ar.take<T>(SC_COUNT);
  a.state = ar.take<CgState>(2);
  unsigned* persist_ws = ar.take<unsigned>(kPersistWsWords);
  a.nx = nxp; a.ny = nyp; a.per_x = per_x; a.per_y = per_y;
  a.nx_true = padded ? nx : 0; a.ny_true = padded ? ny : 0; a.ncells = padded ? (double)n_true : 0.0;
  a.ntx = a.nty = a.rows_per_wave = 0; a.nA = a.nB = 0; a.accuracy = accuracy;
  a.gA = nullptr; a.gB = nullptr;
  a.nt = 0;
  if (opt(OPT_CG_NT) > 0) a.nt = opt(OPT_CG_NT);
  if (!ar.ok()) { set_error_msg("piso_cg_solve: workspace too small"); return PISO_ERR_INVALID_ARG; }

  PISO_HIP_CHECK(hipMemsetAsync(flags, 0, 4 * sizeof(int), stream));
  cg_zero_partials<T><<<(3 * kMaxPartials + 255) / 256, 256, 0, stream>>>(a.partsA, a.partsB, a.partsS);
  const int gs = grid_for((long long)n_true, kBlock * 4);
  if (padded) {                                             // zero coefficients and a zero right-hand side keep the padding at zero
    PISO_HIP_CHECK(hipMemsetAsync(cC, 0, n * sizeof(T), stream));
    PISO_HIP_CHECK(hipMemsetAsync(oT, 0, 4 * n * sizeof(T), stream));
    PISO_HIP_CHECK(hipMemsetAsync(oF, 0, 4 * n * sizeof(float), stream));
    PISO_HIP_CHECK(hipMemsetAsync(b_pad, 0, n * sizeof(T), stream));
    cg_copy_rows<T><<<gs, kBlock, 0, stream>>>(b, b_pad, nx, ny, nx, nxp);
  }
  cg_setup_coeffs<T><<<gs, kBlock, 0, stream>>>(L, cC, oT, oF, a.partsS, flags, n_true, nx, ny, per_x, per_y, padded ? nxp : 0, padded ? n : 0);
  PISO_LAUNCH_CHECK();
  // The off-diagonals of the PISO pressure matrix are float32 face coefficients (laplace_op.cu.cc:140-177): stored as
  // float they are exact and K1 reads 24 instead of 40 coefficient bytes per cell.  Any other input keeps them in T.
  int hflags[3] = {1, 1, 1};
  {
    PISO_HIP_CHECK(hipMemcpyAsync(hflags, flags, 3 * sizeof(int), hipMemcpyDeviceToHost, stream));
    PISO_HIP_CHECK(hipStreamSynchronize(stream));
  }
  if (opt_on(OPT_CG_NO_COMPACT)) hflags[0] = hflags[1] = 1;         // tuning / test knob: plain T coefficients
  if (opt_on(OPT_CG_NO_RECON)) hflags[1] = 1;
  if (opt_on(OPT_CG_NO_SYM)) hflags[2] = 1;
  const bool symmetric = !hflags[2];
  constexpr int VMID = 16 / sizeof(T);
  const bool aligned = ((reinterpret_cast<uintptr_t>(a.b) | reinterpret_cast<uintptr_t>(a.x)) & 15) == 0;
  const bool vec = aligned && (nxp % VMID == 0);
  int rc = PISO_OK;
#define PISO_CG_RUN(CT, V, RECON) \
  rc = cg_run<T, CT, V, RECON>(a, persist_ws, symmetric, accuracy, max_iterations, rank_deficient, reset, fixed, iterations_out, kernel_ms_out, stream)
  if (!hflags[0]) {                                         // (fp32 state: trivially exact - the same path, so that the diagonal can be rebuilt there too)
    a.oS = oF; a.oW = oF + n; a.oE = oF + 2 * n; a.oN = oF + 3 * n;
    if (!hflags[1]) { if (vec) PISO_CG_RUN(float, VMID, true); else PISO_CG_RUN(float, 1, true); }
    else if (vec) PISO_CG_RUN(float, VMID, false);
    else PISO_CG_RUN(float, 1, false);
  } else {
    a.oS = oT; a.oW = oT + n; a.oE = oT + 2 * n; a.oN = oT + 3 * n;
    if (vec) PISO_CG_RUN(T, VMID, false);
    else PISO_CG_RUN(T, 1, false);
  }
#undef PISO_CG_RUN
  if (rc != PISO_OK) return rc;
  if (padded) {                                             // (cg_run has synchronised the stream: x_pad is final)
    cg_copy_rows<T><<<gs, kBlock, 0, stream>>>(x_pad, x_out, nx, ny, nxp, nx);
    PISO_LAUNCH_CHECK();
    PISO_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return PISO_OK;
}

}  // namespace piso

using namespace piso;

extern "C" {

size_t piso_cg_workspace_bytes(int nx, int ny, int elem_size) {
  return elem_size == 8 ? cg_workspace_bytes<double>(nx, ny) : cg_workspace_bytes<float>(nx, ny);
}

int piso_cg_solve_f64(int nx, int ny, int periodic_x, int periodic_y, const double* laplace, const double* divergence,
                      double* x_out, float accuracy, int max_iterations, int rank_deficient, int residual_reset,
                      int* iterations_out, void* workspace, size_t workspace_bytes, piso_stream_t stream) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  return cg_solve<double>(nx, ny, periodic_x, periodic_y, laplace, divergence, x_out, accuracy, max_iterations,
                          rank_deficient, residual_reset, 0, iterations_out, nullptr, workspace, workspace_bytes, stream);
}

int piso_cg_solve_f32(int nx, int ny, int periodic_x, int periodic_y, const float* laplace, const float* divergence,
                      float* x_out, float accuracy, int max_iterations, int rank_deficient, int residual_reset,
                      int* iterations_out, void* workspace, size_t workspace_bytes, piso_stream_t stream) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  return cg_solve<float>(nx, ny, periodic_x, periodic_y, laplace, divergence, x_out, accuracy, max_iterations,
                         rank_deficient, residual_reset, 0, iterations_out, nullptr, workspace, workspace_bytes, stream);
}

int piso_cg_solve_async_f64(int nx, int ny, int periodic_x, int periodic_y, const double* laplace, const double* divergence,
                            double* x_out, float accuracy, int max_iterations, int rank_deficient, int residual_reset,
                            int* iterations_dev, void* workspace, size_t workspace_bytes, piso_stream_t stream) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  if (!iterations_dev) { set_error_msg("piso_cg_solve_async: iterations_dev is NULL"); return PISO_ERR_INVALID_ARG; }
  return cg_solve<double>(nx, ny, periodic_x, periodic_y, laplace, divergence, x_out, accuracy, max_iterations,
                          rank_deficient, residual_reset, 0, nullptr, nullptr, workspace, workspace_bytes, stream, iterations_dev);
}

int piso_cg_solve_async_f32(int nx, int ny, int periodic_x, int periodic_y, const float* laplace, const float* divergence,
                            float* x_out, float accuracy, int max_iterations, int rank_deficient, int residual_reset,
                            int* iterations_dev, void* workspace, size_t workspace_bytes, piso_stream_t stream) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  if (!iterations_dev) { set_error_msg("piso_cg_solve_async: iterations_dev is NULL"); return PISO_ERR_INVALID_ARG; }
  return cg_solve<float>(nx, ny, periodic_x, periodic_y, laplace, divergence, x_out, accuracy, max_iterations,
                         rank_deficient, residual_reset, 0, nullptr, nullptr, workspace, workspace_bytes, stream, iterations_dev);
}

int piso_cg_fixed_iterations_f64(int nx, int ny, int periodic_x, int periodic_y, const double* laplace,
                                 const double* divergence, double* x_out, int rank_deficient, int iterations,
                                 float* kernel_ms_out, void* workspace, size_t workspace_bytes, piso_stream_t stream) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  if (iterations < 1) { set_error_msg("piso_cg_fixed_iterations: iterations < 1"); return PISO_ERR_INVALID_ARG; }
  return cg_solve<double>(nx, ny, periodic_x, periodic_y, laplace, divergence, x_out, 0.f, iterations, rank_deficient,
                          1 << 30, iterations, nullptr, kernel_ms_out, workspace, workspace_bytes, stream);
}

void piso_cg_profile_enable(int enable, int stride) {
  g_prof.enabled = enable;
  if (stride > 0) g_prof.stride = stride;
  for (int q = 0; q < 4; ++q) { g_prof.ms[q] = 0; g_prof.count[q] = 0; }
}

int piso_cg_persist_fallbacks(void) { return g_persist_fallbacks; }
int piso_cg_last_xcd_map(int* out, int capacity) {
  const int n = tl_xcd_map_n < capacity ? tl_xcd_map_n : capacity;
  for (int i = 0; i < n; ++i) out[i] = tl_xcd_map[i];
  return tl_xcd_map_n;
}
long long piso_cg_tiny_solves(void) { return g_tiny_solves; }
void piso_cg_verify_stats(long long* runs_out, int* failures_out) {
  if (runs_out) *runs_out = g_verify_runs;
  if (failures_out) *failures_out = g_verify_failures;
}

void piso_cg_profile_read(double* ms_sum, long long* count) {
  for (int q = 0; q < 4; ++q) { ms_sum[q] = g_prof.ms[q]; count[q] = g_prof.count[q]; }
}

}  // extern "C"
