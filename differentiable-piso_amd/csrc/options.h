// Tuning / test knobs of libpiso_hip.so.  Every knob has a default taken ONCE, when the library is loaded, from the
// environment variable PISO_<NAME> (upper case); afterwards only piso_set_option() changes it.  -1 = "not set / automatic".
// None of them changes WHAT is computed: they pick between implementations that return bitwise the same result (kernel instance, staging,
// launch shape) or switch a check / a measurement aid on and off.  Stores and loads are atomic; a call works on a snapshot (OptScope).
#pragma once

namespace piso {

enum Opt {
  OPT_CG_PERSIST = 0,      // 0 forbid / 1 force the persistent CG kernel (default: by grid size)
  OPT_CG_PERSIST_R,        // rows per region of the persistent kernel: 2 | 4 | 16
  OPT_CG_PERSIST_HALF,     // 0: never run small regions with ONE working wave per SIMD and twice the workgroups (default: where the chip holds them)
  OPT_CG_PERSIST_NQ,       // regions of 2 rows, ONE per wave instead of two: 0 never, 1 wherever the chip holds them (default: grids that leave one XCD anyway, and one-XCD grids of at most 256 regions)
  OPT_CG_SEGMENT,          // CG iterations per persistent launch
  OPT_CG_PERSIST_TIMING,   // per-phase clocks of the persistent kernel (diagnostic builds only)
  OPT_CG_RPW,              // two-kernel path: rows per wave of K1
  OPT_CG_MAXBLOCKS,        // two-kernel path: grid cap of K1
  OPT_CG_NT,               // two-kernel path: non-temporal access bits
  OPT_CG_NO_COMPACT,       // keep the off-diagonals in T (no exact-float32 compression)
  OPT_CG_NO_RECON,         // read the diagonal instead of recomputing it
  OPT_CG_NO_SYM,           // stream all four off-diagonal arrays even if the matrix is symmetric
  OPT_CG_VERIFY,           // 0: skip the true-residual check of solves that used the persistent kernel (default: on)
  OPT_CG_PAD,              // 0: never embed a small wall-bounded grid in a padded one for the persistent kernel (default: on)
  OPT_CG_XCD_LOCAL,        // 0: never run a small grid's persistent solve on the workgroups of ONE XCD (default: on)
  OPT_CG_TINY,             // 0: never solve a tiny grid (<= 4 608 cells) inside one workgroup (cg_tiny.h; default: on)
  OPT_CG_XCD_MAP,          // 1 (tests): keep the XCD of every workgroup of a solve's last chip-wide persistent launch for piso_cg_last_xcd_map
  OPT_CONV_LDS,            // 0: the closure's forward / input-gradient convolutions read their operands straight from L2 (default: staged through LDS)
  OPT_BICG_FOLD,           // 0: the BiCGStab scalar stages always run as launches of their own (default: folded into their consumers on one GPU)
  OPT_BICG_SWEEP_LDS,      // 0: the triangular sweeps address memory in scan order (bi_sweep) instead of staging rows through LDS (bi_sweep_lds)
  OPT_BICG_FUSE_P,         // 0: the direction update of BiCGStab runs as a launch of its own (bi_update_p) instead of inside the forward sweep that reads it
  OPT_SLAB_FORCE,          // 1: a communicator of ONE rank still runs the slab code paths (ring of one: halo messages and sums to itself; tests)
  OPT_SLAB_HOP_TICKS,      // measurements only: the persistent slab kernel's cross-GPU records leave this many 10 ns ticks late (an emulated link latency)
  OPT_COUNT
};

int opt(Opt o);                       // value of the calling entry point's snapshot (inside an OptScope), else the live value (-1 = not set)
inline bool opt_on(Opt o) { return opt(o) > 0; }

// Every public entry point that reads knobs opens an OptScope first: ALL knobs are copied once (atomically, one by one) into a
// thread-local snapshot and `opt()` answers from that copy until the scope closes.  A piso_set_option() from another thread - a test
// flipping a knob while autograd's backward thread is inside a solve - therefore never changes a decision in the MIDDLE of a call
// (which kernel instance runs, whether the result is verified): a call sees the knobs as they were when it started.  Re-entrant.
struct OptScope {
  OptScope();
  ~OptScope();
  OptScope(const OptScope&) = delete;
  OptScope& operator=(const OptScope&) = delete;
};

}  // namespace piso
