// Library bookkeeping: version, last-error text, device probing.
#include "piso_common.h"
#include "options.h"
#include <stdlib.h>
#include <atomic>

namespace piso {
static thread_local char g_err[512] = "";
void set_error(const char* what, hipError_t err) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(err));
}
void set_error_msg(const char* what) { snprintf(g_err, sizeof(g_err), "%s", what); }

static const char* const kOptNames[OPT_COUNT] = {"cg_persist", "cg_persist_r", "cg_persist_half", "cg_persist_nq", "cg_segment", "cg_persist_timing",
                                                 "cg_rpw", "cg_maxblocks", "cg_nt", "cg_no_compact", "cg_no_recon", "cg_no_sym", "cg_verify", "cg_pad", "cg_xcd_local", "cg_tiny", "cg_xcd_map", "conv_lds", "bicg_fold", "bicg_sweep_lds", "bicg_fuse_p", "slab_force", "slab_hop_ticks"};
struct Options {
  std::atomic<int> v[OPT_COUNT];
  Options() {                                      // the environment is read here, once, and never again
    for (int i = 0; i < OPT_COUNT; ++i) {
      char env[64] = "PISO_";
      size_t k = 5;
      for (const char* c = kOptNames[i]; *c && k + 1 < sizeof(env); ++c) env[k++] = (char)((*c >= 'a' && *c <= 'z') ? *c - 32 : *c);
      env[k] = 0;
      const char* e = getenv(env);
      v[i].store((e && *e) ? atoi(e) : -1, std::memory_order_relaxed);
    }
    if (getenv("PISO_CG_NO_PERSIST")) v[OPT_CG_PERSIST].store(0, std::memory_order_relaxed);
  }
};
static Options g_opt;
static thread_local int tl_snapshot[OPT_COUNT];
static thread_local int tl_scope_depth = 0;
int opt(Opt o) { return tl_scope_depth > 0 ? tl_snapshot[o] : g_opt.v[o].load(std::memory_order_relaxed); }
OptScope::OptScope() {
  if (tl_scope_depth++ == 0)
    for (int i = 0; i < OPT_COUNT; ++i) tl_snapshot[i] = g_opt.v[i].load(std::memory_order_relaxed);
}
OptScope::~OptScope() { --tl_scope_depth; }
static int opt_index(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, kOptNames[i]) == 0) return i;
  return -1;
}
}  // namespace piso

extern "C" {
const char* piso_version(void) { return "libpiso_hip 0.1 (gfx950)"; }
const char* piso_last_error_string(void) { return piso::g_err; }
int piso_set_option(const char* name, int value) {
  const int i = piso::opt_index(name);
  if (i < 0) { piso::set_error_msg("piso_set_option: unknown option"); return PISO_ERR_INVALID_ARG; }
  piso::g_opt.v[i].store(value, std::memory_order_relaxed);
  return PISO_OK;
}
int piso_get_option(const char* name, int* value_out) {
  const int i = piso::opt_index(name);
  if (i < 0 || !value_out) { piso::set_error_msg("piso_get_option: unknown option"); return PISO_ERR_INVALID_ARG; }
  *value_out = piso::g_opt.v[i].load(std::memory_order_relaxed);
  return PISO_OK;
}
int piso_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
}
