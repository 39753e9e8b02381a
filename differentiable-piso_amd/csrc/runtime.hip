// Library bookkeeping: version, last-error text, device probing.
#include "piso_common.h"

namespace piso {
static thread_local char g_err[512] = "";
void set_error(const char* what, hipError_t err) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(err));
}
void set_error_msg(const char* what) { snprintf(g_err, sizeof(g_err), "%s", what); }
}  // namespace piso

extern "C" {
const char* piso_version(void) { return "libpiso_hip 0.1 (gfx950)"; }
const char* piso_last_error_string(void) { return piso::g_err; }
int piso_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
}
