// Error plumbing shared by every entry point of liboatrans_hip.so.
// Convention (include/oatrans_hip.h): every oat_* function returns 0 on success, a negative
// code on error, and leaves a human-readable message for oat_last_error() (thread-local).
// The library never allocates, frees or synchronises: all buffers (incl. workspaces) are
// caller-owned, every launch is stream-ordered on the stream handed in.
#include "common.h"
#include <string.h>
#include <stdio.h>

namespace oat {
static thread_local char g_err[512] = "";
void set_error(const char* msg) {
  strncpy(g_err, msg, sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return -100;
}
}  // namespace oat

extern "C" const char* oat_last_error(void) { return oat::g_err; }
extern "C" int oat_abi_version(void) { return 2; }   // 2 (round 6): per-call `tune` / `grid` arguments, no oat_*_set_* entry points
