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

namespace oat {
// One wave that sleeps for `ticks` of the 100 MHz real-time counter: a stream-ordered delay (see oat_delay).
__global__ void delay_kernel(int ticks) {
  const unsigned long long t0 = wall_clock64();
  while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
}
}  // namespace oat

extern "C" int oat_delay(int nanoseconds, void* stream) {
  if (nanoseconds <= 0) return 0;
  OAT_LAUNCH(oat::delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (nanoseconds + 9) / 10);
  return oat::check_launch("delay");
}

extern "C" const char* oat_last_error(void) { return oat::g_err; }
extern "C" int oat_abi_version(void) { return 1; }
