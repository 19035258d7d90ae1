// Launch tape: record the kernel launches (and the few stream / memory operations) of a schedule once, replay them from C.
//
// The launch schedules of the two encoders are written in Python (OATrans/engine/video.py, text.py) for clarity; every
// step they issue the same ~1000 launches with the same arguments - all buffers are plan-owned and static, step-dependent
// scalars (Adam step, dropout offset, fp8 scales) live in device memory.  Issuing them through ctypes costs ~20-25 us of
// host time each; hipGraph replay on ROCm 7 still costs ~22 us per kernel node.  A tape replays at the cost of the bare
// hipLaunchKernel call (~4 us).  Tapes are identified by small integers; segments (oat_tape_mark) let the host run
// code between parts of a tape (the gradient all-reduce announcements during backward).
#include "common.h"
#include <mutex>
#include <vector>

namespace oat {

struct Tape {
  std::vector<std::function<void()>> ops;
  std::vector<size_t> seg_end;          // ops index where segment k ends
  std::vector<hipEvent_t> events;       // owned
};
static thread_local Tape* g_rec = nullptr;
static thread_local bool g_paused = false;
static std::mutex g_mu;
static std::vector<Tape*> g_tapes;

bool tape_recording() { return g_rec != nullptr && !g_paused; }
void tape_push(std::function<void()>&& op) { g_rec->ops.emplace_back(std::move(op)); }

}  // namespace oat

using namespace oat;

extern "C" int oat_tape_begin(void) {
  if (g_rec) { set_error("tape_begin: a tape is already being recorded on this thread"); return -1; }
  g_rec = new Tape();
  g_paused = false;
  return 0;
}
// abandon the recording (an exception unwound the schedule)
extern "C" void oat_tape_abort(void) {
  if (!g_rec) return;
  for (hipEvent_t e : g_rec->events) (void)hipEventDestroy(e);
  delete g_rec;
  g_rec = nullptr;
}
// launches issued while paused are executed but not recorded (host callbacks that run between segments)
extern "C" void oat_tape_pause(int on) { g_paused = on != 0; }
// close the current segment; returns its index
extern "C" int oat_tape_mark(void) {
  if (!g_rec) { set_error("tape_mark: no tape is being recorded"); return -1; }
  g_rec->seg_end.push_back(g_rec->ops.size());
  return (int)g_rec->seg_end.size() - 1;
}
// returns the tape id (>= 0)
extern "C" int oat_tape_end(void) {
  if (!g_rec) { set_error("tape_end: no tape is being recorded"); return -1; }
  if (g_rec->seg_end.empty() || g_rec->seg_end.back() != g_rec->ops.size()) g_rec->seg_end.push_back(g_rec->ops.size());
  std::lock_guard<std::mutex> lk(g_mu);
  int id = -1;
  for (size_t i = 0; i < g_tapes.size(); ++i) if (!g_tapes[i]) { id = (int)i; break; }
  if (id < 0) { g_tapes.push_back(nullptr); id = (int)g_tapes.size() - 1; }
  g_tapes[id] = g_rec;
  g_rec = nullptr;
  return id;
}
extern "C" int oat_tape_segments(int id) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (id < 0 || id >= (int)g_tapes.size() || !g_tapes[id]) { set_error("tape_segments: unknown tape"); return -1; }
  return (int)g_tapes[id]->seg_end.size();
}
extern "C" int oat_tape_ops(int id) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (id < 0 || id >= (int)g_tapes.size() || !g_tapes[id]) { set_error("tape_ops: unknown tape"); return -1; }
  return (int)g_tapes[id]->ops.size();
}
// replay segments [seg_lo, seg_hi) (seg_hi < 0: to the end)
extern "C" int oat_tape_replay(int id, int seg_lo, int seg_hi) {
  Tape* t;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (id < 0 || id >= (int)g_tapes.size() || !g_tapes[id]) { set_error("tape_replay: unknown tape"); return -1; }
    t = g_tapes[id];
  }
  if (g_rec) { set_error("tape_replay: not while recording"); return -1; }
  const int ns = (int)t->seg_end.size();
  if (seg_hi < 0 || seg_hi > ns) seg_hi = ns;
  if (seg_lo < 0 || seg_lo > seg_hi) { set_error("tape_replay: bad segment range"); return -3; }
  const size_t lo = seg_lo == 0 ? 0 : t->seg_end[seg_lo - 1], hi = seg_hi == 0 ? 0 : t->seg_end[seg_hi - 1];
  for (size_t i = lo; i < hi; ++i) t->ops[i]();
  return check_launch("tape_replay");
}
extern "C" int oat_tape_free(int id) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (id < 0 || id >= (int)g_tapes.size() || !g_tapes[id]) return 0;
  for (hipEvent_t e : g_tapes[id]->events) (void)hipEventDestroy(e);
  delete g_tapes[id];
  g_tapes[id] = nullptr;
  return 0;
}

// ---- the stream / memory operations a schedule needs between kernels; recorded like launches ---------------------
// everything enqueued on `to` after this call waits for what is on `from` now
extern "C" int oat_stream_edge(void* from, void* to) {
  hipEvent_t ev;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { set_error("stream_edge: hipEventCreate failed"); return -100; }
  hipStream_t f = (hipStream_t)from, t = (hipStream_t)to;
  auto op = [=] { (void)hipEventRecord(ev, f); (void)hipStreamWaitEvent(t, ev, 0); };
  op();
  if (tape_recording()) { g_rec->events.push_back(ev); tape_push(op); }
  else (void)hipEventDestroy(ev);        // destruction is deferred by the runtime until the recorded work completed
  return check_launch("stream_edge");
}
extern "C" int oat_memset_async(void* dst, int byte_value, size_t bytes, void* stream) {
  if (!bytes) return 0;
  hipStream_t s = (hipStream_t)stream;
  auto op = [=] { (void)hipMemsetAsync(dst, byte_value, bytes, s); };
  op();
  if (tape_recording()) tape_push(op);
  return check_launch("memset_async");
}
extern "C" int oat_copy_async(void* dst, const void* src, size_t bytes, void* stream) {
  if (!bytes) return 0;
  hipStream_t s = (hipStream_t)stream;
  auto op = [=] { (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s); };
  op();
  if (tape_recording()) tape_push(op);
  return check_launch("copy_async");
}
