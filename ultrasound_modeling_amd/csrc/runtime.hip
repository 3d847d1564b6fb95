// Error reporting for the C ABI: a thread-local message, never an exception or an exit.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";
thread_local const float* usseg_epi_scale[4] = {nullptr, nullptr, nullptr, nullptr};
thread_local UssegTapMask usseg_tap_mask = {0, {0x1ff, 0x1ff, 0x1ff, 0x1ff}};

void usseg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int usseg_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    usseg_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return USSEG_ERR_LAUNCH;
  }
  return USSEG_OK;
}

extern "C" const char* usseg_last_error(void) { return g_err; }
extern "C" int usseg_version(void) { return 1; }

// ---- opt-in per-launch timing ------------------------------------------------------------------
#include <vector>
struct ProfKind { std::vector<hipEvent_t> start, stop; int used = 0; };
static thread_local ProfKind g_prof[5];      // kinds are bits: 1 conv forward / backward-data, 2 weight gradients, 4 fused tile kernels (cardinal, stem)
//      // per host thread, like the deferral queues: two threads driving two streams do not share them
static thread_local int g_prof_mask = 0;

extern "C" int usseg_prof_enable(int32_t kind_mask, int32_t capacity) {
  USSEG_CHECK_ARG(capacity > 0 && kind_mask > 0 && kind_mask < 8, "prof_enable: bad args");
  usseg_prof_disable();
  for (int k = 1; k <= 4; k <<= 1) {
    if (!(kind_mask & k)) continue;
    g_prof[k].start.resize(capacity);
    g_prof[k].stop.resize(capacity);
    for (int i = 0; i < capacity; ++i) {
      if (hipEventCreate(&g_prof[k].start[i]) != hipSuccess || hipEventCreate(&g_prof[k].stop[i]) != hipSuccess) {
        usseg_set_error("prof_enable: hipEventCreate failed");
        return USSEG_ERR_LAUNCH;
      }
    }
    g_prof[k].used = 0;
  }
  g_prof_mask = kind_mask;
  return USSEG_OK;
}

extern "C" int usseg_prof_disable(void) {
  g_prof_mask = 0;
  for (int k = 1; k <= 4; k <<= 1) {
    for (auto e : g_prof[k].start) (void)hipEventDestroy(e);
    for (auto e : g_prof[k].stop) (void)hipEventDestroy(e);
    g_prof[k].start.clear();
    g_prof[k].stop.clear();
    g_prof[k].used = 0;
  }
  return USSEG_OK;
}

extern "C" int usseg_prof_read(int32_t kind, double* total_ms, int64_t* launches) {
  USSEG_CHECK_ARG((kind == 1 || kind == 2 || kind == 4) && total_ms && launches, "prof_read: bad args");
  ProfKind& p = g_prof[kind];
  double tot = 0.0;
  for (int i = 0; i < p.used; ++i) {
    if (hipEventSynchronize(p.stop[i]) != hipSuccess) { usseg_set_error("prof_read: event sync failed"); return USSEG_ERR_LAUNCH; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.start[i], p.stop[i]) != hipSuccess) { usseg_set_error("prof_read: elapsed failed"); return USSEG_ERR_LAUNCH; }
    tot += ms;
  }
  *total_ms = tot;
  *launches = p.used;
  p.used = 0;
  return USSEG_OK;
}

int usseg_prof_start(int kind, hipStream_t s) {
  if (!(g_prof_mask & kind)) return -1;
  ProfKind& p = g_prof[kind];
  if (p.used >= (int)p.start.size()) return -1;
  int slot = p.used++;
  (void)hipEventRecord(p.start[slot], s);
  return slot;
}
void usseg_prof_stop(int kind, int slot, hipStream_t s) {
  if (slot >= 0) (void)hipEventRecord(g_prof[kind].stop[slot], s);
}
