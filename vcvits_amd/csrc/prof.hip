// prof.hip -- event-pair pool behind vcv_prof_begin / vcv_prof_end
#include <cstdio>
#include <mutex>
#include <vector>

#include "common.h"
#include "prof.h"

namespace {
std::mutex g_mu;
bool g_on = false, g_paused = false;
std::vector<hipEvent_t> g_start, g_stop;
std::vector<int> g_cls;
std::vector<double> g_flops, g_bytes, g_roof;
double g_last_bytes[VCV_PROF_NCLS] = {};
double g_last_roof[VCV_PROF_NCLS] = {};
std::vector<int> g_tags;  // 12 ints per slot
constexpr int NTAG = 12;
size_t g_used = 0, g_last_used = 0;
std::vector<float> g_ms;
}  // namespace

int vcv_prof_start(int cls, double flops, hipStream_t st, const int* tag, int ntag) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_on || g_used >= g_start.size()) return -1;
  const int slot = (int)g_used++;
  g_cls[slot] = cls;
  g_flops[slot] = flops;
  g_roof[slot] = flops / VCV_PEAK_F32_MFMA;
  for (int i = 0; i < NTAG; ++i) g_tags[slot * NTAG + i] = (tag && i < ntag) ? tag[i] : 0;
  hipEventRecord(g_start[slot], st);
  return slot;
}

void vcv_prof_events(int cls, double flops, const int* tag, int ntag, hipEvent_t* start, hipEvent_t* stop, double bytes,
                     double roof_s) {
  *start = *stop = nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_on || g_paused || g_used >= g_start.size()) return;
  const int slot = (int)g_used++;
  g_cls[slot] = cls;
  g_flops[slot] = flops;
  g_bytes[slot] = bytes;
  g_roof[slot] = roof_s > 0.0 ? roof_s : flops / VCV_PEAK_F32_MFMA;
  for (int i = 0; i < NTAG; ++i) g_tags[slot * NTAG + i] = (tag && i < ntag) ? tag[i] : 0;
  *start = g_start[slot];
  *stop = g_stop[slot];
}

void vcv_prof_stop(int slot, hipStream_t st) {
  if (slot < 0) return;
  hipEventRecord(g_stop[slot], st);
}

extern "C" int vcv_prof_begin(int max_launches) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (max_launches <= 0) return VCV_EINVAL;
  while ((int)g_start.size() < max_launches) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return VCV_EHIP;
    g_start.push_back(a);
    g_stop.push_back(b);
  }
  g_cls.assign(g_start.size(), 0);
  g_flops.assign(g_start.size(), 0.0);
  g_bytes.assign(g_start.size(), 0.0);
  g_roof.assign(g_start.size(), 0.0);
  g_tags.assign(g_start.size() * NTAG, 0);
  g_used = 0;
  g_on = true;
  g_paused = false;
  return VCV_OK;
}

// Sampling: while paused, launches get no events (plain launches); the window stays open.  bench.py times every other
// step of its timed region this way: dispatch-attached events cost ~4 % of the step (they keep consecutive kernels from
// overlapping their launch latencies).
// 1 while launches get events attached (window open and not paused)
extern "C" int vcv_prof_active(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_on && !g_paused ? 1 : 0;
}

extern "C" int vcv_prof_pause(int paused) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_paused = paused != 0;
  return VCV_OK;
}

// out[cls*3 + {0,1,2}] = {launch count, total milliseconds, total algorithmic flops}; returns the
// number of launches that did not fit the pool (0 = all timed).  Synchronises on the events.
extern "C" int vcv_prof_end(double* out, int ncls) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!out || ncls < VCV_PROF_NCLS) return VCV_EINVAL;
  g_on = false;
  for (int i = 0; i < ncls * 3; ++i) out[i] = 0.0;
  for (int c = 0; c < VCV_PROF_NCLS; ++c) g_last_bytes[c] = g_last_roof[c] = 0.0;
  for (size_t i = 0; i < g_used; ++i) {
    if (hipEventSynchronize(g_stop[i]) != hipSuccess) return VCV_EHIP;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_start[i], g_stop[i]) != hipSuccess) return VCV_EHIP;
    const int c = g_cls[i];
    out[c * 3 + 0] += 1.0;
    out[c * 3 + 1] += ms;
    out[c * 3 + 2] += g_flops[i];
    if (c >= 0 && c < VCV_PROF_NCLS) g_last_bytes[c] += g_bytes[i], g_last_roof[c] += g_roof[i];
  }
  g_last_used = g_used;
  g_ms.assign(g_used, 0.f);
  for (size_t i = 0; i < g_used; ++i) hipEventElapsedTime(&g_ms[i], g_start[i], g_stop[i]);
  g_used = 0;
  return 0;
}

// out[cls] = total algorithmic HBM bytes (operands once + result once) of the class's launches in the last window
extern "C" int vcv_prof_bytes(double* out, int ncls) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!out || ncls < VCV_PROF_NCLS) return VCV_EINVAL;
  for (int c = 0; c < VCV_PROF_NCLS; ++c) out[c] = g_last_bytes[c];
  return VCV_OK;
}

// out[cls] = sum over the class's launches in the last window of their time at the dense peak of the matrix pipe each
// one runs on (seconds; see prof.h)
extern "C" int vcv_prof_roof(double* out, int ncls) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!out || ncls < VCV_PROF_NCLS) return VCV_EINVAL;
  for (int c = 0; c < VCV_PROF_NCLS; ++c) out[c] = g_last_roof[c];
  return VCV_OK;
}

// debugging aid: per-launch records of the last vcv_prof_begin/end window as CSV
// (cls, ms, gflop, then the 12 shape tags the launcher attached)
extern "C" int vcv_prof_dump(const char* path) {
  std::lock_guard<std::mutex> lk(g_mu);
  FILE* f = fopen(path, "w");
  if (!f) return VCV_EINVAL;
  for (size_t i = 0; i < g_last_used; ++i) {
    fprintf(f, "%d,%.6f,%.6f", g_cls[i], g_ms[i], g_flops[i] / 1e9);
    for (int t = 0; t < NTAG; ++t) fprintf(f, ",%d", g_tags[i * NTAG + t]);
    fprintf(f, "\n");
  }
  fclose(f);
  return VCV_OK;
}
