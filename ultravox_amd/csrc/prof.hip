// Live kernel timing for bench.py: HIP events recorded on the SAME stream as the kernel, immediately
// around the launch (torch.cuda.Event would only see torch's current stream).  Off by default; costs
// nothing when off.  One pool of event pairs, reused between uvx_prof_begin / uvx_prof_end.
#include <algorithm>
#include <utility>
#include <vector>
#include "common.h"
#include "kernels.h"
#include "../../include/uvx.h"

namespace uvx {
bool g_prof_on = false;
namespace {
struct Rec { hipEvent_t a, b; int cls; double flops, bytes; int m, n, k, batch, variant; };
std::vector<Rec> g_pool;
size_t g_used = 0;
}  // namespace

void prof_record_begin(hipStream_t st, int cls, double flops, double bytes) {
  if (g_used == g_pool.size()) {
    Rec r;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { g_prof_on = false; return; }
    g_pool.push_back(r);
  }
  Rec& r = g_pool[g_used];
  r.cls = cls; r.flops = flops; r.bytes = bytes; r.m = r.n = r.k = r.batch = r.variant = 0;
  hipEventRecord(r.a, st);
}
// The GEMM family attaches its events to the dispatch itself (gemm.hip UVX_GEMM_LAUNCH): take a record and its two events, launch,
// commit.  (Launches that fall back to a plain launch after taking - none today - would leave the events unrecorded.)
bool prof_take(int cls, double flops, double bytes, hipEvent_t* a, hipEvent_t* b) {
  if (g_used == g_pool.size()) {
    Rec r;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { g_prof_on = false; return false; }
    g_pool.push_back(r);
  }
  Rec& r = g_pool[g_used];
  r.cls = cls; r.flops = flops; r.bytes = bytes; r.m = r.n = r.k = r.batch = r.variant = 0;
  *a = r.a; *b = r.b;
  return true;
}
void prof_commit() { if (g_used < g_pool.size()) ++g_used; }
void prof_tag(int m, int n, int k, int batch, int variant) {
  if (g_used < g_pool.size()) { Rec& r = g_pool[g_used]; r.m = m; r.n = n; r.k = k; r.batch = batch; r.variant = variant; }
}
void prof_record_end(hipStream_t st) {
  if (g_used < g_pool.size()) hipEventRecord(g_pool[g_used++].b, st);
}
}  // namespace uvx

extern "C" int32_t uvx_prof_begin(void) {
  uvx::g_used = 0;
  uvx::g_prof_on = true;
  return UVX_OK;
}

// Pause / resume inside a region (records are kept): bench.py times the GEMM launches of every N-th step only - a timed launch
// carries a completion signal with timestamps, ~1.4 us each (profiles/r03_prof_event_overhead.txt).
extern "C" int32_t uvx_prof_enable(int32_t on) {
  uvx::g_prof_on = on != 0;
  return UVX_OK;
}

// Per-launch records of the last profiled region (call BEFORE uvx_prof_end): out[i*6 + {M, N, K, batch,
// variant, ms}] for the first min(count, max_records) GEMM launches; returns the number of records.
extern "C" int32_t uvx_prof_records(double* out, int32_t max_records) {
  int n = 0;
  for (size_t i = 0; i < uvx::g_used && n < max_records; ++i) {
    auto& r = uvx::g_pool[i];
    if (r.cls != uvx::PROF_GEMM) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) break;
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.a, r.b);
    double* o = out + (size_t)n * 6;
    o[0] = r.m; o[1] = r.n; o[2] = r.k; o[3] = r.batch; o[4] = r.variant; o[5] = ms;
    ++n;
  }
  return n;
}

// Wall time during which at least one launch of class `cls` was executing (union of the [begin, end] event intervals), in ms.
// Equals the summed durations when every launch is on one stream; with the two-stream schedule (tuning option 11) launches of
// the two chains overlap, each one's own interval includes the time it shared the chip, and the union is the honest
// denominator for an aggregate rate.  Call BEFORE uvx_prof_end.  Negative on error.
extern "C" double uvx_prof_union_ms(int32_t cls) {
  std::vector<std::pair<float, float>> iv;
  hipEvent_t ref = nullptr;
  for (size_t i = 0; i < uvx::g_used; ++i) {
    auto& r = uvx::g_pool[i];
    if (hipEventSynchronize(r.b) != hipSuccess) return -1.0;
    if (!ref) ref = r.a;
    if (r.cls != cls) continue;
    float ta = 0.f, tb = 0.f;
    if (hipEventElapsedTime(&ta, ref, r.a) != hipSuccess || hipEventElapsedTime(&tb, ref, r.b) != hipSuccess) return -1.0;
    iv.emplace_back(ta, tb);
  }
  std::sort(iv.begin(), iv.end());
  double total = 0.0;
  float lo = 0.f, hi = 0.f;
  bool open = false;
  for (auto& x : iv) {
    if (open && x.first <= hi) { hi = std::max(hi, x.second); continue; }
    if (open) total += hi - lo;
    lo = x.first; hi = x.second; open = true;
  }
  if (open) total += hi - lo;
  return total;
}

// out[cls] = {launches, total_ms, total_flops, total_bytes}; synchronises on the recorded events.
extern "C" int32_t uvx_prof_end(double* out, int32_t n_classes) {
  uvx::g_prof_on = false;
  for (int i = 0; i < n_classes * 4; ++i) out[i] = 0.0;
  for (size_t i = 0; i < uvx::g_used; ++i) {
    auto& r = uvx::g_pool[i];
    UVX_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    UVX_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    if (r.cls < n_classes) {
      out[r.cls * 4 + 0] += 1.0; out[r.cls * 4 + 1] += ms; out[r.cls * 4 + 2] += r.flops; out[r.cls * 4 + 3] += r.bytes;
    }
  }
  uvx::g_used = 0;
  return UVX_OK;
}
