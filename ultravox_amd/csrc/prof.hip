// Live kernel timing for bench.py: HIP events recorded on the SAME stream as the kernel, immediately
// around the launch (torch.cuda.Event would only see torch's current stream).  Off by default; costs
// nothing when off.  One pool of event pairs, reused between uvx_prof_begin / uvx_prof_end.
#include <vector>
#include "common.h"
#include "kernels.h"
#include "../../include/uvx.h"

namespace uvx {
bool g_prof_on = false;
namespace {
struct Rec { hipEvent_t a, b; int cls; double flops, bytes; };
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
  r.cls = cls; r.flops = flops; r.bytes = bytes;
  hipEventRecord(r.a, st);
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
