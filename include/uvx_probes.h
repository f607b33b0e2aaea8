/* Probe / diagnostic entry points of libuvx_probes.so (the library built with -DUVX_PROBES: ultravox_amd/build.py --probes).  They are NOT part
 * of the product ABI: libuvx.so does not export them (round 6; until ABI 15 they sat in uvx.h and returned UVX_ERR_UNSUPPORTED there).  The tools
 * under tools/ that use them select the probes library with UVX_LIB=ultravox_amd/libuvx_probes.so. */
#ifndef UVX_PROBES_H
#define UVX_PROBES_H
#include "uvx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic for the stream-K GEMM launches (probe variants 39..42; the production picker never selects them): blocks that wait for another block's partial sums spin for a bounded time (~1 s) and then give
 * up rather than hang the queue.  Returns how many did since the last call (0 on a healthy run; a non-zero count means
 * wrong output tiles), -1 on a HIP error.  Synchronises the device. */
int32_t uvx_gemm_streamk_timeouts(void);

/* while `stamps` is non-null, every wave of the fused attention
 * backward kernel writes a 16 x u64 record of cycle-counter stamps into stamps[((b * Hq + h) * 8 + wave) * 16 + slot]: 0 start,
 * 1 prologue done, 2 + 2 p / 3 + 2 p pass p of phase 1 done / its dK, dV stored, 8 phase 2 done, 9 dQ stored, 10 cycles inside
 * phase-1 steps, 11 cycles at phase-1 barriers + staging stores, 12 steps taken, 13 cycles inside phase-2 products.
 * tools/gpu_attn_timeline.py prints the breakdown.  null switches the stamps off. */
int32_t uvx_probe_attn_timeline(void* stamps);

#ifdef __cplusplus
}
#endif
#endif
