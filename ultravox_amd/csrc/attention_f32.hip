#include "common.h"
#include "kernels.h"
namespace uvx {
int attention_fwd_f32(hipStream_t, const AttnDesc&) { uvx_set_error("f32 attention not built yet"); return UVX_ERR_UNSUPPORTED; }
int attention_bwd_f32(hipStream_t, const AttnBwdDesc&) { uvx_set_error("f32 attention not built yet"); return UVX_ERR_UNSUPPORTED; }
}
