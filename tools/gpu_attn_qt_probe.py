"""GPU probe: the head_dim-64 forward attention kernel (the Whisper tower, B = 8, 16 heads, 1500 frames) with 1..4 16-row query tiles per wave (uvx_attention_force_qt)."""
import torch
from ultravox_amd import _lib, ops
L = _lib.lib()
B, H, T, D = 8, 16, 1500, 64
qkv = torch.randn(B, T, 3 * H * D, device="cuda").bfloat16()
q, k, v = (qkv[..., i * H * D:(i + 1) * H * D].view(B, T, H, D) for i in range(3))
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rnd in range(3):
    for qt in (2, 1, 3, 4):
        L.uvx_attention_force_qt(qt)
        print(rnd, "qt", qt, round(timeit(lambda: ops.attention(q, k, v, causal=False)), 1), "us", flush=True)
L.uvx_attention_force_qt(0)
