"""Run the encoder-shaped and the LLM-shaped attention forward a few times (for rocprofv3 passes / timing)."""
import sys, torch
from ultravox_amd import ops
torch.manual_seed(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "enc"
if mode == "enc":
    B, T, H, Hkv, D, causal = 8, 1500, 16, 16, 64, False
else:
    B, T, H, Hkv, D, causal = 8, 316, 32, 8, 128, True
qkv = torch.randn(B, T, (H + 2 * Hkv) * D, device="cuda").bfloat16()
q = qkv[..., :H * D].view(B, T, H, D); k = qkv[..., H * D:(H + Hkv) * D].view(B, T, Hkv, D); v = qkv[..., (H + Hkv) * D:].view(B, T, Hkv, D)
from ultravox_amd import _lib
if len(sys.argv) > 2: _lib.lib().uvx_attention_force_qt(int(sys.argv[2]))
for _ in range(3): o, lse = ops.attention(q, k, v, causal=causal)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): o, lse = ops.attention(q, k, v, causal=causal)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
fl = 4.0 * B * H * T * T * D * (0.5 if causal else 1.0)
print(sys.argv[1:], "attention fwd (+V transpose):", ms * 1e3, "us", fl / ms / 1e9, "TF")
