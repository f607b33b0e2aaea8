"""GPU probe: time the bf16 attention kernels at the step's shapes (encoder: 8 x 1500 x 16 heads x 64, non-causal; LLM:
8 x 316 x 32 / 8 heads x 128, causal), forward and backward, 50 launches each.  UVX_LIB selects the library (A/B of two builds).
usage: [UVX_LIB=...] PYTHONPATH=. python tools/gpu_attn_shapes_probe.py"""
import torch
from ultravox_amd import ops

dev = "cuda"
torch.manual_seed(0)


def timed(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(reps):
        fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / reps * 1e3


for name, (B, T, Hq, Hkv, D, causal) in {"encoder": (8, 1500, 16, 16, 64, False), "llm": (8, 316, 32, 8, 128, True)}.items():
    qkv = (torch.randn(B, T, (Hq + 2 * Hkv) * D, device=dev) * 0.5).bfloat16()
    q = qkv[..., :Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hkv) * D].view(B, T, Hkv, D)
    v = qkv[..., (Hq + Hkv) * D:].view(B, T, Hkv, D)
    o, lse = ops.attention(q, k, v, causal=causal)
    dout = (torch.randn(B, T, Hq * D, device=dev) * 0.1).bfloat16()
    fwd = timed(lambda: ops.attention(q, k, v, causal=causal))
    bwd = timed(lambda: ops.attention_bwd(q, k, v, o, lse, dout, causal=causal))
    fl = 4.0 * B * Hq * T * T * D * (0.5 if causal else 1.0)
    print(f"{name:8s} fwd {fwd:7.1f} us ({fl / fwd / 1e6:6.0f} TF/s)   bwd {bwd:7.1f} us ({2.5 * fl / bwd / 1e6:6.0f} TF/s)   [incl. the wrappers' allocations]")
