"""GPU probe: the LLM's attention kernels at the C2 shape (B = 8, 32 query / 8 kv heads, T = 316, head_dim 128, causal), timed
with HIP events over back-to-back launches through the C ABI (uvx_attention_fwd / uvx_attention_bwd: the backward includes its
three operand transposes and the GQA reduction).  Also the encoder's shape (B = 8, 16 heads, T = 1500, head_dim 64)."""
import torch
from ultravox_amd import ops

torch.manual_seed(0)
dev = "cuda"
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, (B, Hq, Hkv, T, D, causal) in {"llm": (8, 32, 8, 316, 128, True), "encoder": (8, 16, 16, 1500, 64, False)}.items():
    q = torch.randn(B, T, Hq, D, device=dev).bfloat16()
    k = torch.randn(B, T, Hkv, D, device=dev).bfloat16()
    v = torch.randn(B, T, Hkv, D, device=dev).bfloat16()
    do = torch.randn(B, T, Hq * D, device=dev).bfloat16()
    o, lse = ops.attention(q, k, v, causal=causal)
    f = timeit(lambda: ops.attention(q, k, v, causal=causal))
    fl = 4.0 * B * Hq * T * T * D * (0.5 if causal else 1.0)
    b = timeit(lambda: ops.attention_bwd(q, k, v, o, lse, do, causal=causal))
    print(f"{name}: fwd {f:7.1f} us ({fl / f / 1e6:6.1f} TF/s)   bwd {b:7.1f} us ({2.5 * fl / b / 1e6:6.1f} TF/s)", flush=True)
