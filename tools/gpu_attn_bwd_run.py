"""Run the LLM-shaped attention backward (C2: B = 8, 32 / 8 heads, T = 316, head_dim 128, causal) a few times - the
workload of tools/pmc_attn_bwd.sh's rocprofv3 passes."""
import torch
from ultravox_amd import ops
torch.manual_seed(0)
B, T, Hq, Hkv, D = 8, 316, 32, 8, 128
q = torch.randn(B, T, Hq, D, device="cuda").bfloat16()
k = torch.randn(B, T, Hkv, D, device="cuda").bfloat16()
v = torch.randn(B, T, Hkv, D, device="cuda").bfloat16()
do = torch.randn(B, T, Hq * D, device="cuda").bfloat16()
o, lse = ops.attention(q, k, v, causal=True)
for _ in range(6): ops.attention_bwd(q, k, v, o, lse, do, causal=True)
torch.cuda.synchronize()
