"""GPU probe: where does the time of the fused attention backward kernel (attn_bwd_fused_k, one block per (batch, query head)) go?
Runs the probes build with uvx_probe_attn_timeline switched on: every wave stamps the cycle counter at the phase boundaries and
accumulates the cycles spent inside the phase-1 steps, at the phase-1 barriers (+ staging stores) and inside the phase-2 products.
Prints, per wave index (the waves own different key / query tiles, so their loads differ), the median over the blocks.
usage: PYTHONPATH=. python tools/gpu_attn_timeline.py [B T Hq Hkv]     (default: the C2 LLM shape 8 x 316 x 32 / 8, head_dim 128)"""
import os
os.environ.setdefault("UVX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultravox_amd", "libuvx_probes.so"))
import sys
import ctypes as C
import torch
from ultravox_amd import ops, _lib

B, T, Hq, Hkv = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (8, 316, 32, 8)
D = 128
dev = "cuda"
L = _lib.lib()
torch.manual_seed(0)
qkv = (torch.randn(B, T, (Hq + 2 * Hkv) * D, device=dev) * 0.5).bfloat16()     # the QKV GEMM's output: token stride = all heads
q = qkv[..., :Hq * D].view(B, T, Hq, D)
k = qkv[..., Hq * D:(Hq + Hkv) * D].view(B, T, Hkv, D)
v = qkv[..., (Hq + Hkv) * D:].view(B, T, Hkv, D)
o, lse = ops.attention(q, k, v, causal=True)
dout = (torch.randn(B, T, Hq * D, device=dev) * 0.1).bfloat16()


def timed(reps=20):
    ops.attention_bwd(q, k, v, o, lse, dout, causal=True)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(reps):
        ops.attention_bwd(q, k, v, o, lse, dout, causal=True)
    e[1].record()
    torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / reps * 1e3


us_plain = timed()
stamps = torch.zeros(B * Hq * 8 * 16, device=dev, dtype=torch.int64)
assert L.uvx_probe_attn_timeline(C.c_void_p(stamps.data_ptr())) == 0, "needs libuvx_probes.so"
us_probe = timed()
ops.attention_bwd(q, k, v, o, lse, dout, causal=True)
torch.cuda.synchronize()
L.uvx_probe_attn_timeline(None)
s = stamps.cpu().view(B * Hq, 8, 16).double()
rel = s[:, :, :10] - s[:, :, :1]                    # stamps relative to the wave's start
total = rel[:, :, 9]
print(f"B={B} T={T} Hq={Hq} Hkv={Hkv} D={D}: backward (fused kernel + GQA reduce) {us_plain:.1f} us per call; with stamps {us_probe:.1f} us")
print(f"kernel length in counter ticks: median {total.median().item():.0f} (min {total.min().item():.0f}, max {total.max().item():.0f}) "
      f"-> {total.median().item() / max(us_probe, 1e-9):.1f} ticks per us of the whole call")
names = ["prologue", "pass0 loop", "pass0 store", "pass1 loop", "pass1 store", "pass2 loop", "pass2 store", "phase2 loop", "dQ store"]
print("per-wave medians over the blocks, ticks (the interval ENDING at the named point):")
print("wave " + " ".join(f"{n:>12s}" for n in names) + f" {'p1 steps':>10s} {'p1 wait':>10s} {'steps':>6s} {'p2 prod':>10s}")
for w in range(8):
    iv = [(rel[:, w, i + 1] - rel[:, w, i]).median().item() for i in range(9)]
    extra = [s[:, w, 10].median().item(), s[:, w, 11].median().item(), s[:, w, 12].median().item(), s[:, w, 13].median().item()]
    print(f"{w:4d} " + " ".join(f"{x:12.0f}" for x in iv) + f" {extra[0]:10.0f} {extra[1]:10.0f} {extra[2]:6.0f} {extra[3]:10.0f}")
allw = [(rel[:, :, i + 1] - rel[:, :, i]).max(dim=1).values.median().item() for i in range(9)]
print("slowest wave per block, median: " + " ".join(f"{n} {x:.0f}" for n, x in zip(names, allw)))
start_spread = (s[:, :, 0].max(dim=1).values - s[:, :, 0].min(dim=1).values).median().item()
first = s[:, :, 0].min().item()
print(f"block start spread across the launch: first wave start to last block's start {(s[:, :, 0].max().item() - first):.0f} ticks; "
      f"within-block wave start spread {start_spread:.0f}")
print(f"per step (phase 1): {s[:, :, 10].sum().item() / max(s[:, :, 12].sum().item(), 1):.0f} ticks; MFMA floor of a step = 32 MFMAs x 16 cycles = 512 cycles per wave")
