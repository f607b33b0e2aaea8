"""GPU probe: where does the time of the bf16 attention FORWARD kernel go?  (libuvx_probes.so, uvx_probe_attn_timeline.)
Every wave accumulates the cycles of the four sections of a 64-key iteration - S = K Q^T products (with their LDS operand reads
and the next tile's global-load issue), online softmax (VALU), O += V^T P products (transposing LDS reads), staging stores +
barrier - and stamps the kernel's start, loop start, loop end and end.
usage: PYTHONPATH=. python tools/gpu_attn_fwd_timeline.py [encoder|llm]"""
import os
os.environ.setdefault("UVX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultravox_amd", "libuvx_probes.so"))
import sys
import ctypes as C
import torch
from ultravox_amd import ops, _lib

which = sys.argv[1] if len(sys.argv) > 1 else "encoder"
B, T, Hq, Hkv, D, causal, BQ = {"encoder": (8, 1500, 16, 16, 64, False, 128), "llm": (8, 316, 32, 8, 128, True, 64)}[which]
dev = "cuda"
L = _lib.lib()
torch.manual_seed(0)
qkv = (torch.randn(B, T, (Hq + 2 * Hkv) * D, device=dev) * 0.5).bfloat16()
q = qkv[..., :Hq * D].view(B, T, Hq, D)
k = qkv[..., Hq * D:(Hq + Hkv) * D].view(B, T, Hkv, D)
v = qkv[..., (Hq + Hkv) * D:].view(B, T, Hkv, D)


def timed(reps=30):
    ops.attention(q, k, v, causal=causal); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(reps):
        ops.attention(q, k, v, causal=causal)
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / reps * 1e3


us_plain = timed()
nblk = Hq * B * ((T + BQ - 1) // BQ)
stamps = torch.zeros(nblk * 8 * 16, device=dev, dtype=torch.int64)
assert L.uvx_probe_attn_timeline(C.c_void_p(stamps.data_ptr())) == 0, "needs libuvx_probes.so"
us_probe = timed()
L.uvx_probe_attn_timeline(None)
s = stamps.cpu().view(nblk, 8, 16)[:, :4].double()
s = s[s[:, :, 8].min(dim=1).values > 0]                       # blocks that ran iterations
tot = s[:, :, 3] - s[:, :, 0]
it = s[:, :, 8]
print(f"{which}: B={B} T={T} Hq={Hq} Hkv={Hkv} D={D} causal={causal}: {us_plain:.1f} us per call; with stamps {us_probe:.1f} us; {nblk} blocks of 4 waves")
print(f"per wave: kernel {tot.median().item():.0f} ticks = prologue {(s[:, :, 1] - s[:, :, 0]).median().item():.0f} + loop {(s[:, :, 2] - s[:, :, 1]).median().item():.0f} "
      f"+ epilogue {(s[:, :, 3] - s[:, :, 2]).median().item():.0f};  iterations {it.median().item():.0f}")
names = ["S products", "softmax", "PV products", "commit+barrier"]
per = [(s[:, :, 4 + i] / it).median().item() for i in range(4)]
print("per 64-key iteration (median ticks): " + ", ".join(f"{n} {x:.0f}" for n, x in zip(names, per)) + f"  = {sum(per):.0f}")
qt = BQ // 64
print(f"MFMA floor per wave and iteration: S {4 * (D // 32) * qt * 16} + PV {(D // 16) * 2 * qt * 16} cycles; the waves of {('three' if D == 64 else 'one or two')} blocks share a SIMD")
first, last = s[:, :, 0].min().item(), s[:, :, 3].max().item()
print(f"launch span {last - first:.0f} ticks; block starts: {(s[:, 0, 0] - first).quantile(torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0], dtype=torch.double)).tolist()}")
