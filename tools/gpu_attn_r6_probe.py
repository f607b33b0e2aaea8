"""GPU probe (round 6): the Whisper tower's attention kernels (B = 8, 16 heads, 1500 frames, head_dim 64, non-causal) in every tile
form of tuning options 19 (backward pair) and 20 (forward row max), timed with HIP events over back-to-back launches through the C ABI
on ONE box; also the LLM's shape for reference.  Each form's outputs are compared bit for bit with the round-5 form."""
import ctypes as C
import sys
import torch
from ultravox_amd import _lib, ops

L = _lib.lib()
torch.manual_seed(0)
dev = "cuda"


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B, H, T, D = 8, 16, 1500, 64
qkv = torch.randn(B, T, 3 * H * D, device=dev).bfloat16()          # the encoder's fused q | k | v rows (row stride 3 d)
q, k, v = (qkv[..., i * H * D:(i + 1) * H * D].view(B, T, H, D) for i in range(3))
do = torch.randn(B, T, H * D, device=dev).bfloat16()
fl = 4.0 * B * H * T * T * D
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for rnd in range(rounds):
    for form, name in ((1, "fwd ds_bpermute max (rounds 1-5)"), (0, "fwd v_permlane max")):
        L.uvx_set_option(20, form)
        o, lse = ops.attention(q, k, v, causal=False)
        if form == 1:
            o_ref, lse_ref = o, lse
        us = timeit(lambda: ops.attention(q, k, v, causal=False))
        print(f"[{rnd}] encoder {name:34s} {us:7.1f} us ({fl / us / 1e6:6.1f} TF/s)  identical={torch.equal(o, o_ref) and torch.equal(lse, lse_ref)}", flush=True)
    names = {1: "dq x1 tile, dkdv x1 (rounds 1-5)", 0: "dq x2, dkdv x1 (default)", 2: "dq x2, dkdv x2", 3: "dq x1, dkdv x2", 4: "x2 x2, 64-row steps",
             5: "8 waves, 64-row steps, x1 x1", 6: "8 waves, 64-row steps, dq x2"}
    for form in (1, 0, 2, 3, 4, 5, 6):
        L.uvx_set_option(19, form)
        g = ops.attention_bwd(q, k, v, o_ref, lse_ref, do, causal=False)
        if form == 1:
            g_ref = g
        us = timeit(lambda: ops.attention_bwd(q, k, v, o_ref, lse_ref, do, causal=False))
        same = all(torch.equal(a, b) for a, b in zip(g, g_ref))
        print(f"[{rnd}] encoder bwd {names[form]:34s} {us:7.1f} us ({2.5 * fl / us / 1e6:6.1f} TF/s by 5 products)  identical={same}", flush=True)
    L.uvx_set_option(19, 0)
    L.uvx_set_option(20, 0)
