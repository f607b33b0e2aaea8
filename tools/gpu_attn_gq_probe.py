"""GPU probe (round 6): the LLM's causal attention forward (head_dim 128, GQA) in the grouped-query block form (tuning option 25: the four waves of a block take
four query heads of one KV head, attention.hip attn_fwd_k<.., GQ>) against the default form, at the shapes the training step and the prefill launch it with.
Outputs compared bit for bit; HIP-event timing over back-to-back launches through the C ABI on ONE box."""
import sys
import torch
from ultravox_amd import _lib, ops

L = _lib.lib()
torch.manual_seed(0)
dev = "cuda"


def timeit(fn, n=60):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


names = {5: "one head per block (rounds 1-5)", 4: "GQ, 1 query tile per wave", 1: "GQ, 2 query tiles", 3: "GQ, 3 query tiles"}
for (B, Hq, Hkv, T, tag, D) in ((8, 32, 8, 316, "C2 step: Llama-3-8B", 128), (8, 32, 8, 176, "KL teacher", 128), (1, 64, 8, 316, "70B prefill, 1 prompt", 128), (8, 64, 8, 316, "70B step", 128),
                              (8, 32, 8, 316, "Llama-3.2-1B step (hd 64)", 64), (1, 32, 8, 316, "Llama-3.2-1B, 1 prompt", 64)):
    qkv = torch.randn(B, T, (Hq + 2 * Hkv) * D, device=dev).bfloat16()
    q = qkv[..., :Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hkv) * D].view(B, T, Hkv, D)
    v = qkv[..., (Hq + Hkv) * D:].view(B, T, Hkv, D)
    kv_start = torch.zeros(B, dtype=torch.int32, device=dev)
    kv_len = torch.full((B,), T, dtype=torch.int32, device=dev)
    if B > 1:
        kv_start[1] = 7
        kv_len[B - 1] = T - 11
    fl = 2.0 * B * Hq * T * T * D
    for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
        for form in (5, 4, 1, 3) if D == 128 else (5, 4, 1):
            L.uvx_set_option(25, form)
            o, lse = ops.attention(q, k, v, causal=True, kv_start=kv_start, kv_len=kv_len)
            if form == 5:
                o_ref, lse_ref = o, lse
            us = timeit(lambda: ops.attention(q, k, v, causal=True, kv_start=kv_start, kv_len=kv_len))
            print(f"[{rnd}] {tag:24s} B={B} Hq={Hq} Hkv={Hkv} T={T} {names[form]:30s} {us:7.1f} us ({fl / us / 1e6:6.1f} TF/s)  identical={torch.equal(o, o_ref) and torch.equal(lse, lse_ref)}", flush=True)
L.uvx_set_option(25, 0)
