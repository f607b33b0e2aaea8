"""GPU diagnostic (round 6): which trainable tensors differ between the fused (GemmDesc::act 2 / 3, LoRA terms) and the separate-kernel forms of the
training tower, at the small test configuration; and the GEMM-level identity of act 3 on the tower's own shapes and value ranges."""
import ctypes as C
import sys
import torch
sys.path.insert(0, "tests")
from test_lora_gpu import _setup
from ultravox_amd import _lib, ops

L = _lib.lib()
cfg, sd, model, oracle, gb, ob, mel = _setup(torch.bfloat16, r=8)
model.train()


def run(o21, o22):
    L.uvx_set_option(21, o21); L.uvx_set_option(22, o22)
    loss = model.forward_backward(audio_values=mel, **gb)
    torch.cuda.synchronize()
    return loss.clone(), {k: v.clone() for k, v in model.projector_grads().items()}


l0, g0 = run(1, 1)
for o21, o22 in ((1, 1), (0, 1), (1, 0), (0, 0)):
    l1, g1 = run(o21, o22)
    bad = {k: int((g1[k] != g0[k]).sum()) for k in g0 if not torch.equal(g1[k], g0[k])}
    print(f"options 21={o21} 22={o22}: loss equal {torch.equal(l0, l1)}; tensors that differ: {bad}", flush=True)
L.uvx_set_option(21, 0); L.uvx_set_option(22, 0)
# GEMM level, the tower's shapes: M = 300 rows, N = ffn, K = d
a_cfg = cfg.audio_config
M, N, K = 300, a_cfg.encoder_ffn_dim, a_cfg.d_model
g = torch.Generator(device="cuda").manual_seed(1)
for scale_d, scale_x in ((1.0, 1.5), (1e-3, 4.0), (1e-6, 8.0), (1e-9, 0.2)):
    a = (torch.randn(M, K, device="cuda", generator=g) * scale_d).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    x = (torch.randn(M, N, device="cuda", generator=g) * scale_x).bfloat16()
    d_ref = ops.gemm(a, w)
    g_ref = torch.empty_like(d_ref)
    L.uvx_gelu_bwd(None, _lib.BF16, C.c_void_p(d_ref.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(g_ref.data_ptr()), C.c_longlong(d_ref.numel()))
    got = ops.gemm(a, w, act="gelu_bwd", c2=x)
    n = int((got != g_ref).sum())
    print(f"act 3 at {M}x{N}x{K}, |d| ~ {scale_d:g}, |x| ~ {scale_x:g}: {n} of {got.numel()} differ", flush=True)
    if n:
        i = (got != g_ref).nonzero()[0]
        print("   first:", got[i[0], i[1]].item(), g_ref[i[0], i[1]].item(), "d", d_ref[i[0], i[1]].item(), "x", x[i[0], i[1]].item())
