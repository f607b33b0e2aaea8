"""Per-kernel parity on a real MI355X: every HIP kernel (through the C ABI) against a plain PyTorch fp32
reference of the same op, with the bf16 rounding points of the reference's GPU path restated.
Tolerances are written per test; integer / index work is asserted bit-exact."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def ops():
    from ultravox_amd import ops as o
    return o


def bf(x):
    return x.to(torch.bfloat16)


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (100, 132, 192), (37, 8, 64), (1504, 2048, 1024),
                                   (2528, 4096, 512), (1, 4, 64)])
def test_gemm_matches_fp32_matmul(M, N, K):
    torch.manual_seed(M + N + K)
    a = bf(torch.randn(M, K, device=DEV) * 0.5)
    b = bf(torch.randn(N, K, device=DEV) * 0.5)  # asymmetric: catches operand / output transposes
    out = ops().gemm(a, b)
    ref = a.float() @ b.float().t()
    # bf16 output: one rounding of the f32 accumulator -> <= 2^-8 relative to each element (+ f32 sum order)
    assert torch.allclose(out.float(), ref, rtol=2 ** -7, atol=1e-3 * math.sqrt(K))


def test_gemm_epilogues_and_strides():
    torch.manual_seed(0)
    M, N, K = 300, 256, 256
    a = bf(torch.randn(M, K, device=DEV) * 0.5)
    b = bf(torch.randn(N, K, device=DEV) * 0.5)
    bias = bf(torch.randn(N, device=DEV))
    res = bf(torch.randn(M, N, device=DEV))
    lin = (a.float() @ b.float().t() + bias.float()).bfloat16().float()
    assert torch.allclose(ops().gemm(a, b, bias=bias).float(), lin, rtol=2 ** -7, atol=2e-2)
    g = F.gelu(lin).bfloat16().float()
    assert torch.allclose(ops().gemm(a, b, bias=bias, act="gelu").float(), g, rtol=2 ** -7, atol=2e-2)
    r = (lin + res.float()).bfloat16().float()
    assert torch.allclose(ops().gemm(a, b, bias=bias, residual=res).float(), r, rtol=2 ** -7, atol=3e-2)
    # positional-embedding style residual: row m uses residual[m % res_mod]
    pos = bf(torch.randn(100, N, device=DEV))
    r2 = (a.float() @ b.float().t()).bfloat16().float() + pos.float()[torch.arange(M, device=DEV) % 100]
    assert torch.allclose(ops().gemm(a, b, residual=pos, res_mod=100).float(), r2, rtol=2 ** -6, atol=3e-2)
    # f32 output + accumulate (weight gradients)
    acc = torch.randn(M, N, device=DEV)
    want = acc + a.float() @ b.float().t()
    got = ops().gemm(a, b, out=acc.clone(), out_f32=True, accumulate=True)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-3)
    # strided A rows (the conv2 "3 consecutive frames" view): row stride 2*K' with K = 3*K'
    base = bf(torch.randn(2 * 40 + 2, 64, device=DEV))
    view = base.as_strided((40, 192), (128, 1))
    w = bf(torch.randn(128, 192, device=DEV))
    assert torch.allclose(ops().gemm(view, w).float(), view.float() @ w.float().t(), rtol=2 ** -7, atol=2e-2)


def test_gemm_rejects_bad_shapes():
    a = bf(torch.randn(8, 60, device=DEV))
    b = bf(torch.randn(8, 60, device=DEV))
    with pytest.raises(ValueError, match="multiple of 64"):
        ops().gemm(a, b)


# ------------------------------------------------------------------ norms
@pytest.mark.parametrize("rows,cols", [(7, 384), (300, 1024), (33, 1280), (1501, 512), (6, 4096)])
def test_layernorm(rows, cols):
    torch.manual_seed(1)
    x = bf(torch.randn(rows, cols, device=DEV) * 2 + 0.3)
    w, b = bf(torch.randn(cols, device=DEV)), bf(torch.randn(cols, device=DEV))
    ref = F.layer_norm(x.float(), (cols,), w.float(), b.float(), 1e-5)
    got = ops().layernorm(x, w, b, 1e-5)
    assert torch.allclose(got.float(), ref, rtol=2 ** -7, atol=2e-2)
    assert torch.equal(got, F.layer_norm(x, (cols,), w, b, 1e-5)) or rel_l2(got, ref) < 3e-3


@pytest.mark.parametrize("rows,cols", [(5, 2048), (64, 4096), (19, 8192)])
def test_rmsnorm_fwd_bwd(rows, cols):
    torch.manual_seed(2)
    x = bf(torch.randn(rows, cols, device=DEV))
    w = bf(0.4 + 0.1 * torch.randn(cols, device=DEV))
    dy = bf(torch.randn(rows, cols, device=DEV))
    xr = x.float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    h = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6)
    y = wr * h
    y.backward(dy.float())
    got = ops().rmsnorm(x, w, 1e-6)
    # LlamaRMSNorm rounds x_hat to bf16 before the weight multiply: restate that rounding
    ref = (w.float() * h.detach().bfloat16().float()).bfloat16()
    assert (got.float() - ref.float()).abs().max() <= 2 ** -7 * ref.float().abs().max()
    add = bf(torch.randn(rows, cols, device=DEV))
    dx, dw = ops().rmsnorm_bwd(dy, x, w, 1e-6, dx_add=add, want_dw=True)
    assert rel_l2(dx, xr.grad + add.float()) < 6e-3
    assert rel_l2(dw, wr.grad) < 6e-3
    dx_only = ops().rmsnorm_bwd(dy, x, w, 1e-6, dx_add=add, want_dw=False)       # the frozen-LLM variant (row cached in registers)
    dx_only = dx_only[0] if isinstance(dx_only, tuple) else dx_only
    assert rel_l2(dx_only, xr.grad + add.float()) < 6e-3


def test_stack_rmsnorm_is_pad_view_norm(golden_dir):
    import os
    from ultravox_amd import _lib
    import ctypes as C
    z = np.load(os.path.join(golden_dir, "projector_ln_mid.npz"))
    x = torch.from_numpy(z["x"]).to(DEV).bfloat16()          # [3, 21, 32] -> 3 rows of 8 frames, last has 5
    stacked_ref = torch.from_numpy(z["stacked"]).to(DEV)       # reference StackAudioFrames output (f32 input)
    # bit-exact re-indexing: stacking is a pure copy with zero padding
    from oracle.reference_cpu import rmsnorm_ref
    w = torch.from_numpy(z["w.ln_pre.weight"]).to(DEV).bfloat16()
    B, T, Cc = x.shape
    J = (T + 7) // 8
    y = torch.empty(B * J, Cc * 8, device=DEV, dtype=torch.bfloat16)
    st = torch.empty_like(y)
    # reach the fused kernel through the projector entry point's first stage: use rmsnorm on the stacked copy
    pad = F.pad(x, (0, 0, 0, J * 8 - T)).reshape(B * J, Cc * 8)
    assert torch.equal(pad.float(), stacked_ref.bfloat16().float().reshape(B * J, Cc * 8))
    got = ops().rmsnorm(pad.contiguous(), w, 1e-6)
    ref = rmsnorm_ref(pad.cpu(), w.cpu(), 1e-6)
    assert rel_l2(got.cpu(), ref) < 4e-3


# ------------------------------------------------------------------ SwiGLU / RoPE
@pytest.mark.parametrize("gate_first", [False, True])
def test_swiglu_fwd_bwd(gate_first):
    torch.manual_seed(3)
    x = bf(torch.randn(77, 512, device=DEV) * 2)
    dout = bf(torch.randn(77, 256, device=DEV))
    xr = x.float().requires_grad_(True)
    a, b = xr.chunk(2, dim=-1)
    val, gate = (b, a) if gate_first else (a, b)
    out = F.silu(gate) * val
    out.backward(dout.float())
    got = ops().swiglu(x, gate_first)
    assert torch.allclose(got.float(), out.detach(), rtol=2 ** -6, atol=1e-2)
    din = ops().swiglu_bwd(dout, x, gate_first)
    assert rel_l2(din, xr.grad) < 6e-3


def test_rope_forward_and_inverse():
    from ultravox_amd.config import TextConfig
    from ultravox_amd.weights import rope_table
    from oracle.reference_cpu import rope_cos_sin_ref, _rotate_half
    tc = TextConfig(hidden_size=512, num_attention_heads=4, num_key_value_heads=2, head_dim=128, rope_theta=500000.0)
    B, T, H = 2, 50, 6  # 4 q heads + 2 k heads rotated, 2 v heads untouched
    torch.manual_seed(4)
    x = bf(torch.randn(B * T, 8 * 128, device=DEV))
    x0 = x.clone()
    tab = rope_table(tc, 64, DEV)
    ops().rope_(x, tab, T, H, 128, inverse=False)
    cos, sin = rope_cos_sin_ref(tc, T, torch.bfloat16)
    xr = x0.cpu().view(B, T, 8, 128)
    want = xr.clone()
    rot = xr[:, :, :H]
    want[:, :, :H] = rot * cos[None, :, None, :] + _rotate_half(rot) * sin[None, :, None, :]
    got = x.cpu().view(B, T, 8, 128)
    assert torch.equal(got[:, :, H:], xr[:, :, H:])                      # v heads bit-identical
    assert (got.float() - want.float()).abs().max() <= 2 ** -6 * want.float().abs().max()
    ops().rope_(x, tab, T, H, 128, inverse=True)                          # R(-theta) R(theta) = I up to rounding
    assert rel_l2(x, x0) < 8e-3


# ------------------------------------------------------------------ attention
def sdpa_ref(q, k, v, causal, block, scale, kv_start=None, kv_len=None, window=0):
    B, T, Hq, D = q.shape
    Hkv = k.shape[2]
    qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
    kf = kf.repeat_interleave(Hq // Hkv, 1)
    vf = vf.repeat_interleave(Hq // Hkv, 1)
    s = qf @ kf.transpose(-1, -2) * scale
    idx = torch.arange(T, device=q.device)
    ok = torch.ones(B, 1, T, T, dtype=torch.bool, device=q.device)
    if causal:
        ok = ok & (idx[None, :] <= idx[:, None])[None, None]
    if block:
        ok = ok & ((idx[None, :] // block) <= (idx[:, None] // block))[None, None]
    if window:
        ok = ok & (idx[None, :] > idx[:, None] - window)[None, None]
    if kv_len is not None:
        ok = ok & (idx[None, None, None, :] < kv_len.view(-1, 1, 1, 1))
    if kv_start is not None:
        ok = ok & (idx[None, None, None, :] >= kv_start.view(-1, 1, 1, 1))
    s = s.masked_fill(~ok, float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, 0.0)
    return (p @ vf).transpose(1, 2).reshape(B, T, Hq * D), ok


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("D,Hq,Hkv,T,window", [(128, 4, 2, 700, 128), (128, 4, 1, 330, 100), (64, 4, 4, 513, 64), (256, 2, 2, 300, 77),
                                                (128, 8, 2, 316, 400)])
def test_sliding_window_attention_forward_backward(dtype, D, Hq, Hkv, T, window):
    """Causal attention with a sliding window (Gemma-3's local layers: a query sees keys in (q - window, q]) - forward and both
    backward kernels skip the key / query ranges no pair of which is visible and mask the boundary tiles; with left padding; a
    window that covers the sequence (last case) takes the plain-causal paths (incl. the fused backward)."""
    from ultravox_amd import ops
    torch.manual_seed(11)
    B = 2
    q, k, v = ((torch.randn(B, T, h, D, device=DEV)).to(dtype) for h in (Hq, Hkv, Hkv))
    do = torch.randn(B, T, Hq * D, device=DEV).to(dtype)
    kv_start = torch.tensor([0, 37], device=DEV, dtype=torch.int32)
    o, lse = ops.attention(q, k, v, causal=True, kv_start=kv_start, window=window)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ref, ok = sdpa_ref(qr, kr, vr, True, 0, D ** -0.5, kv_start=kv_start, window=window)
    rows = ok.any(-1)[:, 0]                                   # rows with at least one visible key (left-padded rows have none)
    bf = dtype == torch.bfloat16
    assert rel_l2(o[rows], ref.detach()[rows]) < (8e-3 if bf else 1e-5)
    dq, dk, dv = ops.attention_bwd(q, k, v, o, lse, do, causal=True, kv_start=kv_start, window=window)
    (ref * rows[..., None]).backward(do.float() * rows[..., None])
    tol = 2e-2 if bf else 2e-5
    assert rel_l2(dq[rows], qr.grad[rows]) < tol and rel_l2(dk, kr.grad) < tol and rel_l2(dv, vr.grad) < tol


@pytest.mark.parametrize("D,Hq,Hkv,T,causal,block", [(64, 4, 4, 200, False, 0), (64, 2, 2, 333, False, 50),
                                                     (128, 8, 2, 316, True, 0), (128, 4, 1, 70, True, 0),
                                                     (64, 6, 6, 1500, False, 0)])
def test_attention_forward(D, Hq, Hkv, T, causal, block):
    torch.manual_seed(5)
    B = 2
    qkv = bf(torch.randn(B, T, (Hq + 2 * Hkv) * D, device=DEV))
    q = qkv[..., : Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hkv) * D].view(B, T, Hkv, D)
    v = qkv[..., (Hq + Hkv) * D:].view(B, T, Hkv, D)
    kv_len = torch.tensor([T, max(1, T - 37)], device=DEV, dtype=torch.int32) if not causal else None
    o, lse = ops().attention(q, k, v, causal=causal, block=block, kv_len=kv_len)
    ref, _ = sdpa_ref(q, k, v, causal, block, D ** -0.5, kv_len=kv_len)
    # P is rounded to bf16 before P.V (as flash / SDPA kernels do) and O to bf16: 2^-7 of the row scale
    assert (o.float() - ref).abs().max().item() < 2e-2
    assert rel_l2(o, ref) < 8e-3


def test_attention_left_padding_and_rescale_branch():
    torch.manual_seed(6)
    B, T, Hq, Hkv, D = 2, 130, 4, 2, 128
    q = bf(torch.randn(B, T, Hq, D, device=DEV))
    k = bf(torch.randn(B, T, Hkv, D, device=DEV))
    v = bf(torch.randn(B, T, Hkv, D, device=DEV))
    k[0, 100] = q[0, 120, 0] * 4  # spike: the running max jumps in the second key block (online-softmax rescale)
    kv_start = torch.tensor([0, 17], device=DEV, dtype=torch.int32)
    kv_len = torch.tensor([T, T], device=DEV, dtype=torch.int32)
    o, _ = ops().attention(q, k, v, causal=True, kv_start=kv_start, kv_len=kv_len)
    ref, ok = sdpa_ref(q, k, v, True, 0, D ** -0.5, kv_start=kv_start, kv_len=kv_len)
    valid = ok.any(-1)[:, 0]  # rows with at least one visible key
    m = valid[:, :, None].expand(B, T, Hq * D)
    assert (o.float() - ref)[m].abs().max().item() < 3e-2
    assert torch.isfinite(o.float()).all()


@pytest.mark.parametrize("D,Hq,Hkv,T,causal,block", [(128, 8, 2, 316, True, 0), (128, 4, 4, 45, True, 0), (64, 4, 2, 130, True, 0),
                                                     # the Whisper encoder's masks (non-causal, key padding, latency blocks)
                                                     (64, 4, 4, 200, False, 0), (64, 2, 2, 333, False, 50), (64, 3, 3, 1500, False, 0)])
def test_attention_backward(D, Hq, Hkv, T, causal, block):
    torch.manual_seed(7)
    B = 2
    q = bf(torch.randn(B, T, Hq, D, device=DEV))
    k = bf(torch.randn(B, T, Hkv, D, device=DEV))
    v = bf(torch.randn(B, T, Hkv, D, device=DEV))
    do = bf(torch.randn(B, T, Hq * D, device=DEV))
    kv_len = torch.tensor([T, max(1, T - 37)], device=DEV, dtype=torch.int32) if not causal else None
    o, lse = ops().attention(q, k, v, causal=causal, block=block, kv_len=kv_len)
    dq, dk, dv = ops().attention_bwd(q, k, v, o, lse, do, causal=causal, block=block, kv_len=kv_len)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ref, _ = sdpa_ref(qr, kr, vr, causal, block, D ** -0.5, kv_len=kv_len)
    ref.backward(do.float())
    assert rel_l2(dq, qr.grad) < 2e-2 and rel_l2(dk, kr.grad) < 2e-2 and rel_l2(dv, vr.grad) < 2e-2
    if kv_len is not None:      # padded keys receive no gradient
        assert dk[1, int(kv_len[1]):].abs().max().item() == 0 and dv[1, int(kv_len[1]):].abs().max().item() == 0


@pytest.mark.parametrize("T,Hq,Hkv", [(316, 8, 2), (200, 4, 4), (129, 4, 1)])
def test_attention_backward_causal_left_and_right_padding(T, Hq, Hkv):
    """The LLM's backward kernels (head_dim 128: 128 keys / 128 queries per block, two-deep prefetch) with the collator's
    padding on both sides: sequence 1 starts at key 23 (left padding) and sequence 2 ends 41 keys early (right padding);
    block edges at 128 fall inside all three lengths.  Keys outside the valid range get exactly zero gradient."""
    torch.manual_seed(17)
    B, D = 3, 128
    q = bf(torch.randn(B, T, Hq, D, device=DEV))
    k = bf(torch.randn(B, T, Hkv, D, device=DEV))
    v = bf(torch.randn(B, T, Hkv, D, device=DEV))
    do = bf(torch.randn(B, T, Hq * D, device=DEV))
    kv_start = torch.tensor([0, 23, 0], device=DEV, dtype=torch.int32)
    kv_len = torch.tensor([T, T, T - 41], device=DEV, dtype=torch.int32)
    o, lse = ops().attention(q, k, v, causal=True, kv_start=kv_start, kv_len=kv_len)
    dq, dk, dv = ops().attention_bwd(q, k, v, o, lse, do, causal=True, kv_start=kv_start, kv_len=kv_len)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ref, ok = sdpa_ref(qr, kr, vr, True, 0, D ** -0.5, kv_start=kv_start, kv_len=kv_len)
    valid = ok.any(-1)[:, 0]                                        # query rows that see at least one key
    m = valid[:, :, None].expand(B, T, Hq * D)
    ref.backward(do.float() * m)         # (rows that see no key: the kernels' output there is unspecified, as in SDPA)
    dqm = dq.float() * valid[:, :, None, None]
    assert rel_l2(dqm, qr.grad) < 2e-2 and rel_l2(dk, kr.grad) < 2e-2 and rel_l2(dv, vr.grad) < 2e-2
    assert dk[1, :23].abs().max().item() == 0 and dv[1, :23].abs().max().item() == 0
    assert dk[2, T - 41:].abs().max().item() == 0 and dv[2, T - 41:].abs().max().item() == 0
    again = ops().attention_bwd(q, k, v, o, lse, do, causal=True, kv_start=kv_start, kv_len=kv_len)
    assert all(torch.equal(a, b) for a, b in zip((dq, dk, dv), again))          # fixed reduction order


# ------------------------------------------------------------------ loss / optimizer / merge
def test_ce_loss_and_gradient():
    torch.manual_seed(8)
    B, T, V = 3, 40, 32000
    logits = bf(torch.randn(B, T, V, device=DEV) * 2)
    labels = torch.randint(0, V, (B, T), device=DEV)
    labels[:, :25] = -100
    labels[1, 30] = -100
    lr = logits.float().requires_grad_(True)
    shifted = F.pad(labels, (0, 1), value=-100)[:, 1:]
    ref = F.cross_entropy(lr.view(-1, V), shifted.reshape(-1), ignore_index=-100)
    ref.backward()
    loss, dl = ops().ce_loss(logits, labels)
    assert abs(loss.item() - ref.item()) < 1e-4 * abs(ref.item())          # f32 loss
    assert rel_l2(dl, lr.grad) < 6e-3
    ignored = (shifted == -100)
    assert dl[ignored].abs().max().item() == 0.0                            # ignored rows: exactly zero
    # in place over the logits, scaled
    loss2, dl2 = ops().ce_loss(logits.clone(), labels, grad_scale=0.5, in_place=True)
    assert rel_l2(dl2, 0.5 * lr.grad) < 6e-3
    # no valid label -> mean over an empty set is NaN (torch semantics)
    loss3, _ = ops().ce_loss(logits, torch.full_like(labels, -100), want_grad=False)
    assert math.isnan(loss3.item())


def test_embed_merge_is_bit_exact_last_writer_wins():
    from ultravox_amd import _lib
    import ctypes as C
    from oracle.reference_cpu import merge_ref
    torch.manual_seed(9)
    B, T, D, Na, n_items, V = 3, 50, 64, 12, 5, 100
    table = bf(torch.randn(V, D, device=DEV))
    ids = torch.randint(0, V, (B, T), device=DEV)
    audio = bf(torch.randn(n_items, Na, D, device=DEV))
    start = torch.tensor([3, 10, 0, 45, 20], device=DEV)            # item 1 overlaps item 0; item 3 clipped by len
    tok_len = torch.tensor([12, 9, 7, 5, 0], device=DEV, dtype=torch.int32)
    bsz = torch.tensor([2, 2, 1], device=DEV)
    cfg = _lib.Config(); cfg.dtype = 0; cfg.llm_d = D; cfg.vocab = V
    out = torch.empty(B, T, D, device=DEV, dtype=torch.bfloat16)
    scratch = torch.empty(B * T + n_items, device=DEV, dtype=torch.int32)
    _lib.check(_lib.lib().uvx_embed_merge(_lib.stream_ptr(), C.byref(cfg), _lib.ptr(table), _lib.ptr(ids), _lib.ptr(audio),
                                          _lib.ptr(bsz), _lib.ptr(start), _lib.ptr(tok_len), B, T, n_items, Na,
                                          _lib.ptr(out), _lib.ptr(scratch)))
    want = merge_ref(F.embedding(ids, table).cpu(), audio.cpu(), start.cpu(), tok_len.cpu(), bsz.cpu())
    assert torch.equal(out.cpu(), want)
    # backward = gather of exactly the rows each item still owns
    g = bf(torch.randn(B, T, D, device=DEV))
    da = torch.empty_like(audio)
    _lib.check(_lib.lib().uvx_merge_embeds_bwd(_lib.stream_ptr(), C.byref(cfg), _lib.ptr(g), _lib.ptr(start),
                                               _lib.ptr(tok_len), B, T, n_items, Na, _lib.ptr(da), _lib.ptr(scratch)))
    a_req = audio.float().cpu().requires_grad_(True)
    merged = merge_ref(torch.zeros(B, T, D), a_req, start.cpu(), tok_len.cpu(), bsz.cpu())
    merged.backward(g.float().cpu())
    assert torch.equal(da.float().cpu(), a_req.grad)


@pytest.mark.parametrize("mode", ["bf16_state", "f32_master", "f32"])
def test_adamw_clip_step_matches_torch(mode):
    from ultravox_amd import _lib
    import ctypes as C
    torch.manual_seed(10)
    n = 50000
    dt = torch.float32 if mode == "f32" else torch.bfloat16
    p0 = (torch.randn(n, device=DEV) * 0.1).to(dt)
    grads = [torch.randn(n, device=DEV) * s for s in (0.05, 0.002, 1.0)]  # with and without clipping
    p_ref = torch.nn.Parameter(p0.clone().float() if mode != "bf16_state" else p0.clone())
    opt = torch.optim.AdamW([p_ref], lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, foreach=False)
    p = p0.clone()
    master = p0.float().clone() if mode == "f32_master" else None
    sdt = torch.float32 if mode != "bf16_state" else torch.bfloat16
    m, v = torch.zeros(n, device=DEV, dtype=sdt), torch.zeros(n, device=DEV, dtype=sdt)
    scratch = torch.zeros(1025, device=DEV)
    for step, g in enumerate(grads, 1):
        p_ref.grad = g.to(p_ref.dtype).clone()
        torch.nn.utils.clip_grad_norm_([p_ref], 1.0)
        opt.step()
        _lib.check(_lib.lib().uvx_adamw_clip_step(_lib.stream_ptr(), _lib.dtype_code(dt), _lib.ptr(p), _lib.ptr(master),
                                                  _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), C.c_int64(n), C.c_float(1.0),
                                                  C.c_float(2e-3), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8),
                                                  C.c_float(0.0), step, _lib.ptr(scratch)))
        assert abs(scratch[0].sqrt().item() - g.norm().item()) < 1e-3 * g.norm().item()
    if mode == "f32_master":  # the update runs on the f32 master copy; the bf16 parameter mirrors it
        assert rel_l2(master, p_ref.detach().float()) < 1e-5
        assert torch.equal(p, master.bfloat16())
    else:
        assert rel_l2(p.float(), p_ref.detach().float()) < (1e-5 if mode == "f32" else 2e-2)


# ------------------------------------------------------------------ K1 log-mel
@pytest.mark.parametrize("n_mels", [80, 128])
def test_logmel_matches_hf_fixture(golden_dir, n_mels):
    import os
    from ultravox_amd.frontend import WhisperFeatureExtractor
    z = np.load(os.path.join(golden_dir, "logmel.npz"))
    fe = WhisperFeatureExtractor(feature_size=n_mels)
    out = fe(list(z[f"pcm_{n_mels}"]), sampling_rate=16000, padding="longest", pad_to_multiple_of=160)
    got = out["input_features"].cpu().numpy()
    want = z[f"mel_{n_mels}"]
    assert got.shape == want.shape
    # dense f32 DFT (a fixed-order fmaf chain on the f32 matrix cores) vs HF's FFT: a numpy emulation of the same chain agrees
    # with the fixture to 1e-5 on every clip, the tonal one included, and so does the device (measured on MI355X, round 3:
    # 2.5e-6 / 1.7e-6 on the noise clips, 1.3e-5 on the tonal clip at 80 mels; 4.4e-6 / 2.6e-6 / 2.2e-5 at 128 -
    # profiles/r03_parity/logmel_fixture_errors_*.json).  Round 2's 5e-3 bound on the tonal clip was slack, not error.
    from parity_util import record
    errs = [float(np.abs(got[i] - want[i]).max()) for i in range(3)]
    record(f"logmel_fixture_errors_{n_mels}", {"max_abs_noise": errs[0], "max_abs_quiet_noise": errs[1], "max_abs_tonal": errs[2]})
    assert max(errs[:2]) < 5e-5
    assert errs[2] < 1e-4, errs
    assert out["attention_mask"].sum(-1).tolist() == [200, 200, 200]


@pytest.mark.parametrize("n_mels", [80, 128])
def test_logmel_speech_like_audio_within_1e3(golden_dir, n_mels):
    """The clip class speech belongs to: 30 s of harmonics with pauses (62 % of the bins at the per-clip floor, the rest spread
    over 8 decades) against the installed HF extractor (tests/golden/logmel_speech.npz) - 2e-4 absolute on every bin (north_star asks 1e-3)."""
    import os
    import forward_fixture_util as U
    from parity_util import record
    from ultravox_amd.frontend import WhisperFeatureExtractor
    z = np.load(os.path.join(golden_dir, "logmel_speech.npz"))
    pcm = torch.from_numpy(U.speech_like_pcm())[None].to(DEV)
    got = WhisperFeatureExtractor(n_mels).logmel_device(pcm)[0].cpu().numpy()
    want = z[f"mel_{n_mels}_every4"]
    d = np.abs(got[:, ::4] - want)
    record(f"logmel_speech_errors_{n_mels}", {"max_abs": float(d.max()), "mean_abs": float(d.mean()),
                                              "frac_above_1e-4": float((d > 1e-4).mean()), "floor_frac": float((want == want.min()).mean())})
    assert got.shape == (n_mels, 3000)
    assert d.max() < 2e-4, float(d.max())            # measured 2.1e-5 (80 mels) / 3.7e-5 (128 mels)


def test_logmel_full_size_properties():
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from oracle.reference_cpu import logmel_ref
    fe = WhisperFeatureExtractor(80)
    g = torch.Generator().manual_seed(1234)
    pcm = (0.1 * torch.randn(4, 480000, generator=g)).clamp_(-1, 1)
    got = fe.logmel_device(pcm.to(DEV)).cpu()
    assert got.shape == (4, 80, 3000)
    want = logmel_ref(pcm[:1], 80)
    assert (got[:1] - want).abs().max().item() < 2e-4
    # per-clip affine map: max over the clip minus min is at most 8/4, and scaling the waveform shifts by log10
    assert ((got.amax((1, 2)) - got.amin((1, 2))) <= 2.0 + 1e-6).all()
    got2 = fe.logmel_device((pcm * 0.5).to(DEV)).cpu()
    assert (got2 - (got + 2 * math.log10(0.5) / 4)).abs().max().item() < 1e-3


def _interleave_gate_up(wg, wu):
    """weights.py packing: alternating 16-row blocks of gate / up rows."""
    I, K = wg.shape
    return torch.stack([wg.view(I // 16, 16, K), wu.view(I // 16, 16, K)], 1).reshape(2 * I, K)


@pytest.mark.parametrize("M,I,K", [(300, 512, 256), (2528, 1024, 512), (77, 96, 128)])
def test_gemm_fused_swiglu_epilogues_match_unfused_path(M, I, K):
    """Epilogue 1 (gate|up GEMM + SwiGLU) and epilogue 2 (down-projection dgrad + SwiGLU backward) are bit-identical to
    the GEMM followed by the elementwise kernels they replace (same bf16 rounding points)."""
    from ultravox_amd import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    x = (torch.randn(M, K, device=DEV, generator=g) * 0.5).bfloat16()
    wg = (torch.randn(I, K, device=DEV, generator=g) * 0.1).bfloat16()
    wu = (torch.randn(I, K, device=DEV, generator=g) * 0.1).bfloat16()
    wgu = _interleave_gate_up(wg, wu)
    # forward
    gu_ref = ops.gemm(x, wgu)
    act_ref = ops.swiglu(gu_ref, gate_first=2)
    act = torch.empty(M, I, device=DEV, dtype=torch.bfloat16)
    gu = ops.gemm(x, wgu, epilogue=1, c2=act)
    assert torch.equal(gu, gu_ref) and torch.equal(act, act_ref)
    ref = (F.silu((x.float() @ wg.float().t()).bfloat16().float()).bfloat16().float() * (x.float() @ wu.float().t()).bfloat16().float())
    assert rel_l2(act, ref) < 1e-2
    # backward: d act = dy . Wd (as an NT GEMM on Wd^T [I, D]), then SwiGLU backward
    D = 256
    dy = (torch.randn(M, D, device=DEV, generator=g) * 0.5).bfloat16()
    wd_t = (torch.randn(I, D, device=DEV, generator=g) * 0.1).bfloat16()
    d_act = ops.gemm(dy, wd_t)
    dgu_ref = ops.swiglu_bwd(d_act, gu_ref, gate_first=2)
    dgu = torch.empty(M, 2 * I, device=DEV, dtype=torch.bfloat16)
    ops.gemm(dy, wd_t, out=dgu, epilogue=2, c2=gu_ref)
    assert torch.equal(dgu, dgu_ref)


@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (8, 6144, 4096), (5, 100, 64), (16, 1028, 14336), (3, 32, 128),
                                   (2, 8192, 28672), (4, 4096, 14336), (1, 57344, 8192), (3, 96, 576), (7, 96, 576), (9, 64, 1024),
                                   (24, 6144, 4096), (33, 1028, 14336), (64, 4096, 8192), (17, 64, 2048)])
def test_gemm_few_rows_weight_streaming_kernel(M, N, K):
    """Few rows (the decode step) run the weight-streaming kernels (csrc/gemm_skinny.hip: the row-streaming kernel for M <= 2, the
    MFMA mapping for 3..16 - and, with the weights staged through LDS, up to 64 rows when K % 2048 == 0): against torch, against the tiled kernels (option 4 off) and against each other (option 4 = 2) with every
    epilogue the decode path uses; K that is not a multiple of the 512-element step, ragged N, the 70B shapes."""
    from ultravox_amd import _lib
    g = torch.Generator(device=DEV).manual_seed(11)
    a = (torch.randn(M, K, device=DEV, generator=g) * 0.5).bfloat16()
    b = (torch.randn(N, K, device=DEV, generator=g) * 0.1).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g).bfloat16()
    resid = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    L = _lib.lib()
    out = ops().gemm(a, b, bias=bias, residual=resid, act="gelu")
    L.uvx_set_option(4, 0)
    try:
        tiled = ops().gemm(a, b, bias=bias, residual=resid, act="gelu")
        plain_tiled = ops().gemm(a, b)
    finally:
        L.uvx_set_option(4, 1)
    ref = F.gelu((a.float() @ b.float().t() + bias.float()).bfloat16().float()).bfloat16().float() + resid.float()
    assert rel_l2(out, ref) < 5e-3 and rel_l2(out, tiled) < 5e-3
    assert rel_l2(ops().gemm(a, b), plain_tiled) < 3e-3
    L.uvx_set_option(4, 2)          # the MFMA mapping for every M <= 16
    try:
        assert rel_l2(out, ops().gemm(a, b, bias=bias, residual=resid, act="gelu")) < 5e-3
    finally:
        L.uvx_set_option(4, 1)
    if N % 32 == 0:     # fused SwiGLU epilogue (interleaved gate / up packing)
        act = torch.empty(M, N // 2, device=DEV, dtype=torch.bfloat16)
        gu = ops().gemm(a, b, epilogue=1, c2=act)
        L.uvx_set_option(4, 0)
        try:
            act_t = torch.empty_like(act)
            gu_t = ops().gemm(a, b, epilogue=1, c2=act_t)
        finally:
            L.uvx_set_option(4, 1)
        assert rel_l2(gu, gu_t) < 3e-3 and rel_l2(act, act_t) < 6e-3


@pytest.mark.parametrize("M,N,K,flavor,swiglu", [(1, 10240, 8192, 0, False), (4, 6144, 4096, 0, False), (2, 28672, 4096, 0, True),
                                                  (1, 57344, 8192, 0, True), (2, 4096, 3072, 1, False), (3, 4096, 3072, 1, False), (8, 6144, 4096, 0, False),
                                                  (4, 1024, 8192, 0, False), (8, 57344, 8192, 0, True), (16, 10240, 8192, 0, False),
                                                  (5, 4096, 4096, 1, False), (12, 28672, 4096, 0, True), (3, 10240, 8192, 0, False)])
def test_gemm_with_fused_rmsnorm_matches_the_two_launches(M, N, K, flavor, swiglu):
    """uvx_gemm_rmsnorm (the decode step's input_layernorm -> q|k|v and post_attention_layernorm -> gate|up in one launch: M <= 2 in the
    row-streaming kernel; round 6, opt-in through option 24 = 1: M = 3..16 with K % 2048 == 0 in the staged MFMA kernel) against rmsnorm + gemm
    as two launches: the same rounding points, so the outputs agree to the last-bit noise of a differently ordered sum of squares (rel-L2 <
    2e-3, almost every element identical); other shapes (K = 3072) take the two-launch fallback inside the entry point."""
    from ultravox_amd import _lib
    _lib.lib().uvx_set_option(24, 1)
    try:
        _fused_rmsnorm_case(M, N, K, flavor, swiglu)
    finally:
        _lib.lib().uvx_set_option(24, 0)


def _fused_rmsnorm_case(M, N, K, flavor, swiglu):
    g = torch.Generator(device=DEV).manual_seed(13)
    a = (torch.randn(M, K, device=DEV, generator=g) * 1.7).bfloat16()
    w = (1.0 + 0.2 * torch.randn(K, device=DEV, generator=g)).bfloat16()
    b = (torch.randn(N, K, device=DEV, generator=g) * 0.05).bfloat16()
    eps = 1e-5
    if flavor:      # GemmaRMSNorm (x_hat * (1 + w), one rounding) is not among the single-op wrappers: restated in f32
        x = a.float()
        normed = ((x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)) * (1.0 + w.float())).bfloat16()
    else:
        normed = ops().rmsnorm(a, w, eps=eps)
    if swiglu:
        act2 = torch.empty(M, N // 2, device=DEV, dtype=torch.bfloat16)
        want = ops().gemm(normed, b, epilogue=1, c2=act2)
        act1 = torch.empty_like(act2)
        got = ops().gemm_rmsnorm(a, w, b, eps=eps, flavor=flavor, epilogue=1, c2=act1)
        assert rel_l2(act1, act2) < 4e-3
    else:
        bias = torch.randn(N, device=DEV, generator=g).bfloat16()
        want = ops().gemm(normed, b, bias=bias)
        got = ops().gemm_rmsnorm(a, w, b, eps=eps, flavor=flavor, bias=bias)
    assert rel_l2(got, want) < 2e-3, rel_l2(got, want)
    assert (got == want).float().mean().item() > (0.9 if flavor == 0 else 0.6)
    ref = normed.float() @ b.float().t()
    if not swiglu:
        assert rel_l2(got, (ref + bias.float())) < 5e-3


def test_fused_rmsnorm_at_3_to_16_rows_is_opt_in():
    """Default (option 24 = 0): the 3..16-row problem runs rmsnorm + gemm exactly as the caller would (bit-identical to the two launches);
    option 24 = 1 takes the staged kernel with the norm inside."""
    g = torch.Generator(device=DEV).manual_seed(5)
    a = (torch.randn(8, 4096, device=DEV, generator=g) * 1.3).bfloat16()
    w = (1.0 + 0.2 * torch.randn(4096, device=DEV, generator=g)).bfloat16()
    b = (torch.randn(6144, 4096, device=DEV, generator=g) * 0.05).bfloat16()
    want = ops().gemm(ops().rmsnorm(a, w, eps=1e-5), b)
    from ultravox_amd import _lib
    L = _lib.lib()
    got = ops().gemm_rmsnorm(a, w, b, eps=1e-5)
    assert torch.equal(got, want)
    L.uvx_set_option(24, 1)
    try:
        fused = ops().gemm_rmsnorm(a, w, b, eps=1e-5)
    finally:
        L.uvx_set_option(24, 0)
    assert rel_l2(fused, want) < 2e-3 and (fused == want).float().mean().item() > 0.9


_A4_CHECK = r"""
import sys, torch
from ultravox_amd import _lib, ops
L = _lib.lib(); DEV = "cuda"; variant = int(sys.argv[1])
g = torch.Generator(device=DEV).manual_seed(variant)
def run(v, fn):
    L.uvx_gemm_force_variant(v)
    try: return fn()
    finally: L.uvx_gemm_force_variant(-1)
for (M, N, K) in [(256, 256, 64), (300, 520, 192), (2528, 4096, 256), (1000, 1032, 640), (2528, 6144, 4096)]:
    a = (torch.randn(M, K, device=DEV, generator=g) * 0.5).bfloat16()
    b = (torch.randn(N, K, device=DEV, generator=g) * 0.5).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g).bfloat16()
    resid = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    modes = {"plain": lambda: ops.gemm(a, b), "bias+res": lambda: ops.gemm(a, b, bias=bias, residual=resid),
             "bias+gelu": lambda: ops.gemm(a, b, bias=bias, act="gelu"), "f32": lambda: ops.gemm(a, b, out_f32=True)}
    for name, fn in modes.items():
        want, got = run(31, fn), run(variant, fn)
        assert torch.equal(got, want), (variant, (M, N, K), name, int((got != want).sum()))
    first = run(variant, modes["plain"])
    for _ in range(5):
        assert torch.equal(run(variant, modes["plain"]), first), (variant, (M, N, K), "repeat")
    ref = a.float() @ b.float().t()
    assert ((first.float() - ref).norm() / ref.norm()).item() < 5e-3
print("A4-OK")
"""


@pytest.mark.parametrize("variant", [43, 49, 55])
def test_hand_scheduled_gemm_loops_are_bit_identical_to_the_production_kernel(variant):
    """Round 4's inline-asm K loops (43 = four waves x 128 x 128, 49 = eight free-running waves, 55 = eight waves ping-pong; generated by
    tools/gen_gemm_a4.py) share the production kernel's MFMA, operand roles and k order: every epilogue they serve is BIT-identical to the
    merged-phase 256 x 256 kernel (variant 31) on ragged / one-K-tile / deep-K shapes, and repeated launches are bit-identical (the
    full screen, 12 shapes x 7 epilogues x 30 repeats, is tools/gpu_gemm_a4_check.py).  Round 5: they are a record, never picked, so
    they live in libuvx_probes.so only - the check runs in a child process that loads that library (UVX_LIB) and doubles as the
    compiler-change alarm for the literal-AGPR accumulators."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probes = os.path.join(root, "ultravox_amd", "libuvx_probes.so")
    if not os.path.exists(probes):
        pytest.skip("libuvx_probes.so not built (python -m ultravox_amd.build --probes)")
    from ultravox_amd import _lib
    _lib.lib().uvx_gemm_force_variant(variant)
    try:       # the product library refuses them
        x = torch.zeros(256, 64, device=DEV, dtype=torch.bfloat16)
        with pytest.raises(ValueError, match="libuvx_probes.so"):
            ops().gemm(x, x)
    finally:
        _lib.lib().uvx_gemm_force_variant(-1)
    env = dict(os.environ, UVX_LIB=probes, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", _A4_CHECK, str(variant)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "A4-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _splitk_ref(a, b, bias=None, resid=None, gelu=False):
    t = a.float() @ b.float().t()
    if bias is not None:
        t = t + bias.float()
    t = t.bfloat16().float()
    if gelu:
        t = F.gelu(t).bfloat16().float()
    if resid is not None:
        t = (t + resid.float())
    return t


@pytest.mark.parametrize("M", [65, 188, 316, 632])
@pytest.mark.parametrize("N,K", [(4096, 4096), (6144, 4096), (4096, 14336), (8192, 8192), (10240, 8192), (8192, 28672)])
def test_splitk_gemm_prefill_shapes_match_the_f32_product(M, N, K):
    """Round 5: the prefill's GEMMs (M = one or two prompts' rows; N, K = the q|k|v / o / down shapes of Llama-3-8B and Llama-3.3-70B)
    through uvx_gemm_splitk - the cost model's own (tile, split) choice - against the f32 product of the same bf16 operands: one
    bf16 rounding of an f32 sum (+ summation order), i.e. the bar of test_gemm_matches_fp32_matmul; against the unsplit uvx_gemm the
    two agree to the f32 summation order (rel-L2 <= 1e-3, >= 97 % of the elements identical), and repeated calls are bit-identical
    (fixed slab order in the reduce kernel)."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = (torch.randn(M, K, device=DEV, generator=g) * 0.5).bfloat16()
    b = (torch.randn(N, K, device=DEV, generator=g) * 0.5).bfloat16()
    resid = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    got = ops().gemm_splitk(a, b, residual=resid)
    pre = a.float() @ b.float().t()
    ref = _splitk_ref(a, b, resid=resid)
    # two roundings (the product, then the sum with the residual): one bf16 ulp of each, + the f32 summation order
    assert ((got.float() - ref).abs() <= 2 ** -7 * (pre.abs() + ref.abs()) + 1e-3 * math.sqrt(K)).all()
    plain = ops().gemm(a, b, residual=resid)
    assert rel_l2(got, plain) < 1e-3 and (got == plain).float().mean().item() > 0.97
    for _ in range(3):
        assert torch.equal(ops().gemm_splitk(a, b, residual=resid), got)


@pytest.mark.parametrize("s", [2, 3, 5, 8, 16])
@pytest.mark.parametrize("variant", [0, 31, 32, 33, 34, 59, 60])
def test_splitk_gemm_every_tile_and_factor(variant, s):
    """Every production tile x forced split factors (incl. factors that do not divide the K-tile count: the ranges then differ by one
    K-tile) x every epilogue the reduce kernel restates (bias, GELU, residual, positional residual, alpha, fused SwiGLU) on a ragged
    problem: same bar against the f32 reference as the unsplit kernel, and against the unsplit kernel itself to summation order."""
    from ultravox_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device=DEV).manual_seed(100 * variant + s)
    M, N, K = 316, 1096, 64 * 37
    a = (torch.randn(M, K, device=DEV, generator=g) * 0.5).bfloat16()
    b = (torch.randn(N, K, device=DEV, generator=g) * 0.5).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g).bfloat16()
    resid = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    pos = torch.randn(100, N, device=DEV, generator=g).bfloat16()
    L.uvx_gemm_force_variant(variant)
    try:
        cases = {"plain": dict(), "bias": dict(bias=bias), "bias+gelu": dict(bias=bias, act="gelu"),
                 "bias+res": dict(bias=bias, residual=resid), "pos": dict(residual=pos, res_mod=100), "alpha": dict(alpha=0.25)}
        for name, kw in cases.items():
            got = ops().gemm_splitk(a, b, force_split=s, **kw)
            want = ops().gemm(a, b, **kw)
            assert rel_l2(got, want) < 1.5e-3, (name, rel_l2(got, want))
            assert (got == want).float().mean().item() > 0.95, name
        ref = _splitk_ref(a, b, bias=bias, resid=resid)
        pre = a.float() @ b.float().t() + bias.float()
        got = ops().gemm_splitk(a, b, force_split=s, bias=bias, residual=resid)
        assert ((got.float() - ref).abs() <= 2 ** -7 * (pre.abs() + ref.abs()) + 3e-2).all()
        # fused SwiGLU epilogue: gate|up pre-activations + silu(gate) * up (interleaved 16-column blocks)
        I = 1504
        wgu = (torch.randn(2 * I, K, device=DEV, generator=g) * 0.05).bfloat16()
        act_s = torch.empty(M, I, device=DEV, dtype=torch.bfloat16)
        act_p = torch.empty_like(act_s)
        gu_s = ops().gemm_splitk(a, wgu, force_split=s, epilogue=1, c2=act_s)
        gu_p = ops().gemm(a, wgu, epilogue=1, c2=act_p)
        assert rel_l2(gu_s, gu_p) < 1.5e-3 and rel_l2(act_s, act_p) < 3e-3
        assert torch.equal(act_s, ops().swiglu(gu_s, gate_first=2))       # the activation is exactly that of the stored pre-activations
    finally:
        L.uvx_gemm_force_variant(-1)


@pytest.mark.parametrize("variant,twin", [(59, 33), (60, 34)])
def test_three_buffer_merged_phase_kernels_are_bit_identical_to_their_twins(variant, twin):
    """Round 5: the merged-phase kernel with THREE LDS buffer sets (59 = 160 x 256, 60 = 128 x 256: twice the DMA look-ahead, for the
    prefill's HBM-fed problems) issues the same MFMAs on the same operands in the same k order as its two-set twin: bit-identical on
    one-, two-, three-K-tile loops (shorter than the pipeline), ragged shapes, deep K, every epilogue; 20 repeats bit-identical (race screen)."""
    from ultravox_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device=DEV).manual_seed(variant)

    def run(v, fn):
        L.uvx_gemm_force_variant(v)
        try:
            return fn()
        finally:
            L.uvx_gemm_force_variant(-1)

    for (M, N, K) in [(160, 256, 64), (316, 520, 128), (300, 256, 192), (316, 1032, 256), (632, 4096, 4096), (316, 8192, 8192), (188, 768, 28672)]:
        a = (torch.randn(M, K, device=DEV, generator=g) * 0.5).bfloat16()
        b = (torch.randn(N, K, device=DEV, generator=g) * 0.5).bfloat16()
        bias = torch.randn(N, device=DEV, generator=g).bfloat16()
        resid = torch.randn(M, N, device=DEV, generator=g).bfloat16()
        modes = {"plain": lambda: ops().gemm(a, b), "bias+res": lambda: ops().gemm(a, b, bias=bias, residual=resid),
                 "bias+gelu": lambda: ops().gemm(a, b, bias=bias, act="gelu"), "f32": lambda: ops().gemm(a, b, out_f32=True),
                 "split3": lambda: ops().gemm_splitk(a, b, residual=resid, force_split=3 if K >= 192 else 1)}
        for name, fn in modes.items():
            want, got = run(twin, fn), run(variant, fn)
            assert torch.equal(got, want), (variant, (M, N, K), name, int((got != want).sum()))
        first = run(variant, modes["plain"])
        for _ in range(20):
            assert torch.equal(run(variant, modes["plain"]), first), (variant, (M, N, K), "repeat")
        assert rel_l2(first, a.float() @ b.float().t()) < 5e-3


def test_splitk_gemm_falls_back_to_the_plain_kernel():
    """No scratch, a forced factor of 1, f32 output or a problem with enough tiles: uvx_gemm_splitk is uvx_gemm, bit for bit."""
    g = torch.Generator(device=DEV).manual_seed(5)
    a = (torch.randn(2528, 512, device=DEV, generator=g) * 0.5).bfloat16()
    b = (torch.randn(4096, 512, device=DEV, generator=g) * 0.5).bfloat16()
    assert torch.equal(ops().gemm_splitk(a, b), ops().gemm(a, b))                       # 256 tiles: nothing to split
    a2, b2 = a[:316].contiguous(), b[:1024].contiguous()
    assert torch.equal(ops().gemm_splitk(a2, b2, force_split=1), ops().gemm(a2, b2))
    assert torch.equal(ops().gemm_splitk(a2, b2, workspace=torch.empty(0, device=DEV, dtype=torch.uint8)), ops().gemm(a2, b2))


def test_lds_transpose_read_semantics():
    """ds_read_b64_tr_b16 (gfx950), the instruction the bf16 attention kernels use to read V^T / Q^T / K^T / dO^T out of the
    natural [row][d] LDS tiles: every lane supplies the address of one 8-byte chunk; within each group of 16 lanes, result
    element j of lane i is element (i & 3) of the chunk supplied by lane 4*j + (i >> 2).  Pinned for distinct, permuted and
    strided per-lane addresses (uvx_probe_lds_tr: the LDS image holds value e at 16-bit element e)."""
    import ctypes as C
    from ultravox_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(0)
    cases = [torch.arange(64, dtype=torch.int32) * 8,                                   # consecutive chunks
             torch.randperm(64, generator=g).to(torch.int32) * 8,                       # any chunk per lane
             (torch.arange(64, dtype=torch.int32) // 4) * 256 + (torch.arange(64, dtype=torch.int32) % 4) * 8 + 1024]   # rows of a 256-byte-pitch tile
    for addr in cases:
        a = addr.to(DEV)
        out = torch.zeros(256, dtype=torch.int32, device=DEV)
        _lib.check(L.uvx_probe_lds_tr(None, C.c_void_p(a.data_ptr()), C.c_void_p(out.data_ptr())), "uvx_probe_lds_tr")
        torch.cuda.synchronize()
        got = out.cpu().view(64, 4)
        want = torch.empty(64, 4, dtype=torch.int32)
        for lane in range(64):
            grp, i = lane // 16 * 16, lane % 16
            for j in range(4):
                want[lane, j] = addr[grp + 4 * j + (i >> 2)] // 2 + (i & 3)
        assert torch.equal(got, want), (addr[:8], got[:4], want[:4])


@pytest.mark.parametrize("D,Hq,Hkv,T,causal,block", [(128, 8, 2, 316, True, 0), (64, 3, 3, 700, False, 50), (256, 2, 1, 150, True, 0)])
def test_attention_transposing_lds_reads_equal_the_transposed_copies(D, Hq, Hkv, T, causal, block):
    """Option 12 (default on): natural tiles + ds_read_b64_tr_b16 instead of V^T / Q^T / K^T / dO^T copies in global memory.
    Same operands in the same contraction order: forward output, log-sum-exp and all three gradients are bit-identical."""
    from ultravox_amd import _lib
    torch.manual_seed(11)
    B = 2
    q = bf(torch.randn(B, T, Hq, D, device=DEV)); k = bf(torch.randn(B, T, Hkv, D, device=DEV)); v = bf(torch.randn(B, T, Hkv, D, device=DEV))
    do = bf(torch.randn(B, T, Hq * D, device=DEV))
    kv_len = torch.tensor([T, T - 29], device=DEV, dtype=torch.int32)

    def run():
        o, lse = ops().attention(q, k, v, causal=causal, block=block, kv_len=kv_len)
        return (o, lse) + tuple(ops().attention_bwd(q, k, v, o, lse, do, causal=causal, block=block, kv_len=kv_len))

    _lib.lib().uvx_set_option(13, 0)          # the kernel PAIR on both sides (the fused backward sums in another order)
    try:
        new = run()
        _lib.lib().uvx_set_option(12, 0)
        try:
            old = run()
        finally:
            _lib.lib().uvx_set_option(12, 1)
    finally:
        _lib.lib().uvx_set_option(13, 1)
    for a, b in zip(new, old):
        assert torch.equal(a, b)


@pytest.mark.parametrize("T,Hq,Hkv,pad", [(316, 8, 2, False), (320, 4, 4, False), (17, 4, 1, False), (129, 4, 2, True), (300, 2, 2, True), (64, 2, 1, True)])
def test_fused_attention_backward_matches_reference_and_the_kernel_pair(T, Hq, Hkv, pad):
    """attn_bwd_fused_k (head_dim 128, causal, T <= 320: one block per (batch, query head), S / dP once, dS through LDS) against
    the f32 reference and against the dQ + dK/dV kernel pair it replaces (option 13 = 0): same per-element arithmetic, another
    summation order - agreement to bf16 rounding of the outputs; left / right padding; zero gradient for padded keys;
    repeated launches bit-identical."""
    from ultravox_amd import _lib
    torch.manual_seed(23)
    B, D = 3, 128
    q = bf(torch.randn(B, T, Hq, D, device=DEV)); k = bf(torch.randn(B, T, Hkv, D, device=DEV)); v = bf(torch.randn(B, T, Hkv, D, device=DEV))
    do = bf(torch.randn(B, T, Hq * D, device=DEV))
    kv_start = kv_len = None
    if pad:
        kv_start = torch.tensor([0, min(23, T // 3), 0], device=DEV, dtype=torch.int32)
        kv_len = torch.tensor([T, T, T - min(41, T // 2)], device=DEV, dtype=torch.int32)
    o, lse = ops().attention(q, k, v, causal=True, kv_start=kv_start, kv_len=kv_len)
    fused = ops().attention_bwd(q, k, v, o, lse, do, causal=True, kv_start=kv_start, kv_len=kv_len)
    again = ops().attention_bwd(q, k, v, o, lse, do, causal=True, kv_start=kv_start, kv_len=kv_len)
    assert all(torch.equal(a, b) for a, b in zip(fused, again))
    _lib.lib().uvx_set_option(13, 0)
    try:
        pair = ops().attention_bwd(q, k, v, o, lse, do, causal=True, kv_start=kv_start, kv_len=kv_len)
    finally:
        _lib.lib().uvx_set_option(13, 1)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ref, ok = sdpa_ref(qr, kr, vr, True, 0, D ** -0.5, kv_start=kv_start, kv_len=kv_len)
    valid = ok.any(-1)[:, 0]
    ref.backward(do.float() * valid[:, :, None].expand(B, T, Hq * D))
    for name, got, old, want in zip(("dq", "dk", "dv"), fused, pair, (qr.grad, kr.grad, vr.grad)):
        if name == "dq":
            got, old = got.float() * valid[:, :, None, None], old.float() * valid[:, :, None, None]
        assert rel_l2(got, want) < 2e-2, name
        assert rel_l2(got, old) < 6e-3, name                     # two bf16 roundings of the same sums apart
        assert rel_l2(got, want) < 1.1 * rel_l2(old, want) + 1e-4, name
    if pad:
        s1, e2 = int(kv_start[1]), int(kv_len[2])
        assert fused[1][1, :s1].abs().max().item() == 0 and fused[2][1, :s1].abs().max().item() == 0
        assert fused[1][2, e2:].abs().max().item() == 0 and fused[2][2, e2:].abs().max().item() == 0


@pytest.mark.parametrize("rows,cols", [(8, 8192), (3, 4096), (316, 8192), (512, 4096), (5, 3584), (64, 1536), (7, 5120), (2, 7168)])
def test_rmsnorm_forward_with_the_row_in_registers_is_bit_identical(rows, cols):
    """Round 5: rows of at most 8192 columns keep their values in registers between the sum of squares and the scaling and request the weight
    vectors together with them (one round trip to memory instead of two: the decode step's few-row norms are pure latency) - the same per-thread
    and block summation order, arithmetic and rounding points as the two-pass kernel (option 18 = 1): bit-identical outputs, and the
    bf16-vs-torch check of the rounding-points test still holds for it."""
    from ultravox_amd import _lib, ops
    L = _lib.lib()
    torch.manual_seed(rows + cols)
    x = (torch.randn(rows, cols, device=DEV) * 1.7).bfloat16()
    w = (1.0 + 0.2 * torch.randn(cols, device=DEV)).bfloat16()
    try:
        L.uvx_set_option(18, 0)
        a = ops.rmsnorm(x, w, 1e-5)
        L.uvx_set_option(18, 1)
        b = ops.rmsnorm(x, w, 1e-5)
    finally:
        L.uvx_set_option(18, 0)
    assert torch.equal(a, b)
    xf = x.float()
    want = w.float() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).bfloat16().float()
    assert rel_l2(a, want) < 3e-3


@pytest.mark.parametrize("Hq,Hkv,T,causal,block", [(3, 3, 1500, False, 0), (2, 2, 333, False, 50), (4, 2, 130, True, 0), (2, 2, 97, False, 0)])
def test_head_dim_64_attention_tile_forms_are_bit_identical(Hq, Hkv, T, causal, block):
    """Round 6: the head_dim-64 backward pair with two 16-row tiles per wave (tuning option 19: 0 = the dQ kernel, 1 = the round 1-5
    one-tile form, 2..6 = both kernels / 64-row-step / 8-wave forms kept for A/B) and the forward row max through v_permlane swaps
    (option 20: 1 = the ds_bpermute shuffles).  Every output element sums the same terms in the same order in every form."""
    from ultravox_amd import _lib
    L = _lib.lib()
    torch.manual_seed(23)
    B, D = 2, 64
    q = bf(torch.randn(B, T, Hq, D, device=DEV))
    k = bf(torch.randn(B, T, Hkv, D, device=DEV))
    v = bf(torch.randn(B, T, Hkv, D, device=DEV))
    do = bf(torch.randn(B, T, Hq * D, device=DEV))
    kv_len = torch.tensor([T, max(1, T - 37)], device=DEV, dtype=torch.int32) if not causal else None
    try:
        L.uvx_set_option(20, 1)
        o0, lse0 = ops().attention(q, k, v, causal=causal, block=block, kv_len=kv_len)
        L.uvx_set_option(20, 0)
        o, lse = ops().attention(q, k, v, causal=causal, block=block, kv_len=kv_len)
        assert torch.equal(o, o0) and torch.equal(lse, lse0)
        L.uvx_set_option(19, 1)
        want = ops().attention_bwd(q, k, v, o, lse, do, causal=causal, block=block, kv_len=kv_len)
        for form in (0, 2, 3, 4, 5, 6):
            L.uvx_set_option(19, form)
            got = ops().attention_bwd(q, k, v, o, lse, do, causal=causal, block=block, kv_len=kv_len)
            for a, b, name in zip(got, want, ("dq", "dk", "dv")):
                assert torch.equal(a, b), (form, name)
    finally:
        L.uvx_set_option(19, 0)
        L.uvx_set_option(20, 0)


@pytest.mark.parametrize("M,N,K", [(1504, 2048, 1024), (12000, 4096, 1024), (300, 1024, 4096), (100, 132, 192), (37, 8, 64), (515, 1000, 128)])
def test_gemm_gelu_epilogues_that_keep_and_consume_the_pre_activation(M, N, K):
    """Round 6 (the Whisper tower under LoRA training): act 2 = fc1 whose epilogue writes the pre-activation AND gelu of it, act 3 = the
    fc2 dgrad whose epilogue multiplies by gelu'(pre) - against uvx_gemm + uvx_gelu / uvx_gelu_bwd, bit for bit, on aligned shapes (the
    LDS-staged whole-line epilogue), ragged ones (the fragment-layout fallback) and a tail-split launch; and against torch within bf16."""
    import ctypes as C
    from ultravox_amd import _lib
    torch.manual_seed(31)
    a, w = bf(torch.randn(M, K, device=DEV)), bf(torch.randn(N, K, device=DEV) * K ** -0.5)
    bias = bf(torch.randn(N, device=DEV))
    L = _lib.lib()
    pre_ref = ops().gemm(a, w, bias=bias)
    act_ref = torch.empty_like(pre_ref)
    _lib.check(L.uvx_gelu(None, _lib.BF16, C.c_void_p(pre_ref.data_ptr()), C.c_void_p(act_ref.data_ptr()), C.c_longlong(pre_ref.numel())), "uvx_gelu")
    act = torch.empty_like(pre_ref)
    pre = ops().gemm(a, w, bias=bias, act="gelu_keep", c2=act)
    assert torch.equal(pre, pre_ref) and torch.equal(act, act_ref)
    want = torch.nn.functional.gelu(pre_ref.float())
    assert rel_l2(act, want) < 4e-3
    # backward: d = a . w^T (no bias), times gelu'(x) with x = any saved pre-activation of that shape
    x = bf(torch.randn(M, N, device=DEV) * 1.5)
    d_ref = ops().gemm(a, w)
    g_ref = torch.empty_like(d_ref)
    _lib.check(L.uvx_gelu_bwd(None, _lib.BF16, C.c_void_p(d_ref.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(g_ref.data_ptr()),
                              C.c_longlong(d_ref.numel())), "uvx_gelu_bwd")
    g = ops().gemm(a, w, act="gelu_bwd", c2=x)
    assert torch.equal(g, g_ref)
    xf = x.float().requires_grad_(True)
    torch.nn.functional.gelu(xf).backward(d_ref.float())
    assert rel_l2(g, xf.grad) < 4e-3


@pytest.mark.parametrize("variant,twin", [(61, 31), (62, 34)])
def test_mfma_32x32x16_merged_phase_kernels(variant, twin):
    """Round 6: the merged-phase GEMM on v_mfma_f32_32x32x16_bf16 (61 = 256 x 256, 62 = 128 x 256; LDS image swizzled with (row >> 1) & 7,
    its own whole-line epilogue for bias / GELU / residual) against the f32 reference at the production bar and against its 16 x 16 x 32
    twin to summation order (16-column vs 32-column k steps); epilogues it does not carry (f32 output, ragged N) run the twin bit for bit;
    20 repeats bit-identical (race screen on the new swizzle / fragment reads)."""
    from ultravox_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device=DEV).manual_seed(variant)

    def run(v, fn):
        L.uvx_gemm_force_variant(v)
        try:
            return fn()
        finally:
            L.uvx_gemm_force_variant(-1)

    for (M, N, K) in [(256, 256, 64), (300, 520, 128), (128, 256, 192), (1000, 1032, 256), (2528, 6144, 4096), (12000, 1024, 1024), (316, 768, 8192)]:
        a = (torch.randn(M, K, device=DEV, generator=g) * 0.5).bfloat16()
        b = (torch.randn(N, K, device=DEV, generator=g) * 0.5).bfloat16()
        bias = torch.randn(N, device=DEV, generator=g).bfloat16()
        resid = torch.randn(M, N, device=DEV, generator=g).bfloat16()
        ref = a.float() @ b.float().t()
        modes = {"plain": (lambda: ops().gemm(a, b), ref), "bias+res": (lambda: ops().gemm(a, b, bias=bias, residual=resid), ref + bias.float() + resid.float()),
                 "bias+gelu": (lambda: ops().gemm(a, b, bias=bias, act="gelu"), F.gelu(ref + bias.float()))}
        for name, (fn, want) in modes.items():
            got, tw = run(variant, fn), run(twin, fn)
            assert rel_l2(got, want) < 5e-3, (variant, (M, N, K), name, rel_l2(got, want))
            assert rel_l2(got, tw) < 2e-3 and (got == tw).float().mean().item() > 0.9, (variant, (M, N, K), name)
        first = run(variant, modes["plain"][0])
        for _ in range(20):
            assert torch.equal(run(variant, modes["plain"][0]), first), (variant, (M, N, K), "repeat")
        assert torch.equal(run(variant, lambda: ops().gemm(a, b, out_f32=True)), run(twin, lambda: ops().gemm(a, b, out_f32=True)))
    a = (torch.randn(200, 128, device=DEV, generator=g)).bfloat16()
    b = (torch.randn(132, 128, device=DEV, generator=g)).bfloat16()          # N % 8 != 0: the twin's fragment-layout epilogue
    assert torch.equal(run(variant, lambda: ops().gemm(a, b)), run(twin, lambda: ops().gemm(a, b)))


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 520, 128), (2528, 4096, 6144), (2528, 4096, 4096), (1000, 1032, 256), (12000, 1024, 3072), (2528, 4096, 28672),
                                   (316, 8192, 1024), (128, 264, 192)])
def test_gemm_nn_form_equals_the_nt_kernel_on_the_transposed_matrix(M, N, K):
    """Round 6: C = A . B with B stored [K, N] (uvx_gemm_desc_t.b_kn; the dgrad d x = d y . W on the forward weight as it lies): the W tile is
    staged [k][n] and read through ds_read_b64_tr_b16 in natural k order, so every MFMA sees the operands of the NT kernel on B^T - bit-identical
    output on every merged-phase tile (31..34), ragged M / N (N % 8 == 0), tail-split launches, with a residual; 10 repeats bit-identical."""
    from ultravox_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = (torch.randn(M, K, device=DEV, generator=g) * 0.5).bfloat16()
    w = (torch.randn(K, N, device=DEV, generator=g) * K ** -0.5).bfloat16()          # [K, N]: the forward weight of a dgrad
    wt = w.t().contiguous()                                                           # [N, K]: the copy the NT kernel reads
    resid = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    try:
        for v in (31, 32, 33, 34, -1):
            L.uvx_gemm_force_variant(v)
            want = ops().gemm(a, wt)
            got = ops().gemm(a, w, b_kn=True)
            assert torch.equal(got, want), (v, int((got != want).sum()))
            assert torch.equal(ops().gemm(a, w, b_kn=True, residual=resid), ops().gemm(a, wt, residual=resid)), v
        L.uvx_gemm_force_variant(-1)
        first = ops().gemm(a, w, b_kn=True)
        for _ in range(10):
            assert torch.equal(ops().gemm(a, w, b_kn=True), first)
        assert rel_l2(first, a.float() @ w.float()) < 5e-3
    finally:
        L.uvx_gemm_force_variant(-1)


@pytest.mark.parametrize("B,Hq,Hkv,T,D", [(8, 32, 8, 316, 128), (3, 8, 2, 77, 128), (2, 16, 2, 512, 128), (1, 64, 8, 316, 128), (8, 32, 8, 316, 64), (2, 8, 2, 90, 64)])
def test_attention_forward_grouped_query_block_form_is_bit_identical(B, Hq, Hkv, T, D):
    """attn_fwd_k<128, .., GQ> (tuning option 25): the four waves of a block take four query heads of ONE KV head - per (head, query row) the key tiles arrive
    in the same order as in the default form, so o and the log-sum-exp agree bit for bit, with padding on either side, in every tile count."""
    from ultravox_amd import _lib, ops
    L = _lib.lib()
    torch.manual_seed(B * 1000 + T)
    qkv = torch.randn(B, T, (Hq + 2 * Hkv) * D, device=DEV).bfloat16()
    q = qkv[..., :Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hkv) * D].view(B, T, Hkv, D)
    v = qkv[..., (Hq + Hkv) * D:].view(B, T, Hkv, D)
    kv_start = torch.zeros(B, dtype=torch.int32, device=DEV)
    kv_len = torch.full((B,), T, dtype=torch.int32, device=DEV)
    kv_start[0], kv_len[B - 1] = 5, T - 9
    try:
        L.uvx_set_option(25, 5)
        o_ref, lse_ref = ops.attention(q, k, v, causal=True, kv_start=kv_start, kv_len=kv_len)
        for form in (0, 1, 3, 4):
            L.uvx_set_option(25, form)
            o, lse = ops.attention(q, k, v, causal=True, kv_start=kv_start, kv_len=kv_len)
            assert torch.equal(o, o_ref) and torch.equal(lse, lse_ref), form
    finally:
        L.uvx_set_option(25, 0)
