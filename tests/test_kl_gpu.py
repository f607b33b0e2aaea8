"""KL-distillation loss on the GPU through the C ABI: the kernel against the REFERENCE fixtures
(tests/golden/kl_loss.npz, generated from ultravox_model.py:157-256), and a whole KL train step against the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("case", ["basic", "no_eot", "temp1_w05", "one_empty_row", "padded_tail"])
def test_kl_kernel_matches_reference_fixture(golden_dir, case):
    from ultravox_amd import ops
    from ultravox_amd.model import kl_row_pairs
    z = np.load(os.path.join(golden_dir, "kl_loss.npz"))
    g = lambda k: torch.from_numpy(z[f"{case}.{k}"])
    pair_row, pair_w, _ = kl_row_pairs(g("labels"), g("alt_labels"), float(g("eot_loss_weight")))
    V = g("student").shape[-1]
    s, t = g("student").reshape(-1, V).to(DEV), g("teacher").reshape(-1, V).to(DEV)
    loss, dl = ops.kl_loss(s, t, pair_row.to(DEV), pair_w.to(DEV), float(g("temperature")))
    assert abs(loss.item() - float(g("loss"))) <= 2e-6 * max(1.0, abs(float(g("loss"))))
    assert (dl.cpu() - g("dstudent").reshape(-1, V)).abs().max().item() < 2e-7
    # grad_scale (gradient accumulation) scales the gradient only
    _, dl2 = ops.kl_loss(s, t, pair_row.to(DEV), pair_w.to(DEV), float(g("temperature")), grad_scale=0.25)
    assert (dl2.cpu() - 0.25 * g("dstudent").reshape(-1, V)).abs().max().item() < 1e-7


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_kl_kernel_two_distinct_partners_large_vocab(dtype):
    """A student row whose prediction-mask partner and end-of-turn partner are DIFFERENT teacher rows (label counts
    per sequence differ between the two tokenisations), V = 128256 / 8 wide rows, bf16 storage."""
    from ultravox_amd import ops
    g = torch.Generator().manual_seed(3)
    R, Rt, V, tau = 12, 9, 16032, 2.0
    s = (3 * torch.randn(R, V, generator=g)).to(dtype)
    t = (3 * torch.randn(Rt, V, generator=g)).to(dtype)
    pair_row = torch.full((2, R), -1, dtype=torch.int32)
    pair_w = torch.zeros(2, R)
    pair_row[0, [1, 2, 3, 7, 8]] = torch.tensor([0, 1, 2, 5, 6], dtype=torch.int32); pair_w[0, [1, 2, 3, 7, 8]] = 1 / 5
    pair_row[1, [3, 8, 10]] = torch.tensor([4, 6, 8], dtype=torch.int32); pair_w[1, [3, 8, 10]] = 0.7 / 3
    sr = s.float().clone().requires_grad_(True)
    ref = sum(pair_w[k, r] * F.kl_div(F.log_softmax(sr[r] / tau, -1), F.softmax(t[pair_row[k, r]].float() / tau, -1), reduction="sum")
              for k in range(2) for r in range(R) if pair_row[k, r] >= 0)
    ref.backward()
    loss, dl = ops.kl_loss(s.to(DEV), t.to(DEV), pair_row.to(DEV), pair_w.to(DEV), tau)
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item()) + 1e-7
    if dtype == torch.float32:
        assert (dl.cpu() - sr.grad).abs().max().item() < 1e-8 + 1e-5 * sr.grad.abs().max().item()
    else:
        assert rel_l2(dl, sr.grad) < 4e-3                       # bf16 rounding of the stored gradient
    assert dl[0].abs().max().item() == 0.0                      # rows without a partner: exact zeros


def _alt_fields(b, cfg, audio_start, n_sup, n_transcript=6, seed=99):
    """Text-only twin of a synthetic batch (ultravox_data_proc.py:112-131): the audio placeholder run is replaced by
    transcript tokens, the supervised tail is the same tokens."""
    ids = b["input_ids"]
    B = ids.shape[0]
    Na = int(b["audio_token_len"][0])
    g = torch.Generator().manual_seed(seed)
    tr = torch.randint(0, cfg.text_config.vocab_size - 1, (B, n_transcript), generator=g)
    alt = torch.cat([ids[:, :audio_start], tr, ids[:, audio_start + Na:]], 1)
    alt_labels = alt.clone()
    alt_labels[:, : alt.shape[1] - n_sup] = -100
    return {"alt_input_ids": alt, "alt_attention_mask": torch.ones_like(alt), "alt_labels": alt_labels}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_kl_train_step_matches_oracle(dtype):
    from oracle.reference_cpu import OracleModel, synthetic_batch
    from test_model_gpu import SMALL
    from ultravox_amd.config import LossConfig, LossFunction, UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = UltravoxConfig(**SMALL)
    sd = random_state_dict(cfg, seed=21, dtype=dtype)
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=dtype)
    lc = LossConfig(loss_function=LossFunction.KL_Divergence, kl_temperature=2.0, eot_loss_weight=1.0)
    model.set_loss_config(lc)
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    b = synthetic_batch(cfg, 2, 2.0, n_text=24, audio_start=5, n_supervised=8)
    pcm = b.pop("pcm")
    b.update(_alt_fields(b, cfg, 5, 8))
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV))
    gb = {k: v.to(DEV) for k, v in b.items()}
    ob = {**b, "audio_values": mel.cpu().float(), "kl": {"temperature": 2.0, "eot_loss_weight": 1.0}}
    ref, grads, _ = oracle.train_step(ob)
    # eval mode keeps the CE loss (ultravox_model.py:335: the KL branch is `if self.training`)
    model.eval()
    ce = model.forward(audio_values=mel.to(dtype), **gb).loss
    model.train()
    loss = model.forward_backward(audio_values=mel.to(dtype), **gb)
    assert abs(ce.item() - loss.item()) > 1e-3
    mine = {k: v.clone() for k, v in model.projector_grads().items()}
    if dtype == torch.float32:
        assert abs(loss.item() - ref["loss"].item()) < 1e-5 + 1e-4 * abs(ref["loss"].item())
        for k, g in grads.items():
            assert rel_l2(mine[k], g) < 1e-3, k
    else:
        assert abs(loss.item() - ref["loss"].item()) < 0.05 * abs(ref["loss"].item()) + 1e-4
        for k, g in grads.items():
            assert rel_l2(mine[k], g) < 0.1, k
    # gradient accumulation: grad_scale scales the gradient linearly
    model.forward_backward(grad_scale=0.5, audio_values=mel.to(dtype), **gb)
    half = model.projector_grads()
    for k in mine:
        assert rel_l2(half[k] * 2, mine[k]) < (1e-5 if dtype == torch.float32 else 2e-2), k


def test_kl_requires_alt_fields():
    from test_model_gpu import SMALL
    from ultravox_amd.config import LossConfig, LossFunction, UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    cfg = UltravoxConfig(**SMALL)
    model = UltravoxModel(cfg, device=DEV, dtype=torch.bfloat16, seed=1)
    model.set_loss_config(LossConfig(loss_function=LossFunction.KL_Divergence))
    model.train()
    ids = torch.randint(0, 100, (1, 8), device=DEV)
    with pytest.raises(ValueError):
        model.forward(input_ids=ids, labels=ids)


@pytest.mark.parametrize("tag", ["t2_eot1", "t1_eot0"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_kl_step_matches_the_reference_forward_fixture(tag, dtype):
    """The KL-distillation step of the HIP path against the REFERENCE forward in training mode itself (fixture
    kl_forward_reference.npz: imported reference on the seeded tiny model, audio tower stubbed): loss and projector
    gradients, with and without the end-of-turn term.  bf16 runs the rows-only heads (uvx_llm_fwd_rows / kl_loss_rows)."""
    import json
    import os
    import numpy as np
    import forward_fixture_util as U
    from test_oracle_pinning import load_forward_fixture
    from ultravox_amd.config import LossConfig, LossFunction
    from ultravox_amd.model import UltravoxModel
    here = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(here, "kl_forward_reference.npz"))
    meta = json.load(open(os.path.join(here, "kl_forward_reference.json")))[tag]
    cfg, sd, batch, enc, _ = load_forward_fixture("ln_mid")
    model = UltravoxModel(cfg, state_dict={k: v.to(dtype) for k, v in sd.items()}, device=DEV, dtype=dtype)
    model.set_loss_config(LossConfig(loss_function=LossFunction.KL_Divergence, kl_temperature=meta["kl_temperature"],
                                     eot_loss_weight=meta["eot_loss_weight"]))
    tower = enc.to(DEV, dtype)
    model.audio_tower_forward = lambda audio_values, audio_len: tower[: audio_values.shape[0]]
    gb = {k: v.to(DEV) for k, v in {**batch, **U.alt_batch()}.items()}
    model.train()
    loss = model.forward_backward(audio_values=torch.zeros(len(enc), 80, 3000, device=DEV, dtype=dtype), **gb)
    want = float(z[f"{tag}.loss"])
    assert abs(loss.item() - want) < (1e-5 + 1e-4 * want if dtype == torch.float32 else 0.05 * want + 1e-4)
    mine = model.projector_grads()
    for k in z.files:
        if k.startswith(tag + ".g."):
            assert rel_l2(mine[k[len(tag) + 3:]], torch.from_numpy(z[k])) < (1e-3 if dtype == torch.float32 else 0.1), k


@pytest.mark.parametrize("side_stream", [True, False])
def test_kl_rows_step_with_the_compact_last_layer_matches_the_full_row_one(side_stream):
    """Round 6: uvx_llm_fwd_rows / uvx_llm_bwd_rows run the LAST layer's row-wise half (o_proj, MLP, final norm and their gradients) on
    the listed rows only - student (stash kept for the backward) and teacher (no stash: the other layer slot is the gather scratch) -
    as the CE pair does on the supervised rows.  Against the same step with that turned off (uvx_set_option(3, 0): every row through the
    last layer): the same per-row arithmetic, so the loss agrees to the f32 row-sum grouping and the projector gradients to the bf16
    round-off of one scatter (the CE pair's bar, tests/test_model_gpu.py)."""
    from oracle.reference_cpu import synthetic_batch
    from test_model_gpu import SMALL
    from ultravox_amd import _lib
    from ultravox_amd.config import LossConfig, LossFunction, UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = UltravoxConfig(**dict(SMALL, text_config=dict(SMALL["text_config"], num_hidden_layers=3)))
    sd = random_state_dict(cfg, seed=31, dtype=torch.bfloat16)
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16)
    model.set_loss_config(LossConfig(loss_function=LossFunction.KL_Divergence, kl_temperature=2.0, eot_loss_weight=1.0))
    model.kl_teacher_side_stream = side_stream
    b = synthetic_batch(cfg, 3, 2.0, n_text=24, audio_start=5, n_supervised=8)
    pcm = b.pop("pcm")
    b.update(_alt_fields(b, cfg, 5, 8))
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV)).bfloat16()
    gb = {k: v.to(DEV) for k, v in b.items()}
    model.train()
    L = _lib.lib()
    runs = []
    try:
        for opt in (1, 0, 1):
            L.uvx_set_option(3, opt)
            loss = model.forward_backward(audio_values=mel, **gb)
            assert model._llm_top_rows == bool(opt)
            runs.append((loss.clone(), model.proj_grad.clone()))
    finally:
        L.uvx_set_option(3, 1)
    torch.cuda.synchronize()
    (l1, g1), (l0, g0), (l2, g2) = runs
    assert torch.equal(l1, l2) and torch.equal(g1, g2)                     # the compact step is reproducible bit for bit
    assert abs(l1.item() - l0.item()) < 1e-6 * abs(l0.item()) + 1e-9
    assert rel_l2(g1, g0) < 5e-3 and g1.float().abs().max().item() > 0
