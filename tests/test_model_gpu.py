"""Hot-path parity on a real MI355X through the C ABI: encoder, projector (fwd + bwd), merge, LLM
(fwd + activation-gradient bwd), and one whole adapter-training step, each against the CPU oracle
(oracle/reference_cpu.py, pinned in tests/test_oracle_pinning.py) on the same seeded inputs/weights.

Tolerances.  The production path stores activations in bf16 (what the reference does on GPU,
config_base.py:245-249 forces f32 only off-GPU) while the oracle runs in f32 on the SAME bf16-rounded
weights, so each comparison states a relative-L2 bound of a few bf16 ulps (2^-8 = 3.9e-3) accumulated
over the layers involved; integer/index outputs are bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

SMALL = dict(
    audio_config=dict(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256, num_mel_bins=80,
                      max_source_positions=1500),
    text_config=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                     num_key_value_heads=2, vocab_size=512, rope_theta=10000.0, max_position_embeddings=512,
                     eos_token_id=2),
    hidden_size=256, stack_factor=8, projector_ln_mid=True)


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def build(seed=0, **kw):
    from oracle.reference_cpu import OracleModel
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = UltravoxConfig(**{**SMALL, **kw})
    sd = {k: v.bfloat16() for k, v in random_state_dict(cfg, seed=seed).items()}  # both sides see bf16-rounded weights
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16)
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    return cfg, sd, model, oracle


def batch_for(cfg, B=2, seconds=2.0, n_text=24, audio_start=5, n_sup=8):
    from oracle.reference_cpu import synthetic_batch, logmel_ref
    b = synthetic_batch(cfg, B, seconds, n_text=n_text, audio_start=audio_start, n_supervised=n_sup)
    b["audio_values"] = logmel_ref(b.pop("pcm"), cfg.audio_config.num_mel_bins)
    return b


def test_encoder_matches_oracle():
    from oracle.reference_cpu import whisper_encoder_ref
    cfg, sd, model, oracle = build(1)
    torch.manual_seed(0)
    mel = torch.randn(3, 80, 300)
    lens = torch.tensor([300, 201, 64])
    got = model.audio_tower_forward(mel.to(DEV), lens.to(DEV))
    want = whisper_encoder_ref(oracle.sd, cfg, mel.bfloat16().float(), lens)
    assert got.shape == want.shape == (3, 150, 128)
    assert rel_l2(got, want) < 2e-2
    # odd frame count (Te = (F-1)//2 + 1) and no padding mask
    mel2 = torch.randn(1, 80, 77)
    got2 = model.audio_tower_forward(mel2.to(DEV), None)
    want2 = whisper_encoder_ref(oracle.sd, cfg, mel2.bfloat16().float(), None)
    assert got2.shape == want2.shape == (1, 39, 128) and rel_l2(got2, want2) < 2e-2
    with pytest.raises(ValueError, match="of length 3000 or less, but found 3002"):
        model.audio_tower_forward(torch.zeros(1, 80, 3002, device=DEV), None)


def test_encoder_latency_block_mask():
    from oracle.reference_cpu import whisper_encoder_ref
    cfg, sd, model, oracle = build(2, audio_latency_block_size=50)
    mel = torch.randn(2, 80, 400)
    lens = torch.tensor([400, 250])
    got = model.audio_tower_forward(mel.to(DEV), lens.to(DEV))
    want = whisper_encoder_ref(oracle.sd, cfg, mel.bfloat16().float(), lens)
    assert rel_l2(got, want) < 2e-2
    with pytest.raises(AssertionError, match="must divide 3000 evenly"):
        build(2, audio_latency_block_size=13)


@pytest.mark.parametrize("variant", ["mid", "post"])
def test_projector_against_reference_fixture(golden_dir, variant):
    """fixture = outputs + gradients of the REFERENCE UltravoxProjector (tests/golden/make_golden.py)."""
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    z = np.load(os.path.join(golden_dir, f"projector_ln_{variant}.npz"))
    cfg = UltravoxConfig(audio_config=dict(d_model=32, encoder_layers=1, encoder_attention_heads=1, encoder_ffn_dim=64),
                         text_config=dict(hidden_size=64, intermediate_size=64, num_hidden_layers=1, num_attention_heads=1,
                                          num_key_value_heads=1, vocab_size=128), hidden_size=256,
                         projector_ln_mid=(variant == "mid"))
    sd = random_state_dict(cfg, seed=0)
    for k in z.files:
        if k.startswith("w."):
            sd["multi_modal_projector." + k[2:]] = torch.from_numpy(z[k])
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16)
    x = torch.from_numpy(z["x"]).to(DEV).bfloat16()
    y = model.multi_modal_projector_forward(x)
    assert tuple(y.shape) == z["y"].shape == (3, 3, 64)
    assert rel_l2(y, torch.from_numpy(z["y"])) < 1.5e-2
    model._projector_backward(torch.from_numpy(z["gy"]).to(DEV).bfloat16())
    grads = model.projector_grads()
    for k in z.files:
        if k.startswith("g."):
            assert rel_l2(grads["multi_modal_projector." + k[2:]], torch.from_numpy(z[k])) < 3e-2, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["gelu", "silu", "relu", "gelu_pytorch_tanh"])
def test_projector_with_a_plain_activation_against_reference_fixture(golden_dir, act, dtype):
    """projector_act != "swiglu" (round 5; ultravox_model.py:754-755: ACT2FN[projector_act], the width is kept): forward and every gradient
    of uvx_projector_fwd / _bwd against the REFERENCE UltravoxProjector's (fixture projector_act.npz), f32 mode tight, bf16 at the stage bars;
    and a whole training step with such a projector against the oracle."""
    from oracle.reference_cpu import OracleModel, synthetic_batch
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    z = np.load(os.path.join(golden_dir, "projector_act.npz"))
    cfg = UltravoxConfig(audio_config=dict(d_model=32, encoder_layers=1, encoder_attention_heads=1, encoder_ffn_dim=64),
                         text_config=dict(hidden_size=64, intermediate_size=64, num_hidden_layers=1, num_attention_heads=1,
                                          num_key_value_heads=1, vocab_size=128), hidden_size=128,
                         projector_ln_mid=bool(z[f"{act}.ln_mid"]), projector_act=act)
    sd = random_state_dict(cfg, seed=0)
    assert tuple(sd["multi_modal_projector.linear_2.weight"].shape) == (64, 128)
    for k in z.files:
        if k.startswith(f"{act}.w."):
            sd["multi_modal_projector." + k[len(act) + 3:]] = torch.from_numpy(z[k])
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=dtype)
    x = torch.from_numpy(z[f"{act}.x"]).to(DEV).to(dtype)
    y = model.multi_modal_projector_forward(x)
    f32 = dtype == torch.float32
    assert tuple(y.shape) == z[f"{act}.y"].shape == (3, 3, 64)
    assert rel_l2(y, torch.from_numpy(z[f"{act}.y"])) < (1e-5 if f32 else 1.5e-2)
    model._projector_backward(torch.from_numpy(z[f"{act}.gy"]).to(DEV).to(dtype))
    grads = model.projector_grads()
    n = 0
    for k in z.files:
        if k.startswith(f"{act}.g."):
            assert rel_l2(grads["multi_modal_projector." + k[len(act) + 3:]], torch.from_numpy(z[k])) < (2e-5 if f32 else 3e-2), k
            n += 1
    assert n == 4
    # the whole step: encoder -> this projector -> LLM -> loss -> projector gradients, against the oracle's autograd
    cfg2 = UltravoxConfig(**{**SMALL, "projector_act": act, "projector_ln_mid": act in ("gelu", "relu")})
    sd2 = {k: v.to(dtype) for k, v in random_state_dict(cfg2, seed=31).items()}
    m2, oracle = UltravoxModel(cfg2, state_dict=sd2, device=DEV, dtype=dtype), OracleModel(cfg2, sd2, dtype=torch.float32)
    b = synthetic_batch(cfg2, 2, 3.0, n_text=24, audio_start=5, n_supervised=8)
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg2.audio_config.num_mel_bins).logmel_device(pcm.to(DEV)).to(dtype)
    ref, g_ref, _ = oracle.train_step({**b, "audio_values": mel.cpu().float()})
    m2.train()
    loss = m2.forward_backward(audio_values=mel, **{k: v.to(DEV) for k, v in b.items()})
    assert abs(loss.item() - ref["loss"].item()) < (1e-4 if f32 else 2e-2) * abs(ref["loss"].item())
    mine = m2.projector_grads()
    assert set(mine) == set(g_ref)
    for k, g in g_ref.items():
        assert rel_l2(mine[k], g) < (2e-3 if f32 else 8e-2), k


def test_forward_logits_loss_match_oracle():
    cfg, sd, model, oracle = build(3)
    b = batch_for(cfg)
    out = model.forward(**{k: v.to(DEV) for k, v in b.items()})
    with torch.no_grad():
        ref = oracle.forward(**{**b, "audio_values": b["audio_values"].bfloat16().float()})
    assert tuple(out.logits.shape) == tuple(ref["logits"].shape)
    assert rel_l2(out.logits, ref["logits"]) < 3e-2
    assert abs(out.loss.item() - ref["loss"].item()) < 2e-2 * abs(ref["loss"].item())
    # audio rows land exactly where the reference puts them: rows outside [start, start+len) are the
    # (bit-exact) token embeddings
    emb = torch.nn.functional.embedding(b["input_ids"], sd["language_model.model.embed_tokens.weight"])
    merged = model._embed_merge(None, b["input_ids"], model.multi_modal_projector_forward(
        model.audio_tower_forward(b["audio_values"].to(DEV), b["audio_lens"].to(DEV))), b["audio_token_start_idx"],
        b["audio_token_len"], b["audio_batch_size"], *b["input_ids"].shape).cpu()
    s, n = int(b["audio_token_start_idx"][0]), int(b["audio_token_len"][0])
    assert torch.equal(merged[:, :s], emb[:, :s]) and torch.equal(merged[:, s + n:], emb[:, s + n:])
    assert rel_l2(merged[:, s:s + n], ref["audio_embeds"][:, :n]) < 2e-2


def test_text_only_and_padding_masks():
    cfg, sd, model, oracle = build(4)
    torch.manual_seed(1)
    B, T = 3, 40
    ids = torch.randint(0, 512, (B, T))
    labels = ids.clone()
    labels[:, :10] = -100
    am = torch.ones(B, T, dtype=torch.long)
    am[1, 30:] = 0      # right padding
    am[2, :7] = 0       # left padding
    labels[am == 0] = -100
    out = model.forward(input_ids=ids.to(DEV), labels=labels.to(DEV), attention_mask=am.to(DEV))
    with torch.no_grad():
        ref = oracle.forward(input_ids=ids, labels=labels, attention_mask=am)
    keep = am.bool()
    assert rel_l2(out.logits.cpu()[keep], ref["logits"][keep]) < 3e-2
    assert abs(out.loss.item() - ref["loss"].item()) < 2e-2 * abs(ref["loss"].item())


def test_forward_argument_checks_mirror_reference():
    cfg, sd, model, oracle = build(5)
    b = {k: v.to(DEV) for k, v in batch_for(cfg).items()}
    bad = dict(b); bad["audio_token_len"] = b["audio_token_len"][:1]
    with pytest.raises(AssertionError, match="must have the same batch size"):
        model.forward(**bad)
    bad = dict(b); bad.pop("audio_lens")
    with pytest.raises(AssertionError, match="must be provided"):
        model.forward(**bad)
    bad = dict(b); bad["audio_batch_size"] = b["audio_batch_size"][:1]
    with pytest.raises(AssertionError, match="audio_batch_size and inputs_embeds must have the same batch size"):
        model.forward(**bad)


def test_train_step_matches_oracle():
    """loss, projector gradients, clipped AdamW update — one HF-Trainer optimizer step."""
    from ultravox_amd.model import UltravoxTrainer
    cfg, sd, model, oracle = build(6)
    b = batch_for(cfg, B=3, seconds=3.0, n_text=32, n_sup=12)
    trainer = UltravoxTrainer(model, lr=2e-3, master_weights=True)
    params = [oracle.sd[k] for k in oracle.trainable]
    opt = torch.optim.AdamW(params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    ob = {**b, "audio_values": b["audio_values"].bfloat16().float()}
    for step in range(2):
        out, grads, gn = oracle.train_step(ob, opt)
        loss = trainer.train_step(**{k: v.to(DEV) for k, v in b.items()})
        assert abs(loss.item() - out["loss"].item()) < 2e-2 * abs(out["loss"].item()), step
        mine = model.projector_grads()
        for k in oracle.trainable:
            assert rel_l2(mine[k], grads[k]) < 6e-2, (step, k)
        assert abs(trainer.grad_norm().item() - gn.item()) < 5e-2 * gn.item()
        new = model.projector_state_dict()
        for k in oracle.trainable:
            # Adam's first steps move every weight by ~lr regardless of gradient scale: compare the UPDATE
            delta_ref = oracle.sd[k].detach() - sd[k].float()
            delta = new[k].float().cpu() - sd[k].float()
            assert rel_l2(delta, delta_ref) < 0.25, (step, k)


def test_llm_input_gradient_matches_oracle():
    from oracle.reference_cpu import llama_ref, causal_lm_loss_ref
    cfg, sd, model, oracle = build(7)
    torch.manual_seed(2)
    B, T, D = 2, 48, 256
    emb = (torch.randn(B, T, D) * 0.5).bfloat16()
    labels = torch.randint(0, 512, (B, T)); labels[:, :30] = -100
    out = model.language_model_forward(emb.to(DEV), labels=labels.to(DEV), want_logits=False, save_for_bwd=True)
    import ctypes as C
    from ultravox_amd import _lib
    d = model.language_model_backward(1.0)        # uvx_llm_bwd_train: pairs with the forward above
    e = emb.float().requires_grad_(True)
    loss = causal_lm_loss_ref(llama_ref(oracle.sd, cfg, e, None), labels)
    loss.backward()
    assert abs(out.loss.item() - loss.item()) < 2e-2 * loss.item()
    assert rel_l2(d, e.grad) < 6e-2


def test_processor_collator_model_end_to_end_ragged_batch():
    """Whole reference call chain on ragged input: UltravoxProcessor (log-mel ON DEVICE, chunking of a 35 s clip
    into 2 encoder items, placeholder expansion) -> DataCollatorForSeq2SeqWithAudio (right padding, a text-only
    sample with audio_batch_size 0, audio right-padded to the longest item) -> UltravoxModel.forward, against
    the oracle fed with the same batch."""
    from fake_tokenizer import FakeTokenizer
    from oracle.reference_cpu import FeatureExtractorRef
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.processing import DataCollatorForSeq2SeqWithAudio, UltravoxProcessor
    cfg, sd, model, oracle = build(8)
    tok = FakeTokenizer()
    proc = UltravoxProcessor(WhisperFeatureExtractor(80, device=DEV), tokenizer=tok)
    proc_ref = UltravoxProcessor(FeatureExtractorRef(80), tokenizer=FakeTokenizer())
    rng = np.random.RandomState(5)
    clips = [rng.randn(16000 * 35).astype(np.float32) * 0.1, rng.randn(16000 * 3 + 77).astype(np.float32) * 0.1]
    samples = [("Describe <|audio|> please now", [clips[0]]), ("no audio in this one at all", []),
               ("two words <|audio|>", [clips[1]])]
    feats, feats_ref = [], []
    for text, aud in samples:
        for P, out in ((proc, feats), (proc_ref, feats_ref)):
            kw = dict(audios=aud, sampling_rate=16000) if aud else {}
            r = P(text, **kw)
            f = {k: (v[0] if k in ("input_ids", "attention_mask") else v) for k, v in r.items()}
            ids = f["input_ids"] % 512          # fold the fake token ids into the tiny vocabulary
            f["input_ids"] = ids
            f["labels"] = ids.clone()
            f["labels"][: len(ids) // 2] = -100
            if "audio_batch_size" not in f:
                f["audio_batch_size"] = torch.tensor([0])
            out.append(f)
    batch = DataCollatorForSeq2SeqWithAudio(tok)(feats)
    ref_batch = DataCollatorForSeq2SeqWithAudio(tok)(feats_ref)
    for bt in (batch, ref_batch):   # the collator pads with the (huge) EOS id: fold it into the tiny vocabulary too
        bt["input_ids"] = bt["input_ids"] % 512
    # integer contract identical whether the mel came from the device or from the oracle
    for k in ("input_ids", "attention_mask", "labels", "audio_token_start_idx", "audio_lens", "audio_token_len", "audio_batch_size"):
        assert torch.equal(batch[k].cpu(), ref_batch[k].cpu()), k
    assert batch["audio_batch_size"].reshape(-1).tolist() == [2, 0, 1] and batch["audio_values"].shape[0] == 3
    assert (batch["audio_values"].cpu() - ref_batch["audio_values"]).abs().max().item() < 5e-4
    gb = {k: v.to(DEV) for k, v in batch.items()}
    gb["audio_batch_size"] = gb["audio_batch_size"].reshape(-1)
    out = model.forward(**gb)
    ob = {k: v.cpu() for k, v in batch.items()}
    ob["audio_batch_size"] = ob["audio_batch_size"].reshape(-1)
    ob["audio_values"] = ob["audio_values"].bfloat16().float()
    with torch.no_grad():
        ref = oracle.forward(**ob)
    keep = ob["attention_mask"].bool()
    assert rel_l2(out.logits.cpu()[keep], ref["logits"][keep]) < 3e-2
    assert abs(out.loss.item() - ref["loss"].item()) < 2e-2 * abs(ref["loss"].item())


def test_overlapping_audio_ranges_follow_reference_loop_order():
    """Two audio items whose placeholder ranges overlap: the reference's python loop lets the LATER item win
    (ultravox_model.py:390-394); the device merge must do the same, bit-exactly in placement."""
    cfg, sd, model, oracle = build(9)
    b = batch_for(cfg, B=2, seconds=2.0, n_text=40, audio_start=5, n_sup=8)
    # give sample 0 both audio items: item 1 starts inside item 0's range
    b["audio_batch_size"] = torch.tensor([2, 0])
    b["audio_token_start_idx"] = torch.tensor([5, 10])
    out = model.forward(**{k: v.to(DEV) for k, v in b.items()})
    with torch.no_grad():
        ref = oracle.forward(**{**b, "audio_values": b["audio_values"].bfloat16().float()})
    assert rel_l2(out.logits, ref["logits"]) < 3e-2


@pytest.mark.parametrize("T,first_label", [(48, 30), (300, 10)])
def test_loss_head_on_supervised_rows_split_k(T, first_label):
    """The loss head runs on the supervised rows only (uvx_llm_fwd/bwd): here with a vocabulary / MLP wide enough that the
    head dgrad takes its split-K path, and (T = 300) with more supervised rows than the split-K cap of 512, so that the
    remainder GEMM + offset scatter run too.  Loss and d loss / d inputs_embeds against the f32 oracle."""
    from oracle.reference_cpu import llama_ref, causal_lm_loss_ref
    tc = dict(SMALL["text_config"], vocab_size=4096, intermediate_size=1024)
    cfg, sd, model, oracle = build(9, text_config=tc)
    torch.manual_seed(3)
    B, D = 2, 256
    emb = (torch.randn(B, T, D) * 0.5).bfloat16()
    labels = torch.randint(0, 4096, (B, T)); labels[:, :first_label] = -100
    labels[1, T - 5:] = -100                                   # ragged supervision
    out = model.language_model_forward(emb.to(DEV), labels=labels.to(DEV), want_logits=False, save_for_bwd=True)
    from ultravox_amd import _lib
    d = model.language_model_backward(1.0)        # uvx_llm_bwd_train: pairs with the forward above
    e = emb.float().requires_grad_(True)
    loss = causal_lm_loss_ref(llama_ref(oracle.sd, cfg, e, None), labels)
    loss.backward()
    assert abs(out.loss.item() - loss.item()) < 2e-2 * loss.item()
    assert rel_l2(d, e.grad) < 6e-2
    # the same step with the full-logits head (option 3 off) agrees to bf16 round-off of one dgrad
    _lib.lib().uvx_set_option(3, 0)
    try:
        out2 = model.language_model_forward(emb.to(DEV), labels=labels.to(DEV), want_logits=False, save_for_bwd=True)
        d2 = model.language_model_backward(1.0)
    finally:
        _lib.lib().uvx_set_option(3, 1)
    assert abs(out2.loss.item() - out.loss.item()) < 1e-6 * out.loss.item()   # same per-row losses; the f32 row sum groups differently
    assert rel_l2(d, d2) < 5e-3
    # ... and so does the plain uvx_llm_fwd / uvx_llm_bwd pair (supervised-rows head, but the last layer on every row):
    # the training pair (uvx_llm_fwd_train / uvx_llm_bwd_train) evaluates the same per-row arithmetic on fewer rows
    model.top_layer_supervised_rows = False
    try:
        out3 = model.language_model_forward(emb.to(DEV), labels=labels.to(DEV), want_logits=False, save_for_bwd=True)
        assert not model._llm_train_pair
        d3 = model.language_model_backward(1.0)
    finally:
        model.top_layer_supervised_rows = True
    assert out3.loss.item() == out.loss.item()
    sup = torch.zeros(B, T, dtype=torch.bool); sup[:, :-1] = labels[:, 1:] != -100
    assert rel_l2(d, d3) < 5e-3
    # rows after the last supervised position of a sequence get no gradient at all, in either pair
    last = [int(sup[b].nonzero().max()) for b in range(B)]
    for b in range(B):
        assert float(d[b, last[b] + 1:].float().abs().max() if last[b] + 1 < T else 0.0) == 0.0


@pytest.mark.parametrize("B,T", [(2, 48), (5, 70), (8, 316)])
def test_two_stream_llm_schedule_is_bit_identical(B, T):
    """uvx_set_option(11, n): the LLM layer chains of the batch slices run on several streams (the caller's and side streams
    forked from / joined into it by events; option value = number of chains).  Same kernels on the same rows: loss, full logits and d loss / d inputs_embeds must be
    BIT-identical to the one-stream schedule - for the plain pair (full logits), the training pair (last layer and head on the
    supervised rows), an odd batch (halves of 3 and 2) and with left / right padding in the attention mask."""
    from ultravox_amd import _lib
    tc = dict(SMALL["text_config"], num_hidden_layers=3)
    cfg, sd, model, oracle = build(13, text_config=tc)
    torch.manual_seed(5)
    emb = (torch.randn(B, T, 256) * 0.5).bfloat16().to(DEV)
    labels = torch.randint(0, 512, (B, T)); labels[:, : T // 2] = -100
    mask = torch.ones(B, T, dtype=torch.long); mask[0, :3] = 0; mask[B - 1, T - 4:] = 0
    labels[B - 1, T - 4:] = -100
    labels, mask = labels.to(DEV), mask.to(DEV)

    def run():
        full = model.language_model_forward(emb, labels=labels, attention_mask=mask, want_logits=True, save_for_bwd=True)
        d_full = model.language_model_backward(1.0).clone()
        tr = model.language_model_forward(emb, labels=labels, attention_mask=mask, want_logits=False, save_for_bwd=True)
        assert model._llm_train_pair
        d_tr = model.language_model_backward(1.0).clone()
        torch.cuda.synchronize()
        return full.logits.clone(), full.loss.clone(), d_full, tr.loss.clone(), d_tr

    L = _lib.lib()
    try:
        L.uvx_set_option(11, 1)
        one = run()
        L.uvx_set_option(11, 2)
        two = run()
        two_again = run()
        more = []
        for n in (3, 4):                           # three / four chains (B = 2: capped at two; B = 5: slices of 2, 1, 1, 1)
            L.uvx_set_option(11, n)
            more.append(run())
    finally:
        L.uvx_set_option(11, 1)                    # the default
    for i, a in enumerate(one):
        for other in (two, two_again, *more):
            assert torch.equal(a, other[i])
    assert torch.isfinite(one[2].float()).all() and one[2].float().abs().max() > 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_streamed_weight_transposes_are_bit_identical(dtype):
    """uvx_config_t.llm_wt_stream (UltravoxModel(stream_weight_transposes=True)): the backward's transposed weight copies are
    made on the fly - lm_head first, then one layer ahead of the layer being differentiated, on a side stream into two
    alternating workspace buffers - instead of being resident (a 70B-parameter LLM then fits one GPU).  Same GEMMs on the
    same operands: loss and d loss / d inputs_embeds BIT-identical to the resident-copy build for the plain pair, the training
    pair and repeated calls (the buffers are reused across calls), and a whole train step gives identical projector gradients."""
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    tc = dict(SMALL["text_config"], num_hidden_layers=5)           # odd depth: both buffers, both parities of the top layer
    cfg = UltravoxConfig(**{**SMALL, "text_config": tc})
    sd = {k: v.to(dtype) for k, v in random_state_dict(cfg, seed=21).items()}
    resident = UltravoxModel(cfg, state_dict=dict(sd), device=DEV, dtype=dtype, stream_weight_transposes=False)
    streamed = UltravoxModel(cfg, state_dict=dict(sd), device=DEV, dtype=dtype, stream_weight_transposes=True)
    assert streamed._c.llm_wt_stream == 1 and resident._c.llm_wt_stream == 0
    assert streamed._llm["lm_head_t"] is None and all(L["wqkv_t"] is None and L["wd_t"] is None for L in streamed._llm["layers"])
    torch.manual_seed(6)
    B, T = 3, 40
    emb = (torch.randn(B, T, 256) * 0.5).to(dtype).to(DEV)
    labels = torch.randint(0, 512, (B, T)); labels[:, : T // 2] = -100
    mask = torch.ones(B, T, dtype=torch.long); mask[0, :3] = 0
    labels, mask = labels.to(DEV), mask.to(DEV)

    def run(m):
        full = m.language_model_forward(emb, labels=labels, attention_mask=mask, want_logits=True, save_for_bwd=True)
        d_full = m.language_model_backward(1.0).clone()
        tr = m.language_model_forward(emb, labels=labels, attention_mask=mask, want_logits=False, save_for_bwd=True)
        d_tr = m.language_model_backward(1.0).clone()
        torch.cuda.synchronize()
        return full.loss.clone(), d_full, tr.loss.clone(), d_tr

    a, b, b2 = run(resident), run(streamed), run(streamed)
    for x, y, z in zip(a, b, b2):
        assert torch.equal(x, y) and torch.equal(x, z)
    assert a[1].float().abs().max() > 0
    batch = {k: v.to(DEV) for k, v in batch_for(cfg).items()}
    batch["audio_values"] = batch["audio_values"].to(dtype)
    for m in (resident, streamed):
        m.train()
    la, lb = resident.forward_backward(**batch), streamed.forward_backward(**batch)
    assert torch.equal(la, lb)
    ga, gb = resident.projector_grads(), streamed.projector_grads()
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


def test_tied_head_llama_train_step_f32_within_1e3():
    """tie_word_embeddings (Llama-3.2-1B / 3B in the reference's recipes): no lm_head.weight - the head IS embed_tokens (one device
    tensor, its transpose made from it); loss, logits and projector gradients against the oracle in f32."""
    from oracle.reference_cpu import OracleModel
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = UltravoxConfig(**{**SMALL, "text_config": dict(SMALL["text_config"], tie_word_embeddings=True)})
    sd = random_state_dict(cfg, seed=17)
    assert cfg.text_config.ties_head and "language_model.lm_head.weight" not in sd
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.float32)
    assert model._llm["lm_head"] is model._llm["embed"]
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    b = batch_for(cfg)
    ref, grads, _ = oracle.train_step(b)
    gb = {k: v.to(DEV) for k, v in b.items()}
    out = model.forward(**gb)
    model.train()
    loss = model.forward_backward(**gb)
    assert (out.logits.cpu() - ref["logits"]).abs().max().item() < 1e-3
    assert abs(loss.item() - ref["loss"].item()) < 1e-4
    mine = model.projector_grads()
    for k, g in grads.items():
        assert rel_l2(mine[k], g) < 2e-3, k


def test_fused_inverse_rope_in_the_attention_backward_is_bit_identical():
    """Option 14 (default on): dq / dk leave the LLM's attention backward already RoPE-inverted (epilogue of the dQ kernel, the
    GQA group reduction) instead of a separate rope pass over d_qkv.  Same arithmetic and rounding points: d inputs_embeds is
    bit-identical - GQA (4 : 2 heads) and, with kv heads = heads, the un-grouped dK epilogue."""
    from ultravox_amd import _lib
    for kv in (2, 4):
        tc = dict(SMALL["text_config"], num_hidden_layers=2, num_key_value_heads=kv)
        cfg, sd, model, oracle = build(14, text_config=tc)
        torch.manual_seed(6)
        B, T = 3, 77
        emb = (torch.randn(B, T, 256) * 0.5).bfloat16().to(DEV)
        labels = torch.randint(0, 512, (B, T)); labels[:, :30] = -100
        mask = torch.ones(B, T, dtype=torch.long); mask[1, :6] = 0

        def run():
            model.language_model_forward(emb, labels=labels.to(DEV), attention_mask=mask.to(DEV), want_logits=False, save_for_bwd=True)
            d = model.language_model_backward(1.0).clone()
            torch.cuda.synchronize()
            return d

        fused = run()
        _lib.lib().uvx_set_option(14, 0)
        try:
            separate = run()
        finally:
            _lib.lib().uvx_set_option(14, 1)
        assert torch.equal(fused, separate) and fused.float().abs().max() > 0


def test_attention_masks_with_holes_are_rejected_on_host_and_device():
    """The device path reduces a mask row to its [first, last] key range, so a mask with holes must not get through silently:
    host masks are checked on the spot, a device mask synchronously the first time its shape is seen, later ones through a
    device flag read at the next synchronisation point (raise_pending_errors) - no per-step host sync."""
    cfg, sd, model, oracle = build(4)
    B, T = 2, 24
    ids = torch.randint(0, 512, (B, T))
    ok = torch.ones(B, T, dtype=torch.long); ok[0, :3] = 0; ok[1, 20:] = 0            # left and right padding: fine
    both = torch.ones(B, T, dtype=torch.long); both[0, :3] = 0; both[0, 21:] = 0       # padding on both sides of one row: fine
    holes = ok.clone(); holes[1, 7:9] = 0
    model.forward(input_ids=ids.to(DEV), attention_mask=ok)                            # host mask
    model.forward(input_ids=ids.to(DEV), attention_mask=both)
    with pytest.raises(ValueError, match="contiguous run"):
        model.forward(input_ids=ids.to(DEV), attention_mask=holes)
    model.forward(input_ids=ids.to(DEV), attention_mask=ok.to(DEV))                    # device mask, first time this shape: sync check
    model.forward(input_ids=ids.to(DEV), attention_mask=holes.to(DEV))                 # same shape again: deferred
    with pytest.raises(ValueError, match="earlier call"):
        model.raise_pending_errors()
    model.raise_pending_errors()                                                       # flag consumed
    with pytest.raises(ValueError, match="contiguous run"):                            # a NEW shape with holes: caught at once
        model.forward(input_ids=ids[:, :20].to(DEV), attention_mask=holes[:, :20].to(DEV))


def test_llm_backward_without_transposed_copies_is_bit_identical():
    """Round 6 (the review's "NN dgrad" item): UltravoxModel(dgrad_nn=True) keeps NO transposed copy of the frozen LLM's linears (only lm_head^T) -
    the backward's dgrads read the forward weights through the GEMM's NN form (uvx_gemm_desc_t.b_kn) - and one training step gives the loss and the
    projector gradients of the model with resident W^T, bit for bit."""
    from oracle.reference_cpu import synthetic_batch
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = UltravoxConfig(**SMALL)
    sd = {k: v.bfloat16() for k, v in random_state_dict(cfg, seed=7).items()}
    b = synthetic_batch(cfg, 2, 3.0, n_text=40, audio_start=5, n_supervised=12)
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV))
    gb = {k: v.to(DEV) for k, v in b.items()}
    res = []
    for nn in (False, True):
        m = UltravoxModel(cfg, state_dict=dict(sd), device=DEV, dtype=torch.bfloat16, dgrad_nn=nn)
        assert all((L["wqkv_t"] is None) == nn for L in m._llm["layers"]) and m._llm["lm_head_t"] is not None
        m.train()
        loss = m.forward_backward(audio_values=mel, **gb)
        res.append((loss.clone(), {k: v.clone() for k, v in m.projector_grads().items()}))
    (l0, g0), (l1, g1) = res
    assert torch.equal(l0, l1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), (k, int((g0[k] != g1[k]).sum()))


@pytest.mark.parametrize("ln_mid", [True, False])
def test_training_step_is_bit_reproducible(ln_mid):
    """Five repetitions of one forward + backward give the same loss and the same projector gradients bit for bit - including the RMSNorm
    weight gradients, whose per-block partial sums are reduced in block order since round 6 (they were summed with f32 atomics before and
    moved by an ulp from run to run: found by this round's epilogue bit-identity tests)."""
    from oracle.reference_cpu import synthetic_batch
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = UltravoxConfig(**dict(SMALL, projector_ln_mid=ln_mid))
    sd = {k: v.bfloat16() for k, v in random_state_dict(cfg, seed=11).items()}
    b = synthetic_batch(cfg, 3, 3.0, n_text=40, audio_start=5, n_supervised=12)
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV))
    gb = {k: v.to(DEV) for k, v in b.items()}
    m = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16)
    m.train()
    first = None
    for _ in range(5):
        loss = m.forward_backward(audio_values=mel, **gb)
        torch.cuda.synchronize()
        cur = (loss.clone(), m.proj_grad.clone())
        if first is None:
            first = cur
        assert torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1]), int((cur[1] != first[1]).sum())
    assert first[1].abs().sum().item() > 0
