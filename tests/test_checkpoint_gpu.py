"""Save -> load -> resume on the GPU: an interrupted run continues bit-identically (HF Trainer resume contract)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _batch(cfg, rank=0):
    from oracle.reference_cpu import synthetic_batch
    from ultravox_amd.frontend import WhisperFeatureExtractor
    b = synthetic_batch(cfg, 2, 2.0, n_text=24, audio_start=5, n_supervised=8, rank=rank)
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV))
    return {"audio_values": mel, **{k: v.to(DEV) for k, v in b.items()}}


@pytest.mark.parametrize("master", [False, True])
def test_resume_is_bit_identical(tmp_path, master):
    from test_model_gpu import SMALL
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel, UltravoxTrainer
    from ultravox_amd.weights import random_state_dict
    cfg = UltravoxConfig(**SMALL)
    sd = random_state_dict(cfg, seed=5, dtype=torch.bfloat16)
    batches = [_batch(cfg, r) for r in range(3)]
    # uninterrupted: 3 steps
    m1 = UltravoxModel(cfg, state_dict=sd, device=DEV)
    t1 = UltravoxTrainer(m1, lr=2e-3, master_weights=master)
    for b in batches:
        t1.train_step(**b)
    # interrupted after 2 steps, checkpointed, resumed in a fresh process-equivalent (new model from the BASE weights)
    m2 = UltravoxModel(cfg, state_dict=sd, device=DEV)
    t2 = UltravoxTrainer(m2, lr=2e-3, master_weights=master)
    for b in batches[:2]:
        t2.train_step(**b)
    t2.save_checkpoint(str(tmp_path))
    m3 = UltravoxModel.from_pretrained(str(tmp_path), base_state_dict=sd, device=DEV)
    assert m3.keep_params == set(m3.projector_state_dict().keys())
    for k, v in m3.projector_state_dict().items():
        assert torch.equal(v, m2.projector_state_dict()[k]), k
    t3 = UltravoxTrainer(m3, lr=2e-3, master_weights=master)
    t3.load_checkpoint(str(tmp_path))
    assert t3.step_count == 2
    l3 = t3.train_step(**batches[2])
    for k, v in m3.projector_state_dict().items():
        assert torch.equal(v, m1.projector_state_dict()[k]), k
    assert torch.equal(t3.exp_avg, t1.exp_avg) and torch.equal(t3.exp_avg_sq, t1.exp_avg_sq)
    # the checkpoint holds only the trainable keys: the frozen towers are not in it
    from ultravox_amd import checkpoint
    _, ck = checkpoint.load_pretrained(str(tmp_path))
    assert all(k.startswith("multi_modal_projector.") for k in ck)


def test_load_state_dict_rejects_foreign_keys_and_shapes():
    from test_model_gpu import SMALL
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    m = UltravoxModel(UltravoxConfig(**SMALL), device=DEV, seed=2)
    k = "multi_modal_projector.linear_1.weight"
    with pytest.raises(ValueError):
        m.load_state_dict({k: torch.zeros(3, 3)})
    with pytest.raises(KeyError):
        m.load_state_dict({"something.else": torch.zeros(1)})
    w = torch.full_like(m.projector_state_dict()[k], 0.5)
    m.load_state_dict({k: w})
    assert torch.equal(m.projector_state_dict()[k], w) and k in m.keep_params


def test_overlapped_all_reduce_schedule_is_bit_identical():
    """UltravoxTrainer(overlap_comm=True): the gradient all-reduce is asynchronous and clip + AdamW are deferred until the
    next step reaches the projector.  Exercised here on a 1-rank RCCL group (the collective machinery runs, the reduction
    is the identity): parameters after 3 steps must equal the sequential schedule's bit for bit, audio and text-only
    batches mixed."""
    import os
    import torch.distributed as dist
    from test_model_gpu import SMALL
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel, UltravoxTrainer
    from ultravox_amd.weights import random_state_dict
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        cfg = UltravoxConfig(**SMALL)
        sd = random_state_dict(cfg, seed=8, dtype=torch.bfloat16)
        batches = [_batch(cfg, r) for r in range(3)]
        text_only = {k: v for k, v in batches[1].items() if k in ("input_ids", "attention_mask", "labels")}
        seq = [batches[0], text_only, batches[2], batches[1]]
        res = []
        for overlap in (False, True):
            m = UltravoxModel(cfg, state_dict=sd, device=DEV)
            t = UltravoxTrainer(m, lr=2e-3, overlap_comm=overlap)
            losses = [t.train_step(**b).item() for b in seq]
            t.flush()
            res.append((losses, {k: v.clone() for k, v in m.projector_state_dict().items()}, t.exp_avg.clone(), t.step_count))
        assert res[0][0] == res[1][0] and res[0][3] == res[1][3] == 4
        for k in res[0][1]:
            assert torch.equal(res[0][1][k], res[1][1][k]), k
        assert torch.equal(res[0][2], res[1][2])
    finally:
        if created:
            dist.destroy_process_group()


def test_gradient_accumulation_and_schedule_on_the_device():
    """gradient_accumulation_steps = 2 with a warm-up schedule: two micro-batches, ONE optimizer step whose gradient is the
    sum of the two (1/2)-scaled micro-batch gradients and whose lr is the schedule's — equal, bit for bit, to doing the
    same by hand on a second model."""
    from test_model_gpu import SMALL
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel, UltravoxTrainer
    from ultravox_amd.weights import random_state_dict
    cfg = UltravoxConfig(**SMALL)
    sd = random_state_dict(cfg, seed=6, dtype=torch.bfloat16)
    b1, b2 = _batch(cfg, 0), _batch(cfg, 1)
    sched = dict(lr_scheduler="constant_with_warmup", lr_warmup_steps=2)
    m1 = UltravoxModel(cfg, state_dict=sd, device=DEV)
    t1 = UltravoxTrainer(m1, lr=2e-3, gradient_accumulation_steps=2, **sched)
    for _ in range(2):                      # 4 micro-batches = 2 optimizer steps (lr 0, then lr / 2)
        t1.train_step(**b1)
        t1.train_step(**b2)
    assert t1.step_count == 2 and t1.last_lr == pytest.approx(1e-3)
    m2 = UltravoxModel(cfg, state_dict=sd, device=DEV)
    t2 = UltravoxTrainer(m2, lr=2e-3, **sched)
    for _ in range(2):
        m2.forward_backward(grad_scale=0.5, **b1)
        g = m2.proj_grad.clone()
        m2.forward_backward(grad_scale=0.5, **b2)
        m2.proj_grad.add_(g)
        t2.optimizer_step()
    assert torch.equal(m1.proj_flat, m2.proj_flat) and torch.equal(t1.exp_avg, t2.exp_avg)
    assert not torch.equal(m1.proj_flat, UltravoxModel(cfg, state_dict=sd, device=DEV).proj_flat)     # the second step moved the weights


def test_lora_checkpoint_round_trips_through_from_pretrained(tmp_path):
    """A checkpoint this framework wrote with encoder + LLM LoRA adapters loads back through from_pretrained onto a base
    state dict that carries NO adapter keys (what a fresh process has), and re-saves the same tensors."""
    from test_model_gpu import SMALL
    from ultravox_amd import checkpoint
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel, UltravoxTrainer
    from ultravox_amd.weights import init_lora_state_dict, random_state_dict
    cfg = UltravoxConfig(**{**SMALL, "audio_model_lora_config": {"r": 4}, "text_model_lora_config": {"r": 2}})
    base = random_state_dict(cfg, seed=8, dtype=torch.bfloat16)
    assert not any(".lora_" in k for k in base)
    m1 = UltravoxModel(cfg, state_dict={**base, **init_lora_state_dict(cfg, seed=8, dtype=torch.bfloat16, random_b=True)}, device=DEV)
    UltravoxTrainer(m1, lr=2e-3).train_step(**_batch(cfg))          # move projector and adapters off their initial values
    m1.save_pretrained(str(tmp_path / "a"))
    m2 = UltravoxModel.from_pretrained(str(tmp_path / "a"), base_state_dict=base, device=DEV)
    s1, s2 = m1.projector_state_dict(), m2.projector_state_dict()
    assert set(s1) == set(s2) and sum(".lora_" in k for k in s2) == 2 * 2 * (2 + 2)
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k
    m2.save_pretrained(str(tmp_path / "b"))
    _, a = checkpoint.load_pretrained(str(tmp_path / "a"))
    _, b = checkpoint.load_pretrained(str(tmp_path / "b"))
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    b_ = _batch(cfg)
    assert m1.forward(**b_).loss.item() == m2.forward(**b_).loss.item()


def test_tower_keys_carried_by_a_checkpoint_are_re_saved(tmp_path):
    """ultravox_model.py:565-591: every keep_param is written again.  A reference checkpoint that carries audio_tower.* /
    language_model.* tensors (fine-tuned towers) must not lose them on the next save - a reload would silently revert those
    towers to their base model ids."""
    from test_model_gpu import SMALL
    from ultravox_amd import checkpoint
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = UltravoxConfig(**SMALL)
    base = random_state_dict(cfg, seed=9, dtype=torch.bfloat16)
    tower_keys = ["audio_tower.layers.0.fc1.weight", "language_model.model.layers.1.mlp.down_proj.weight", "language_model.model.norm.weight"]
    carried = {k: (base[k].float() * 1.5 + 0.01).bfloat16() for k in tower_keys}
    carried.update({k: v for k, v in base.items() if k.startswith("multi_modal_projector.")})
    checkpoint.save_pretrained(str(tmp_path / "ref"), cfg, carried, [k for k in carried if k.startswith("multi_modal_projector.")], tower_keys)
    m = UltravoxModel.from_pretrained(str(tmp_path / "ref"), base_state_dict=base, device=DEV)
    assert set(tower_keys) <= m.keep_params
    plain = UltravoxModel(cfg, state_dict=base, device=DEV)
    b_ = _batch(cfg)
    assert m.forward(**b_).loss.item() != plain.forward(**b_).loss.item()       # the carried tower tensors are in use
    m.save_pretrained(str(tmp_path / "again"))
    _, again = checkpoint.load_pretrained(str(tmp_path / "again"))
    assert set(again) == set(carried) and all(torch.equal(again[k], carried[k]) for k in carried)
    # a keep_param added by NAME (reference-style code does that, ultravox_model.py:59): the tensor is read back from the packed device
    # weights under its checkpoint name (round 5: weights.unpack_*) - here a frozen-tower matrix nobody retained on the host
    m.keep_params.add("audio_tower.layers.1.fc2.weight")
    m.save_pretrained(str(tmp_path / "named"))
    _, named = checkpoint.load_pretrained(str(tmp_path / "named"))
    assert set(named) == set(carried) | {"audio_tower.layers.1.fc2.weight"}
    assert torch.equal(named["audio_tower.layers.1.fc2.weight"], base["audio_tower.layers.1.fc2.weight"])
    assert all(torch.equal(named[k], carried[k]) for k in carried)          # retained originals still win over the read-back
    # a keep_param with no tensor behind it at all: the save is refused (a checkpoint must not lose tensors silently);
    # strict=False reports it and leaves the key out
    m.keep_params.add("audio_tower.layers.1.fc2.no_such_tensor")
    with pytest.raises(KeyError, match="cannot re-save"):
        m.save_pretrained(str(tmp_path / "refused"))
    with pytest.warns(UserWarning, match="cannot re-save"):
        m.save_pretrained(str(tmp_path / "partial"), strict=False)
    _, partial = checkpoint.load_pretrained(str(tmp_path / "partial"))
    assert set(partial) == set(named)
    with pytest.raises(KeyError, match="cannot re-save"):
        m._full_state_dict(strict=True)
