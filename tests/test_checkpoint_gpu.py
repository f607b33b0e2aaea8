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
