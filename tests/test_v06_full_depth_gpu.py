"""The reference's v0.6 backbones AT FULL DEPTH ("q3 / g3 have parity only at depth 1-2").  An f32 oracle of a 27-32 B
parameter LLM does not fit the box's host memory, so the check here is the second opinion alone: the same restatement
(oracle/reference_cpu.py) run by torch-ROCm in bf16 ON THE GPU, whole adapter-train step at B = 1 x 30 s, against the production HIP path
on the same weights.  Two bf16 pipelines that differ only in rounding order sit ~sqrt(2) x (one pipeline's distance to f32) apart - 2.6-3e-2
at these depths (C2 / C3: profiles/r04_parity) - while a depth-dependent bug (a layer-indexed table, a stash slot, a local / global layer
flag) shows as O(1).  Bars: logits rel-L2 < 6e-2, loss within 0.5 %, projector gradients < 8e-2.  Records: gpurun_out/parity/*_full_depth.json."""
import pytest
import torch

from parity_util import record, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("workload", ["g3", "q3"])
def test_v06_backbone_full_depth_matches_torch_bf16_on_the_gpu(workload):
    from bench import WORKLOADS
    from oracle.reference_cpu import OracleModel, fused_attention, synthetic_batch
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    import gc
    wl = WORKLOADS[workload]
    need = {"g3": 175, "q3": 210}[workload]
    gc.collect()
    torch.cuda.empty_cache()
    free = torch.cuda.mem_get_info()[0] / 2 ** 30
    if free < need:
        pytest.skip(f"{free:.0f} GiB free on the device, {need} GiB needed (weights + transposed copies + the bf16 second opinion)")
    cfg = UltravoxConfig(audio_model_id=wl["audio"], text_model_id=wl["text"], hidden_size=4096, stack_factor=8,
                         projector_ln_mid=True, torch_dtype="bfloat16")
    sd = random_state_dict(cfg, seed=11, dtype=torch.bfloat16, device="cuda")
    second = OracleModel(cfg, sd, dtype=torch.bfloat16, device=DEV)              # its own copy of the weights ...
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16, rope_len=512, consume_state_dict=True)   # ... this one eats `sd`
    del sd
    torch.cuda.empty_cache()
    b = synthetic_batch(cfg, 1, wl["seconds"], n_text=128, audio_start=16, n_supervised=32)
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV))
    gb = {k: v.to(DEV) for k, v in b.items()}
    out = model.forward(audio_values=mel, **gb)
    model.train()
    loss = model.forward_backward(audio_values=mel, **gb)
    mine = model.projector_grads()
    with torch.device(DEV), fused_attention():
        r16 = second.forward(audio_values=mel.bfloat16(), **gb)
        r16["loss"].backward()
    rec = {"workload": wl["name"], "llm_layers": cfg.text_config.num_hidden_layers, "encoder_layers": cfg.audio_config.encoder_layers,
           "logits_hip_vs_torch_bf16": rel_l2(out.logits, r16["logits"].detach()),
           "loss": {"hip": loss.item(), "torch_bf16": r16["loss"].item()},
           "grads_hip_vs_torch_bf16": {k: rel_l2(mine[k], second.sd[k].grad) for k in second.trainable},
           "argmax_agreement_all_rows": (out.logits[0].float().argmax(-1) == r16["logits"][0].float().argmax(-1)).float().mean().item()}
    record(f"{workload}_full_depth", rec)
    del model, second, r16, out, mine
    gc.collect()
    torch.cuda.empty_cache()
    assert rec["logits_hip_vs_torch_bf16"] < 6e-2, rec
    assert abs(rec["loss"]["hip"] - rec["loss"]["torch_bf16"]) < 5e-3 * abs(rec["loss"]["torch_bf16"]), rec["loss"]
    for k, e in rec["grads_hip_vs_torch_bf16"].items():
        assert e < 8e-2, (k, e)
