"""Parity AT THE DEPTH THAT IS BENCHMARKED.  `bench.py` times BASELINE.json configs[1] at full depth (whisper-medium 24 layers
+ Llama-3-8B 32 layers); the other width tests stop at depth 2 / 8.  Here one whole adapter-train step of that model at
B = 1 x 30 s (and of C3 = whisper-large-v3 + Llama-3-8B, and of C5 = wav2vec2-large + Gemma-7B) runs through the production bf16
HIP path and through the f32 CPU oracle (oracle/reference_cpu.py, ~12 s on the box's host cores) on the same bf16-rounded weights
and the same inputs.  Bars (round 4: 1.5 x what round 3 recorded at this depth, profiles/r03_parity/c{2,3}_full_depth.json - logits
1.85e-2 / 2.1e-2, gradients 2.0-2.4e-2, loss 0.035 % / 0.045 %): logits rel-L2 <= 2.8e-2 (C3: 3.2e-2), projector gradients <= 3.6e-2,
loss within 0.2 %, encoder output <= 1.5e-2 (C3 1.8e-2), audio embeddings <= 2.2e-2 (C3 2.5e-2), argmax agreement on the supervised
rows >= 0.9 (C2; 0.85 elsewhere: 32 rows, one row = 3 %).  CALIBRATION at this depth: the same restatement run by torch-ROCm in
bf16 ON THE GPU (flash-rounded attention) is the second opinion - the HIP path's distance to the f32 oracle must be <= 1.25 x torch's
own bf16 distance to it, for the tower output, logits, audio embeddings and every projector gradient (C5: 1.5 x, see below).  Per-stage errors are recorded to
gpurun_out/parity/ (committed as profiles/rNN_parity/*_full_depth.json).  Needs ~45 GB of host memory for the f32 oracle weights:
skipped below 48 GB.

Second test: the f32 compute mode (north_star's "logits within 1e-3") at the C2 WIDTH - the real tile shapes of the exact-f32
matrix-core GEMM (K = 4096 / 14336, N = 128256) - at depth 2."""
import time

import pytest
import torch

from parity_util import oracle_threads, record, rel_l2, stage_errors

pytestmark = pytest.mark.gpu
DEV = "cuda"

# bars per workload: 1.5 x the round-3 records (C5: Gemma-7B has no round-3 full-depth record; its width-test picture, 1.5 x)
BARS = {
    "c2": dict(encoder_out=1.5e-2, audio_embeds=2.2e-2, logits=2.8e-2, grads=3.6e-2, loss=2e-3, argmax=0.9),
    "c3": dict(encoder_out=1.8e-2, audio_embeds=2.5e-2, logits=3.2e-2, grads=3.6e-2, loss=2e-3, argmax=0.85),
    "c5": dict(encoder_out=2.2e-2, audio_embeds=3.4e-2, logits=4.2e-2, grads=7.3e-2, loss=2e-3, argmax=0.9),
}


def host_available_gb() -> float:
    try:
        return int(next(l for l in open("/proc/meminfo") if l.startswith("MemAvailable")).split()[1]) / 2 ** 20
    except Exception:
        return 0.0


@pytest.mark.parametrize("workload", ["c2", "c3", "c5"])
def test_full_depth_train_step_matches_oracle(workload):
    from bench import WORKLOADS
    from oracle.reference_cpu import OracleModel, fused_attention, logmel_ref, synthetic_batch, wav2vec2_normalize_ref
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    if host_available_gb() < 48:
        pytest.skip(f"{host_available_gb():.0f} GB of host memory available; the f32 oracle of an 8B-parameter LLM needs ~45 GB")
    wl, bars = WORKLOADS[workload], BARS[workload]
    cfg = UltravoxConfig(audio_model_id=wl["audio"], text_model_id=wl["text"], hidden_size=4096, stack_factor=8,
                         projector_ln_mid=True, torch_dtype="bfloat16")
    a, t = cfg.audio_config, cfg.text_config
    w2v = bool(getattr(a, "is_wav2vec2", False))
    sd = random_state_dict(cfg, seed=7, dtype=torch.bfloat16, device="cuda")        # every layer its own weights
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16, rope_len=512)
    threads = oracle_threads()
    t0 = time.perf_counter()
    oracle = OracleModel(cfg, sd, dtype=torch.float32)                              # same bf16-rounded values, f32 arithmetic
    t_load = time.perf_counter() - t0
    second = OracleModel(cfg, sd, dtype=torch.bfloat16, device=DEV)                 # torch-ROCm bf16 on the GPU: the second opinion
    del sd
    torch.cuda.empty_cache()
    b = synthetic_batch(cfg, 1, wl["seconds"], n_text=128, audio_start=16, n_supervised=32)
    pcm = b.pop("pcm")
    rec = {"workload": wl["name"], "encoder_layers": a.encoder_layers, "llm_layers": t.num_hidden_layers, "clips": 1,
           "seq_len": int(b["input_ids"].shape[1]), "oracle_threads": threads, "oracle_weight_load_s": t_load, "stages": {}}
    if w2v:      # C5: the wav2vec2 tower reads normalised PCM (no mel stage)
        vals = wav2vec2_normalize_ref(pcm).bfloat16().to(DEV)
    else:
        vals = WhisperFeatureExtractor(a.num_mel_bins).logmel_device(pcm.to(DEV))
        rec["stages"]["mel"] = stage_errors(vals, logmel_ref(pcm, a.num_mel_bins))
    gb = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}
    # ---- oracle: one whole step (the tower once; its output is reused for the stage comparison) ----
    t0 = time.perf_counter()
    vals_cpu = vals.cpu().bfloat16().float()      # the device input, so that the comparison isolates the model path (mel is above)
    with torch.no_grad():
        tower_ref, _ = oracle.audio_embeds(vals_cpu, None if w2v else b["audio_lens"])
    for k in oracle.trainable:
        oracle.sd[k].grad = None
    ref = oracle.forward(audio_values=vals_cpu, tower_output=tower_ref, **b)
    ref["loss"].backward()
    grads = {k: oracle.sd[k].grad for k in oracle.trainable}
    rec["oracle_step_s"] = time.perf_counter() - t0
    # ---- HIP path, stage by stage ----
    tower = model.audio_tower_forward(vals, None if w2v else gb["audio_lens"])
    rec["stages"]["encoder_out"] = stage_errors(tower, tower_ref)
    emb = model.multi_modal_projector_forward(tower)
    Na = int(b["audio_token_len"][0])
    rec["stages"]["audio_embeds"] = stage_errors(emb[:, :Na], ref["audio_embeds"].detach()[:, :Na])
    out = model.forward(audio_values=vals, **gb)                                    # full logits + loss
    rec["stages"]["logits"] = stage_errors(out.logits, ref["logits"].detach())
    rec["loss_hip_full_logits"], rec["loss_oracle"] = out.loss.item(), ref["loss"].item()
    model.train()
    loss = model.forward_backward(audio_values=vals, **gb)                          # the step bench.py times (supervised-row head)
    rec["loss_hip_train_step"] = loss.item()
    mine = model.projector_grads()
    rec["grads_rel_l2"] = {k: rel_l2(mine[k], g) for k, g in grads.items()}
    ref_top = ref["logits"][0].argmax(-1)
    hip_top = out.logits[0].float().argmax(-1).cpu()
    agree = (hip_top[-32:] == ref_top[-32:]).float().mean().item()
    rec["argmax_agreement_supervised_rows"] = agree
    rec["argmax_agreement_all_rows"] = (hip_top == ref_top).float().mean().item()
    # ---- second opinion: the same restatement in torch-ROCm bf16 on the GPU (flash-rounded attention), whole step ----
    t0 = time.perf_counter()
    with torch.device(DEV), fused_attention():
        r16 = second.forward(audio_values=vals.bfloat16(), **gb)      # its own tower (C5: conv feature encoder through MIOpen)
        r16["loss"].backward()
        with torch.no_grad():
            tower16, _ = second.audio_embeds(vals.bfloat16(), None if w2v else gb["audio_lens"])
    torch.cuda.synchronize()
    rec["torch_bf16_gpu_step_s"] = time.perf_counter() - t0
    cal = {"logits": (out.logits, r16["logits"].detach().cpu(), ref["logits"].detach()),
           "audio_embeds": (emb[:, :Na], r16["audio_embeds"].detach()[:, :Na].cpu(), ref["audio_embeds"].detach()[:, :Na])}
    cal["encoder_out"] = (tower, tower16.cpu(), tower_ref)
    cal.update({"grad." + k.split(".", 1)[1]: (mine[k], second.sd[k].grad.cpu(), g) for k, g in grads.items()})
    rec["calibration"] = {k: {"hip_vs_f32": rel_l2(h, f), "torch_bf16_vs_f32": rel_l2(t16, f), "hip_vs_torch_bf16": rel_l2(h, t16)}
                          for k, (h, t16, f) in cal.items()}
    rec["calibration"]["loss"] = {"hip": loss.item(), "torch_bf16": r16["loss"].item(), "f32": ref["loss"].item()}
    record(f"{workload}_full_depth", rec)
    if not w2v:
        assert rec["stages"]["mel"]["max_abs"] < 2e-4
    assert rec["stages"]["encoder_out"]["rel_l2"] < bars["encoder_out"], rec["stages"]
    assert rec["stages"]["audio_embeds"]["rel_l2"] < bars["audio_embeds"], rec["stages"]
    assert rec["stages"]["logits"]["rel_l2"] < bars["logits"], rec["stages"]
    assert abs(out.loss.item() - ref["loss"].item()) < bars["loss"] * abs(ref["loss"].item())
    assert abs(loss.item() - ref["loss"].item()) < bars["loss"] * abs(ref["loss"].item())
    for k, e in rec["grads_rel_l2"].items():
        assert e < bars["grads"], (k, e)
    assert agree >= bars["argmax"], (agree, rec["argmax_agreement_all_rows"])
    # C5: 1.5 instead of 1.25.  This test sees ONE 30 s clip, and at depth 24 the wav2vec2 tower's hip / torch distance ratio is a sample
    # of a quantity that scatters 0.84 ... 1.14 from clip to clip (round 5, profiles/r05_c5_tower_stage_probe.txt: the post-LN stack doubles
    # the error between layers 12 and 24; this seed's clip reads 1.14-1.19, the mean over 8 clips 0.97, at every depth and batch size HIP is
    # at or below torch's distance on average - the stem's rounding, blamed in round 4, measures 0.87-0.96 at depth 1).  The calibrated bar
    # on the tower is tests/test_wav2vec2_gpu.py::test_wav2vec2_large_tower_bf16_distance_is_calibrated_over_clips (mean over clips <= 1.08);
    # the Gemma stack alone is calibrated at 1.00 (profiles/r04_parity/c5_text_only_calibration_probe.txt).
    factor = 1.5 if w2v else 1.25
    for k, v in rec["calibration"].items():
        if k != "loss":
            assert v["hip_vs_f32"] <= factor * v["torch_bf16_vs_f32"] + 1e-4, (k, v)


def test_c2_width_f32_mode_logits_within_1e3():
    """f32 compute mode at the C2 width (Llama-3-8B 4096 / 14336 / 128256, whisper-medium 1024 / 4096), depth 2, 2 x 30 s:
    max |logit - oracle logit| <= 1e-3 (north_star's tolerance) on the real tile shapes, loss 1e-4, gradients 2e-3."""
    from oracle.reference_cpu import OracleModel, synthetic_batch
    from parity_util import width_config
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = width_config("meta-llama/Meta-Llama-3-8B-Instruct", "openai/whisper-medium", 2, 2)
    cfg.torch_dtype = "float32"
    sd = random_state_dict(cfg, seed=5, dtype=torch.float32, device="cuda")
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.float32, rope_len=512)
    oracle_threads()
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    del sd
    b = synthetic_batch(cfg, 2, 30.0, n_text=128, audio_start=16, n_supervised=32)
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV))
    gb = {k: v.to(DEV) for k, v in b.items()}
    ob = {**b, "audio_values": mel.cpu()}
    ref, grads, _ = oracle.train_step(ob)
    out = model.forward(audio_values=mel, **gb)
    err = (out.logits.cpu() - ref["logits"].detach()).abs().max().item()
    model.train()
    loss = model.forward_backward(audio_values=mel, **gb)
    mine = model.projector_grads()
    g_err = {k: rel_l2(mine[k], g) for k, g in grads.items()}
    record("c2_width_f32_mode", {"max_abs_logit_diff": err, "logit_rms": ref["logits"].detach().pow(2).mean().sqrt().item(),
                                 "loss_hip": out.loss.item(), "loss_oracle": ref["loss"].item(), "grads_rel_l2": g_err})
    assert err < 1e-3, f"max |logit diff| = {err}"
    assert abs(out.loss.item() - ref["loss"].item()) < 1e-4 and abs(loss.item() - ref["loss"].item()) < 1e-4
    for k, e in g_err.items():
        assert e < 2e-3, (k, e)
