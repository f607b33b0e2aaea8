"""bench.py — adapter-train step of the Ultravox audio->LLM hot path on N MI355X GPUs of one node.

    python bench.py                                         # = --gpus 1 --steps 10 --warmup 5
    python bench.py --gpus 8 --steps 20 --warmup 5          # launches its own 8 ranks (torch.distributed.run, 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): audio-seconds / second / node for one adapter-train step (Whisper-medium encoder +
Llama-3-8B, both frozen; projector trained).  One "step" = log-mel (K1) -> encoder -> projector -> embed/merge
-> LLM forward + CE -> activation-gradient backward -> projector dgrad/wgrad -> DP gradient mean (RCCL) ->
clip + AdamW, over one synthetic batch of B = 8 clips x 30 s (16 kHz) + 128 text tokens per GPU, inputs
resident in HBM before the timed region.  Weights: seeded random init at the true architecture shapes (no
checkpoints offline); data: synthetic.  Weak scaling: per-GPU work is fixed.

The JSON line also carries
  roofline     — the dominant kernel (bf16 MFMA GEMM): algorithmic FLOPs of every launch / its duration,
                 timed live with HIP events on the launch stream inside the timed region, vs 2.5 PFLOP/s.  The events
                 are attached to the GEMM dispatches themselves (hipExtLaunchKernelGGL) of every 4th timed step
                 (--prof-every; `roofline.profiled_steps`): a timed dispatch costs ~1.4 us, and the steps in between
                 run exactly as a training step does;
  cpu_baseline — the CPU oracle (oracle/reference_cpu.py, a port of the reference path) timed on this box's
                 host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md; 2:1-sparse figures excluded)

WORKLOADS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "c2": dict(name="Llama-3-8B (frozen) + whisper-medium, bs=8x30s clips, adapter train",
               audio="openai/whisper-medium", text="meta-llama/Meta-Llama-3-8B-Instruct", B=8, seconds=30.0),
    # BASELINE.json configs[2] per-rank shapes (global batch 64 = 8 clips on each of 8 GPUs): not the quoted configuration,
    # selectable for the DP-8 run (`--gpus 8 --workload c3`)
    "c3": dict(name="Llama-3-8B (frozen) + whisper-large-v3, bs=8x30s clips per GPU, adapter train",
               audio="openai/whisper-large-v3", text="meta-llama/Meta-Llama-3-8B-Instruct", B=8, seconds=30.0),
    # BASELINE.json configs[4] per-rank shapes (alt encoder / backbone; the reference quotes it at DP = 4): not the quoted configuration
    "c5": dict(name="Gemma-7B (frozen) + wav2vec2-large (frozen), bs=8x30s clips per GPU, adapter train",
               audio="facebook/wav2vec2-large-960h", text="google/gemma-7b", B=8, seconds=30.0),
    # not a BASELINE.json configuration: the reference's v0.6 recipe (ultravox/training/configs/v0.6_config_qwen3_32b.yaml: Qwen/Qwen3-32B
    # behind whisper-large-v3-turbo), per-rank shapes as c2 / c3.  64 GB of frozen bf16 weights + their transposed copies on one GPU.
    "q3": dict(name="Qwen3-32B (frozen) + whisper-large-v3-turbo, bs=8x30s clips per GPU, adapter train",
               audio="openai/whisper-large-v3-turbo", text="Qwen/Qwen3-32B", B=8, seconds=30.0),
    # the reference's other v0.6 recipe (v0.6_config_gemma3_27b.yaml): the text stack of google/gemma-3-27b-it
    "g3": dict(name="Gemma-3-27B (frozen) + whisper-large-v3-turbo, bs=8x30s clips per GPU, adapter train",
               audio="openai/whisper-large-v3-turbo", text="google/gemma-3-27b-it", B=8, seconds=30.0),
    # not a BASELINE.json configuration (its configs[3] is 70B INFERENCE): the reference's 70B TRAINING recipes
    # (v0.6_config_llama3_70b.yaml: Llama-3.3-70B behind whisper-large-v3-turbo), per-rank shapes as c2.  141 GB of frozen bf16 weights
    # on ONE 288 GB GPU - possible because the backward's transposed copies are streamed (uvx_config_t.llm_wt_stream), not resident
    "l70": dict(name="Llama-3.3-70B (frozen) + whisper-large-v3-turbo, bs=8x30s clips per GPU, adapter train",
               audio="openai/whisper-large-v3-turbo", text="meta-llama/Llama-3.3-70B-Instruct", B=8, seconds=30.0),
    # BASELINE.json configs[3]: 70B INFERENCE, one replica per GPU (TP = 1; the reference's x8 is eight independent replicas = --gpus 8):
    # `--workload c4` times generate() = encoder + projector + LLM prefill over 30 s audio + 128 text tokens, then `new_tokens` greedy
    # decode steps; value = decoded tokens / s, roofline = weight bytes streamed per decoded token against HBM (run_inference below)
    "c4": dict(name="Llama-3.3-70B (frozen, bf16) + whisper-medium, inference prefill+decode, TP=1 per replica",
               audio="openai/whisper-medium", text="meta-llama/Llama-3.3-70B-Instruct", B=1, seconds=30.0, inference=True, new_tokens=32),
    # the same loop on the C2 model (8B): a quick check of the inference path, not a BASELINE.json configuration
    "c4s": dict(name="Llama-3-8B (frozen, bf16) + whisper-medium, inference prefill+decode",
                audio="openai/whisper-medium", text="meta-llama/Meta-Llama-3-8B-Instruct", B=1, seconds=30.0, inference=True, new_tokens=32),
    # plumbing-sized inference loop (TinyLlama + whisper-tiny, 4 s clips): what the 2-rank replica test runs
    "c4t": dict(name="TinyLlama-1.1B + whisper-tiny, inference prefill+decode (plumbing check)",
                audio="openai/whisper-tiny", text="TinyLlama/TinyLlama-1.1B-Chat-v1.0", B=1, seconds=4.0, inference=True, new_tokens=8),
    # BASELINE.json configs[0] shapes (plumbing-sized), for quick checks
    "c1": dict(name="TinyLlama-1.1B + whisper-tiny, 1x4s clip, adapter train",
               audio="openai/whisper-tiny", text="TinyLlama/TinyLlama-1.1B-Chat-v1.0", B=1, seconds=4.0),
}


PEAK_HBM_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured copy, 7.0-7.1 TB/s non-temporal read stream)


def llm_weight_bytes(cfg) -> float:
    """bf16 bytes of the frozen LLM that ONE decode step streams: every layer's q|k|v, o, gate|up, down matrices + the LM head
    (the embedding table is a gather of B rows; norms are negligible)."""
    t = cfg.text_config
    D, I, L, V = t.hidden_size, t.intermediate_size, t.num_hidden_layers, t.vocab_size
    qkv = (t.num_attention_heads + 2 * t.num_key_value_heads) * t.head_dim * D
    o = t.num_attention_heads * t.head_dim * D
    return 2.0 * (L * (qkv + o + 3 * D * I) + V * D)


def pmc_decode_traffic_per_token(profiles_dir=None):
    """roofline.traffic of `--workload c4`: memory-side read bytes of the decode GEMVs per token from the newest committed PMC summary
    (profiles/rNN_pmc_decode_traffic.json, written from tools/pmc_decode_traffic.sh's separate FETCH_SIZE pass), or None."""
    d = profiles_dir or os.path.join(ROOT, "profiles")
    try:
        names = sorted(n for n in os.listdir(d) if n.startswith("r") and n.endswith("_pmc_decode_traffic.json"))
        return json.load(open(os.path.join(d, names[-1]))).get("traffic_bytes_per_token") if names else None
    except (OSError, ValueError):
        return None


def run_inference(args, wl, dev) -> None:
    """`--workload c4 / c4s`: a "step" = one generate() call (encoder + projector + prefill + new_tokens greedy decode steps) on B
    prompts resident in HBM.  Prefill is also timed alone (max_new_tokens = 1) so that decode ms/token = (whole - prefill) / (new - 1).
    N > 1 = N independent replicas (replicas only: no collective on this path); rank 0 reports N x its own rate (weak scaling)."""
    from ultravox_amd import _lib
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.synthetic import synthetic_batch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.opt:
        for item in args.opt.split(","):
            k, v = item.split("=")
            _lib.lib().uvx_set_option(int(k), int(v))
    B, new = args.batch or wl["B"], wl["new_tokens"]
    free_gb = torch.cuda.mem_get_info()[0] / 2 ** 30
    need_gb = 170 if "70B" in wl["text"] else 40 if "8B" in wl["text"] else 6
    if free_gb < need_gb:          # never drive the box out of memory
        raise SystemExit(f"bench.py --workload {args.workload}: {free_gb:.0f} GiB free on the device, about {need_gb} GiB needed")
    cfg = UltravoxConfig(audio_model_id=wl["audio"], text_model_id=wl["text"], hidden_size=4096, stack_factor=8,
                         projector_ln_mid=True, torch_dtype="bfloat16")
    model = UltravoxModel(cfg, device=str(dev), dtype=torch.bfloat16, seed=0, rope_len=1024, with_backward=False, consume_state_dict=True)
    fe = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins, device=str(dev))
    batch = synthetic_batch(cfg, B, wl["seconds"], n_text=128, audio_start=16, n_supervised=32, rank=rank)
    pcm = batch.pop("pcm").to(dev)
    batch.pop("labels")
    gb = {k: v.to(dev) for k, v in batch.items()}
    T = gb["input_ids"].shape[1]

    def gen(n):
        mel = fe.logmel_device(pcm)                       # K1 on device, inside the step
        return model.generate(audio_values=mel, max_new_tokens=n, eos_token_id=-1, **gb)

    def timed(n, reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = gen(n)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, out

    for _ in range(max(1, args.warmup)):
        gen(new)
    if world > 1:
        torch.distributed.barrier()
    t_prefill, _ = timed(1, max(2, args.steps))
    if world > 1:
        torch.distributed.barrier()
    t_all, out = timed(new, args.steps)
    mine = torch.tensor([t_all], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(mine, op=torch.distributed.ReduceOp.MAX)
    t_all_max = float(mine.item())
    if rank != 0:
        return
    ms_tok = (t_all - t_prefill) / (new - 1) * 1e3
    wbytes = llm_weight_bytes(cfg)
    ach = wbytes / (ms_tok * 1e-3) / 1e9
    # the prefill against BOTH rooflines (round 5): algorithmic FLOPs of encoder + projector + the LLM body on all B x T rows + the LM
    # head on the last position of each prompt, and the bytes it cannot avoid - every LLM weight once (the towers' 0.6 GB ride along) -
    # over the WHOLE prefill time (log-mel, encoder, projector, merge, first-token argmax included)
    fl = flops_per_sample(cfg, wl["seconds"], 128, 1, top_rows=False)
    pf_flops = B * (fl["encoder"] + fl["projector"] + fl["llm_fwd"])
    pf_tf, pf_gb = pf_flops / t_prefill / 1e12, wbytes / t_prefill / 1e9
    print(json.dumps({
        "metric": "decoded tokens/sec, prefill + decode (Whisper-med + Llama-3.3-70B inference)" if args.workload == "c4" else "decoded tokens/sec, prefill + decode",
        "value": B * world * new / t_all_max, "unit": "tokens/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": t_all_max * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic (seeded PCM + token ids; seeded random-init weights)",
        "config": {"workload": wl["name"], "prompts_per_gpu": B, "clip_seconds": wl["seconds"], "text_tokens": 128, "prompt_len": T,
                   "new_tokens": new, "parallelism": f"{world} independent replica(s), TP=1 (replicas only: no collective on this path)",
                   "audio_model": wl["audio"], "text_model": wl["text"], "decoding": "greedy, KV cache, no early stop"},
        "prefill_ms": t_prefill * 1e3,
        "prefill": {"tflop": pf_flops / 1e12, "tflops": pf_tf, "frac_mfma": pf_tf / 2500.0, "gbps": pf_gb, "frac_hbm": pf_gb / PEAK_HBM_GBS,
                    "note": "whole time-to-first-token (log-mel + encoder + projector + LLM prefill + argmax) against the dense bf16 MFMA peak "
                            "(2.5 PF) and against streaming the LLM weights once at 8 TB/s; at 316 rows per prompt the two bounds are within 2x of each other"},
        "decode_ms_per_token": ms_tok, "decode_tokens_per_sec": B * world / (ms_tok * 1e-3),
        "resident_gib": torch.cuda.memory_allocated() / 2 ** 30,
        "roofline": {"bound": "hbm", "kernel": "decode step: gemv_rows_bf16_k / gemm_skinny_bf16_k weight streaming (all layers + LM head)",
                     "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                     "traffic": pmc_decode_traffic_per_token() if args.workload == "c4" and B == 1 else None,
                     "algorithmic_bytes_per_token": wbytes,
                     "note": "achieved = LLM weight bytes / WHOLE decode step time (attention, norms, RoPE, sampling included in the time, not in the bytes)"},
        "output_shape": list(out.shape)}))


def flops_per_sample(cfg, seconds: float, n_text: int = 128, n_supervised: int = 32, top_rows: bool = True, bwd_skip_rows: int = 0):
    """Algorithmic FLOPs of one sample: matmul [m,k]x[k,n] = 2mkn, causal attention = 1/2.  The LM head
    is counted on the supervised positions only (the rows that enter the loss; the other rows of the logits have zero
    weight and zero gradient, and the training step does not compute them), and so is the last LLM layer's o_proj + MLP -
    `step_full_head` keeps the all-rows count.  top_rows = False: the step flavours that DO run the last layer on every row
    (LLM LoRA, UVX_TOP_LAYER_ROWS=0, the f32 / full-logits KL path) - nothing is subtracted for them.  (Round 6: the compact-rows KL step
    - uvx_llm_fwd_rows / uvx_llm_bwd_rows - runs its last layer on the loss rows too, student and teacher.)"""
    a, t = cfg.audio_config, cfg.text_config
    F = int(seconds * 100)
    Te = F // 2
    Na = -(-F // 16)
    T = n_text + Na
    d, Le, ffn = a.d_model, a.encoder_layers, a.encoder_ffn_dim
    E = 2 * F * a.num_mel_bins * 3 * d + 2 * Te * d * 3 * d + Le * (8 * Te * d * d + 4 * Te * Te * d + 4 * Te * d * ffn)
    if getattr(a, "is_wav2vec2", False):      # conv stack + feature projection + grouped positional conv + post-LN layers
        n, Cc, E, cin = int(seconds * 16000), a.conv_dim[0], 0, 1
        for k, st in zip(a.conv_kernel, a.conv_stride):
            n = (n - k) // st + 1
            E += 2 * n * Cc * k * cin
            cin = Cc
        Te = n
        Na = -(-Te // cfg.stack_factor)
        T = n_text + Na
        E += 2 * Te * Cc * d + 2 * Te * d * a.num_conv_pos_embeddings * (d // a.num_conv_pos_embedding_groups)
        E += Le * (8 * Te * d * d + 4 * Te * Te * d + 4 * Te * d * ffn)
    H, D = cfg.hidden_size, t.hidden_size
    P = 2 * Na * (8 * d * H + (H // 2) * D)
    h, kv, dh, I, V, L = t.num_attention_heads, t.num_key_value_heads, t.head_dim, t.intermediate_size, t.vocab_size, t.num_hidden_layers
    attn = L * 2 * T * T * h * dh
    body = L * (2 * T * (2 * D * h * dh + 2 * D * kv * dh) + 6 * T * D * I) + attn
    head, head_full = 2 * n_supervised * D * V, 2 * T * D * V
    # the last layer's o_proj + MLP (row-wise, after the last position mixing) likewise run on the supervised rows only
    top_skip = (T - n_supervised) * (2 * D * h * dh + 6 * D * I) if top_rows else 0
    M = body + head - top_skip
    # bwd_skip_rows: positions per sample whose row-wise backward is not run (uvx_llm_bwd_train_from: the text before the first audio token) - the
    # q|k|v dgrad of every layer, the o_proj + MLP dgrads of every layer but a row-compacted last one; the attention backward still runs on all rows
    prefix_skip = bwd_skip_rows * (L * (2 * D * h * dh + 4 * D * kv * dh) + (L - (1 if top_rows else 0)) * (2 * D * h * dh + 6 * D * I))
    step = E + 3 * P + M + (M + attn) - prefix_skip
    # recipe flavours (meta_config.yaml:5-6): the KL teacher = a forward of the same LLM over the text-only alternative (n_alt tokens, logits on
    # the supervised rows), and - under encoder LoRA - the tower's backward: the dgrads of its layers' linears (= their forward FLOPs) and the
    # attention backward (5 products against the forward's 2); the adapters' own rank-r products are not counted
    n_alt = 16 + 48 + (n_text - 16)
    teacher = L * (2 * n_alt * (2 * D * h * dh + 2 * D * kv * dh) + 6 * n_alt * D * I) + L * 2 * n_alt * n_alt * h * dh + head
    if top_rows:
        teacher -= (n_alt - n_supervised) * (2 * D * h * dh + 6 * D * I)
    enc_bwd = Le * (8 * Te * d * d + 4 * Te * d * ffn + 10 * Te * Te * d)
    return dict(encoder=E, projector=P, llm_fwd=M, step=step, step_full_head=step + 2 * (head_full - head) + 2 * top_skip + prefix_skip, bwd_prefix_skip=prefix_skip,
                kl_teacher=teacher, encoder_lora_bwd=enc_bwd)


def cpu_baseline(cfg, seconds: float, n_text: int = 128):
    """Bounded sample of the SAME workload through the oracle on the host cores: B = 1 clip, the full-size
    log-mel + conv stem + 2 encoder layers, the projector, 1 LLM layer (fwd + bwd) and the lm_head + CE
    (fwd + bwd); per-layer times are scaled to the real layer counts."""
    from oracle import reference_cpu as O
    from ultravox_amd.config import UltravoxConfig
    import dataclasses

    # Thread count: the GPU boxes expose 256 hardware threads, but torch's CPU GEMM peaks far below that
    # (measured: 32 threads 1.7 TFLOP/s, 256 threads 0.3 TFLOP/s on a 4096^3 f32 matmul), so calibrate.
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    xa, xb = torch.randn(3072, 3072), torch.randn(3072, 3072)
    best_t, cores = 1e30, 1
    for th in sorted({min(avail, c) for c in (8, 16, 24, 32, 48, 64)}):
        torch.set_num_threads(th)
        xa @ xb
        dt_ = 1e30
        for _ in range(3):
            t0 = time.perf_counter(); xa @ xb
            dt_ = min(dt_, time.perf_counter() - t0)
        if dt_ < best_t:
            best_t, cores = dt_, th
    torch.set_num_threads(cores)
    a, t = cfg.audio_config, cfg.text_config
    small = UltravoxConfig(audio_config=dataclasses.replace(a, encoder_layers=2),
                           text_config=dataclasses.replace(t, num_hidden_layers=1),
                           hidden_size=cfg.hidden_size, stack_factor=cfg.stack_factor,
                           projector_ln_mid=cfg.projector_ln_mid)
    from ultravox_amd.weights import random_state_dict
    sd = random_state_dict(small, seed=0, dtype=torch.float32)
    om = O.OracleModel(small, sd, dtype=torch.float32)
    b = O.synthetic_batch(small, 1, seconds, n_text=n_text)
    pcm = b.pop("pcm")

    def timed(fn, reps=1):
        fn()  # warm-up (thread pool, allocator)
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        return (time.perf_counter() - t0) / reps, r

    t_mel, mel = timed(lambda: O.logmel_ref(pcm, a.num_mel_bins))
    with torch.no_grad():
        t_enc2, _ = timed(lambda: O.whisper_encoder_ref(om.sd, small, mel, b["audio_lens"]))
        small0 = UltravoxConfig(audio_config=dataclasses.replace(a, encoder_layers=0), text_config=small.text_config,
                                hidden_size=cfg.hidden_size, projector_ln_mid=cfg.projector_ln_mid)
        t_enc0, _ = timed(lambda: O.whisper_encoder_ref(om.sd, small0, mel, b["audio_lens"]))
    per_enc_layer = max(t_enc2 - t_enc0, 0.0) / 2
    batch = {**b, "audio_values": mel}

    def llm_step(n_layers):
        def f():
            for k in om.trainable:
                om.sd[k].grad = None
            _, audio_embeds = om.audio_embeds(mel, b["audio_lens"])   # encoder(2 layers) + projector
            emb = torch.nn.functional.embedding(b["input_ids"], om.sd["language_model.model.embed_tokens.weight"])
            emb = O.merge_ref(emb, audio_embeds, b["audio_token_start_idx"], b["audio_token_len"], b["audio_batch_size"])
            logits = O.llama_ref(om.sd, small, emb, b["attention_mask"], n_layers=n_layers)
            O.causal_lm_loss_ref(logits, b["labels"]).backward()
        return f

    t_l1, _ = timed(llm_step(1))
    t_l0, _ = timed(llm_step(0))
    per_llm_layer = max(t_l1 - t_l0, 0.0)
    rest = t_l0 - t_enc2  # projector fwd+bwd, merge, head + CE fwd+bwd
    step = t_mel + t_enc0 + per_enc_layer * a.encoder_layers + max(rest, 0.0) + per_llm_layer * t.num_hidden_layers
    return {
        "value": seconds / step, "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
        "sample": (f"oracle f32, B=1x{seconds:g}s clip: log-mel + conv stem + 2/{a.encoder_layers} encoder layers, projector, "
                   f"1/{t.num_hidden_layers} LLM layer fwd+bwd, lm_head+CE fwd+bwd; per-layer times scaled to full depth "
                   f"(extrapolated step {step:.1f} s)"),
    }


def hip_step_for_live_parity(cfg, sd, batch, pcm, dev):
    """The HIP side of the bench line's LIVE parity figure (`parity.live`): one B = 1 forward + backward of the production bf16 path at
    FULL depth on the weights and the batch the cpu_baseline leg is about to push through the f32 oracle (the oracle is the checker; this is
    the thing checked).  Returns host-side results; the model is freed before the oracle runs."""
    import gc
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    sd16 = {k: v.to(torch.bfloat16) for k, v in sd.items()}
    model = UltravoxModel(cfg, state_dict=sd16, device=str(dev), dtype=torch.bfloat16, rope_len=1024, consume_state_dict=True)
    del sd16
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(dev))
    gb = {k: v.to(dev) for k, v in batch.items()}
    out = model.forward(audio_values=mel, **gb)                      # full logits + loss
    logits = out.logits.float().cpu()
    model.train()
    loss = float(model.forward_backward(audio_values=mel, **gb).item())
    grads = {k: g.float().cpu() for k, g in model.projector_grads().items()}
    torch.cuda.synchronize()
    del model, out, gb, mel
    gc.collect()
    torch.cuda.empty_cache()
    return {"logits": logits, "loss": loss, "grads": grads}


def cpu_baseline_full(cfg, seconds: float, n_text: int = 128, steps: int = 1, parity_dev=None):
    """ONE WHOLE adapter-train step of the workload at B = 1 through the oracle on the host cores - all encoder and LLM layers,
    nothing extrapolated (the default run's `cpu_baseline` leg when the box has the memory; `python bench.py
    --cpu-baseline-full OUT.json` runs it alone).  Weights: one seeded random layer per tower, copied into DISTINCT memory for every
    layer (generating 8 G random f32 numbers would take longer than the step; the values do not matter for timing, the
    memory traffic does)."""
    from oracle import reference_cpu as O
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.weights import random_state_dict
    import dataclasses

    avail_kb = 0
    try:
        avail_kb = int(next(l for l in open("/proc/meminfo") if l.startswith("MemAvailable")).split()[1])
    except Exception:
        pass
    a, t = cfg.audio_config, cfg.text_config
    need_gb = (t.num_hidden_layers * (4 * t.hidden_size * t.hidden_size // 2 + 3 * t.hidden_size * t.intermediate_size) +
               2 * t.vocab_size * t.hidden_size) * 4 / 2 ** 30 * 1.5
    if avail_kb and avail_kb / 2 ** 20 < need_gb:
        raise MemoryError(f"{avail_kb / 2 ** 20:.0f} GB of host memory available, the f32 oracle needs about {need_gb:.0f} GB")
    cores = min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    one = UltravoxConfig(audio_config=dataclasses.replace(a, encoder_layers=1), text_config=dataclasses.replace(t, num_hidden_layers=1),
                         hidden_size=cfg.hidden_size, stack_factor=cfg.stack_factor, projector_ln_mid=cfg.projector_ln_mid)
    sd = random_state_dict(one, seed=0, dtype=torch.float32)
    for i in range(1, a.encoder_layers):
        for k in [k for k in sd if k.startswith("audio_tower.layers.0.")]:
            sd[k.replace("layers.0.", f"layers.{i}.", 1)] = sd[k].clone()
    for i in range(1, t.num_hidden_layers):
        for k in [k for k in sd if k.startswith("language_model.model.layers.0.")]:
            sd[k.replace("layers.0.", f"layers.{i}.", 1)] = sd[k].clone()
    b = O.synthetic_batch(cfg, 1, seconds, n_text=n_text)
    pcm = b.pop("pcm")
    hip = None
    if parity_dev is not None:
        # live parity (round 6): the weights take bf16-representable values (timing is indifferent to the values) so that the HIP path - run
        # FIRST, then freed - and the f32 oracle see the same numbers
        for k in sd:
            sd[k] = sd[k].to(torch.bfloat16).float()
        try:
            hip = hip_step_for_live_parity(cfg, sd, b, pcm, parity_dev)
        except Exception as e:          # the CPU leg does not depend on it
            hip = {"error": f"{type(e).__name__}: {e}"}
    om = O.OracleModel(cfg, sd, dtype=torch.float32)
    del sd
    ref = {}

    def step():
        mel = O.logmel_ref(pcm, a.num_mel_bins)
        out, grads, _ = om.train_step({**b, "audio_values": mel})
        if hip is not None and "error" not in hip and not ref:
            ref.update(logits=out["logits"].detach().float(), grads={k: g.float() for k, g in grads.items()})
        return float(out["loss"].detach())

    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        loss = step()
        times.append(time.perf_counter() - t0)
    best = min(times)
    del om
    rec = {"value": seconds / best, "unit": "audio-seconds/sec", "cores": cores, "kind": "port", "step_seconds": times,
           "sample": f"oracle f32, ONE WHOLE step at B=1x{seconds:g}s: log-mel, {a.encoder_layers} encoder layers, projector, "
                     f"{t.num_hidden_layers} LLM layers + lm_head + CE forward and backward (nothing extrapolated; loss {loss:.3f})"}
    if hip is not None:
        if "error" in hip:
            rec["_live_parity"] = {"error": hip["error"]}
        else:
            rl2 = lambda x, y: float((x - y).norm() / (y.norm() + 1e-30))
            rec["_live_parity"] = {
                "measured": "in THIS run: production bf16 HIP path vs the f32 oracle of the cpu_baseline leg, same weights (bf16-representable), same B = 1 batch, FULL depth",
                "logits_rel_l2_vs_f32_oracle": rl2(hip["logits"], ref["logits"].reshape(hip["logits"].shape)),
                "logits_max_abs_diff": float((hip["logits"] - ref["logits"].reshape(hip["logits"].shape)).abs().max()),
                "loss": {"hip": hip["loss"], "f32_oracle": loss},
                "projector_grads_rel_l2_vs_f32_oracle": {k.split("multi_modal_projector.")[-1]: rl2(hip["grads"][k], g) for k, g in ref["grads"].items() if k in hip["grads"]}}
    return rec


def pmc_traffic_per_launch(profiles_dir=None):
    """roofline.traffic: memory-side bytes per GEMM launch from the newest committed PMC summary (profiles/rNN_pmc_traffic.json,
    written by tools/pmc_traffic.sh from separate FETCH_SIZE / WRITE_SIZE passes over this same command), or None."""
    d = profiles_dir or os.path.join(ROOT, "profiles")
    try:
        names = sorted(n for n in os.listdir(d) if n.startswith("r") and n.endswith("_pmc_traffic.json"))
        return json.load(open(os.path.join(d, names[-1]))).get("traffic_bytes_per_launch") if names else None
    except (OSError, ValueError):
        return None


def pmc_gemm_counter_per_launch(csv_path: str):
    """(launches, mean Counter_Value per launch) over the gemm_nt_bf16_* rows of a rocprofv3 counter_collection.csv"""
    import csv
    n, total = 0, 0.0
    with open(csv_path) as f:
        for r in csv.DictReader(f):
            if "gemm_nt" in r.get("Kernel_Name", ""):
                n += 1
                total += float(r["Counter_Value"])
    return n, (total / n if n else None)


def measure_traffic_live(child_args, timeout_s: int = 90):
    """roofline.traffic MEASURED IN THIS RUN (round 5; before, the line quoted the committed profiles/rNN_pmc_traffic.json and a traffic regression
    would have gone unseen): two short sub-runs of this same command under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate
    passes: the two derived counters do not fit the TCC's counters together; 2 timed + 1 warm-up step each, no CPU leg, no HIP-event timing), summed
    over the gemm_nt_bf16_* dispatches, FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 bytes: MI355X_MICROARCH.md).  Memory-side (fabric)
    counters: L2-miss traffic, Infinity-Cache hits included.  Returns (bytes per launch, detail dict) or (None, reason): never raises."""
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not rp:
        return None, "rocprofv3 not found"
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="uvx_pmc_", dir="/tmp")
            try:
                cmd = [rp, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable,
                       os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-prof", "--no-live-traffic", *child_args]
                # the sub-run is rocprofv3 -> python: its own session / process group, so that a timeout ends BOTH (killing only
                # rocprofv3 would leave the grandchild python running on the GPU next to the parent - ADVICE r5)
                proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                        start_new_session=True)
                try:
                    rc = proc.wait(timeout=timeout_s)
                except subprocess.TimeoutExpired:
                    import signal
                    try:
                        os.killpg(proc.pid, signal.SIGKILL)
                    except ProcessLookupError:
                        pass
                    proc.wait()
                    return None, f"rocprofv3 --pmc {counter} sub-run timed out after {timeout_s} s (its process group was killed)"
                files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
                if rc != 0 or not files:
                    return None, f"rocprofv3 --pmc {counter} sub-run failed (rc {rc})"
                n, per = pmc_gemm_counter_per_launch(files[0])
                if not n:
                    return None, f"no gemm_nt dispatch in the {counter} pass"
                got[counter] = (n, per)
            finally:
                shutil.rmtree(d, ignore_errors=True)
    except Exception as e:  # a missing tool, a timeout, an unreadable CSV: the committed figure is quoted instead, and the line says so
        return None, f"{type(e).__name__}: {e}"
    fetch_kb, write_kb = got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]
    return (2.0 * fetch_kb + write_kb) * 1024.0, {"launches_per_pass": got["FETCH_SIZE"][0], "fetch_kb_per_launch_raw": fetch_kb,
                                                   "write_kb_per_launch": write_kb, "gfx950_fetch_correction": 2.0}


def _f32_mode_record(path: str):
    """north_star's "logits within 1e-3": the f32 compute mode (dtype UVX_F32, not the benchmarked path) at the C2 width, depth 2"""
    try:
        r = json.load(open(path))
        return {"max_abs_logit_diff_vs_f32_oracle": r["max_abs_logit_diff"], "logit_rms": r["logit_rms"], "bar": 1e-3,
                "config": "C2 width (Llama-3-8B 4096 / 14336 / 128256, whisper-medium), depth 2, 2 x 30 s; f32 kernels, not the benchmarked bf16 path"}
    except (OSError, ValueError, KeyError):
        return "max |logit diff| < 1e-3 vs the f32 oracle asserted by tests/test_c2_full_depth_gpu.py (f32 kernels; not the benchmarked path)"


def parity_record(workload: str, profiles_dir=None):
    """The bench line's `parity` object: what the committed full-depth parity run of THIS workload measured (profiles/rNN_parity/
    <workload>_full_depth.json, written by tests/test_c2_full_depth_gpu.py on the GPU box) - the production bf16 path against the
    f32 oracle, next to torch-ROCm's own bf16 distance to the same oracle.  north_star's 1e-3 on the logits is a statement about
    f32 arithmetic (met by the f32 compute mode, tests/test_c2_full_depth_gpu.py::test_c2_width_f32_mode_logits_within_1e3); a path
    that stores activations in bf16 (2^-8 per stored value) cannot meet it, and the line says so instead of leaving it implicit."""
    d = profiles_dir or os.path.join(ROOT, "profiles")
    try:
        dirs = sorted(n for n in os.listdir(d) if n.startswith("r") and n.endswith("_parity"))
        for n in reversed(dirs):
            f = os.path.join(d, n, f"{workload}_full_depth.json")
            if os.path.exists(f):
                r = json.load(open(f))
                c = r.get("calibration", {})
                if "stages" not in r:
                    continue                 # (records of other kinds of run share the name pattern: c4's generate record)
                return {"source": f"profiles/{n}/{workload}_full_depth.json (full depth, 1 clip; checker: oracle/reference_cpu.py in f32)",
                        "dtype": "bf16 production path",
                        "logits_rel_l2_vs_f32_oracle": r["stages"]["logits"]["rel_l2"],
                        "logits_max_abs_diff": r["stages"]["logits"]["max_abs"], "logits_rms": r["stages"]["logits"]["ref_rms"],
                        "torch_rocm_bf16_logits_rel_l2_vs_f32_oracle": c.get("logits", {}).get("torch_bf16_vs_f32"),
                        "loss": {"hip": r.get("loss_hip_train_step"), "f32_oracle": r.get("loss_oracle")},
                        "projector_grads_rel_l2_vs_f32_oracle_max": max(r.get("grads_rel_l2", {"": None}).values()),
                        "audio_token_placement": "bit-exact (integer path)",
                        "f32_mode": _f32_mode_record(os.path.join(d, n, "c2_width_f32_mode.json"))}
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return None


def _opt_get(opt: str, key: int, default: int) -> int:
    """value of `key` in a --opt 'k=v,...' string (the library's default otherwise)"""
    for item in (opt or "").split(","):
        if "=" in item and int(item.split("=")[0]) == key:
            return int(item.split("=")[1])
    return default


def self_launch(n: int) -> int:
    """Re-execute this command line as n ranks of one node: `python -m torch.distributed.run --nnodes=1 --nproc-per-node n
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5,
                    help="untimed steps first; fewer than ~4 leave the clock / power ramp of a just-initialised GPU inside the timed region "
                         "(5 timed + 2 warm-up steps read 10-25 %% slow, profiles/r04_bench_short_run_bias.txt)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--dgrad-nn", action="store_true",
                    help="no transposed weight copies at all: the frozen LLM's dgrads read the forward weights through the GEMM's NN form "
                         "(bit-identical results; -14 GB at Llama-3-8B; A/B against the resident / streamed copies)")
    ap.add_argument("--stream-wt", default="auto", choices=["auto", "on", "off"],
                    help="transposed weight copies of the frozen LLM for the backward pass: made on the fly on a side stream (on), resident (off), "
                         "or resident unless the LLM exceeds a third of the GPU's memory (auto, the model's default)")
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", default=None, metavar="OUT.json",
                    help="CPU only: time ONE WHOLE step of the workload at B = 1 through the oracle (all layers) and write the record")
    ap.add_argument("--no-prof", action="store_true", help="skip the live HIP-event timing of the GEMM kernel")
    ap.add_argument("--parity-live", action="store_true",
                    help="with the cpu_baseline leg: also run its weights and B = 1 batch through the HIP path at full depth and report the measured "
                         "distances as parity.live (default on for the plain c2 line; adds ~10 s)")
    ap.add_argument("--no-parity-live", action="store_true", help="never run the live parity check")
    ap.add_argument("--no-prefix-skip", action="store_true",
                    help="A/B: run the LLM backward on every row (uvx_llm_bwd_train) instead of from the first audio token (uvx_llm_bwd_train_from)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="quote roofline.traffic from the committed PMC summary instead of measuring it in two rocprofv3 --pmc sub-runs (~15 s each)")
    ap.add_argument("--audio-lora-r", type=int, default=0,
                    help="train rank-r LoRA on the encoder's q_proj/k_proj too (the reference's release recipe, "
                         "audio_model_lora_config.r = 8); 0 = frozen towers, the BASELINE.json configuration")
    ap.add_argument("--loss", default="ce", choices=["ce", "kl"],
                    help="ce = the BASELINE.json configuration; kl = LossFunction.KL_Divergence (the reference's meta_config.yaml "
                         "default): a text-only teacher pass over alt_input_ids (audio replaced by a 48-token transcript)")
    ap.add_argument("--no-kl-side-stream", action="store_true",
                    help="--loss kl: keep the text-only teacher pass on the step's own stream (default: a side stream next to the encoder and the "
                         "student forward - same kernels, identical results; A/B switch)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: sequential all-reduce + optimizer step (no overlap)")
    ap.add_argument("--comm", default="torch", choices=["torch", "abi"],
                    help="N > 1: gradient exchange through torch.distributed (RCCL; default) or through libuvx.so's own RCCL "
                         "communicator (uvx_comm_*: the C-ABI route, torch.distributed only hands the 128-byte id around)")
    ap.add_argument("--prof-every", type=int, default=4,
                    help="HIP events (roofline) on the GEMM launches of every N-th timed step (1 = every step); the others run un-instrumented")
    ap.add_argument("--opt", default=None, help="probe: 'key=value,...' for uvx_set_option")
    ap.add_argument("--gemm-override", default=None, help="probe: 'MxNxK=variant,...' tile-variant overrides")
    ap.add_argument("--gemm-table", default=None, help="write a per-shape GEMM time table (from the HIP events) here")
    ap.add_argument("--gemm-raw", default=None, help="probe: write every GEMM launch of the timed region, in issue order (M N K variant us)")
    ap.add_argument("--attn-qt", type=int, default=0, help="probe: uvx_attention_force_qt (query tiles per wave of the attention forward)")
    ap.add_argument("--autotune", action="store_true",
                    help="let the trainer pick the LLM schedule ({one chain, fused attention backward} or {two chains, kernel pair}) from "
                         "timed warm-up steps instead of keeping the library default (one chain, fused); on the boxes measured so far "
                         "the default is within 0.3 %% of the better one or ahead by up to 4 %%")
    ap.add_argument("--probe-skip", type=int, default=0,
                    help="TIMING PROBE: uvx_set_option(15, mask) after the warm-up - the masked kernel classes are not launched in the "
                         "timed steps (garbage results; the line is marked invalid): what a class costs inside the overlapped schedule")
    args = ap.parse_args()

    if args.cpu_baseline_full:
        from ultravox_amd.config import UltravoxConfig
        wl = WORKLOADS[args.workload]
        cfg = UltravoxConfig(audio_model_id=wl["audio"], text_model_id=wl["text"], hidden_size=4096, stack_factor=8, projector_ln_mid=True)
        rec = cpu_baseline_full(cfg, wl["seconds"])
        rec["workload"] = args.workload
        with open(args.cpu_baseline_full, "w") as f:
            json.dump(rec, f, indent=1)
        print(json.dumps(rec))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` (how the driver may invoke it): become the launcher - one rank per GPU under
        # torch.distributed.run on 127.0.0.1, as the reference is started by torchrun (train.py:126-130, README.md:144-148).
        # Rank 0 prints the JSON line; the launcher only forwards the exit code.
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (torch.distributed.run --nproc-per-node must equal --gpus)")
    # test hook for the multi-rank code path on a 1-GPU box: UVX_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo
    # (RCCL refuses two ranks on one device); never set by the driver, and the JSON line says so if it is
    share_gpu = os.environ.get("UVX_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    if WORKLOADS[args.workload].get("inference"):
        run_inference(args, WORKLOADS[args.workload], dev)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    from ultravox_amd import _lib
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel, UltravoxTrainer
    from ultravox_amd.synthetic import synthetic_batch

    if args.opt:
        for item in args.opt.split(","):
            k, v = item.split("=")
            _lib.lib().uvx_set_option(int(k), int(v))
    if args.attn_qt:
        _lib.lib().uvx_attention_force_qt(args.attn_qt)
    if args.gemm_override:
        for item in args.gemm_override.split(","):
            shp, v = item.split("=")
            m, n, k = (int(x) for x in shp.split("x"))
            _lib.lib().uvx_gemm_override_variant(m, n, k, int(v))
    wl = WORKLOADS[args.workload]
    B = args.batch or wl["B"]
    cfg = UltravoxConfig(audio_model_id=wl["audio"], text_model_id=wl["text"], hidden_size=4096, stack_factor=8,
                         projector_ln_mid=True, torch_dtype="bfloat16",
                         audio_model_lora_config={"r": args.audio_lora_r} if args.audio_lora_r else None)
    model = UltravoxModel(cfg, device=str(dev), dtype=torch.bfloat16, seed=0, rope_len=1024,
                          stream_weight_transposes={"auto": None, "on": True, "off": False}[args.stream_wt], dgrad_nn=args.dgrad_nn)
    model.skip_prefix_backward = not args.no_prefix_skip
    comm = None
    if args.comm == "abi" and world > 1 and not share_gpu:
        from ultravox_amd.parallel import UvxComm
        comm = UvxComm.from_torch_distributed()
    trainer = UltravoxTrainer(model, lr=2e-3, max_grad_norm=1.0, overlap_comm=world > 1 and not args.no_overlap, comm=comm)
    fe = None if cfg.audio_config.is_wav2vec2 else WhisperFeatureExtractor(cfg.audio_config.num_mel_bins, device=str(dev))
    batch = synthetic_batch(cfg, B, wl["seconds"], n_text=128, audio_start=16, n_supervised=32, rank=rank)
    pcm = batch.pop("pcm").to(dev)
    if args.loss == "kl":
        from ultravox_amd.config import LossConfig, LossFunction
        model.set_loss_config(LossConfig(loss_function=LossFunction.KL_Divergence))
        model.kl_teacher_side_stream = not args.no_kl_side_stream
        ids, Na = batch["input_ids"], int(batch["audio_token_len"][0])
        g = torch.Generator().manual_seed(777 + rank)
        tr = torch.randint(0, cfg.text_config.vocab_size - 1, (B, 48), generator=g)
        alt = torch.cat([ids[:, :16], tr, ids[:, 16 + Na:]], 1)
        alt_labels = alt.clone()
        alt_labels[:, : alt.shape[1] - 32] = -100
        batch.update(alt_input_ids=alt, alt_attention_mask=torch.ones_like(alt), alt_labels=alt_labels)
    batch = {k: v.to(dev) for k, v in batch.items()}
    T = batch["input_ids"].shape[1]

    if cfg.audio_config.is_wav2vec2:
        # raw-waveform tower: the feature extractor's per-clip normalisation (host arithmetic in the reference's data loader)
        # is applied once, outside the timed region; the step starts from input_values resident in HBM
        values = ((pcm - pcm.mean(-1, keepdim=True)) / torch.sqrt(pcm.var(-1, unbiased=False, keepdim=True) + 1e-7)).contiguous()

    def step():
        if cfg.audio_config.is_wav2vec2:
            return trainer.train_step(audio_values=values, **batch)
        mel = fe.logmel_device(pcm)                        # K1 on device, inside the step
        return trainer.train_step(audio_values=mel, **batch)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # the warm-up steps double as the schedule tuner's trial steps (1 throw-away + 2 per candidate when there are >= 5; with fewer
    # the tuner simply finishes during the first timed steps - it only reads event timers)
    tune = args.autotune and _opt_get(args.opt, 11, -1) < 0 and _opt_get(args.opt, 13, -1) < 0 and B >= 2 and not args.audio_lora_r
    if tune:
        trainer.autotune_schedule(rounds=2 if args.warmup >= 5 else 1)
    for _ in range(args.warmup):
        loss = step()
    trainer.flush()
    trainer.measure_comm = world > 1 and trainer.overlap_comm
    if args.probe_skip:      # after the warm-up: the skipped kernels' outputs then hold realistic (stale) data, not zeros
        _lib.lib().uvx_set_option(15, args.probe_skip)
    barrier()
    prof = (C.c_double * 12)()
    if not args.no_prof:
        _lib.lib().uvx_prof_begin()
    t0 = time.perf_counter()
    profiled_steps = 0
    for i in range(args.steps):
        if not args.no_prof:   # the GEMM launches of every N-th timed step carry HIP events (a timed dispatch costs ~1.4 us: profiles/r03_prof_event_overhead.txt)
            on = i % max(1, args.prof_every) == 0
            _lib.lib().uvx_prof_enable(int(on))
            profiled_steps += int(on)
        loss = step()
    trainer.flush()            # the last step's deferred all-reduce + optimizer step belong to the timed region
    barrier()
    dt = time.perf_counter() - t0
    shapes = None
    if not args.no_prof:
        if (args.gemm_table or args.gemm_raw) and rank == 0:
            buf = (C.c_double * (6 * 20000))()
            n = _lib.lib().uvx_prof_records(buf, 20000)
            if args.gemm_raw:
                with open(args.gemm_raw, "w") as f:
                    for i in range(n):
                        f.write("%d %d %d %d %d %.2f\n" % (buf[i * 6], buf[i * 6 + 1], buf[i * 6 + 2], buf[i * 6 + 3], buf[i * 6 + 4], buf[i * 6 + 5] * 1e3))
            agg = {}
            for i in range(n):
                key = tuple(int(buf[i * 6 + j]) for j in range(5))
                a = agg.setdefault(key, [0, 0.0])
                a[0] += 1; a[1] += buf[i * 6 + 5]
            shapes = sorted(((k, c, ms) for k, (c, ms) in agg.items()), key=lambda t: -t[2])
        gemm_union_ms = float(_lib.lib().uvx_prof_union_ms(0))      # wall time with >= 1 GEMM executing (= the sum on one stream)
        _lib.check(_lib.lib().uvx_prof_end(prof, 3), "uvx_prof_end")
    loss_val = float(loss.item())
    # per-rank diagnostics for the scaling runs: every rank's own wall time over the timed region and the time its compute
    # stream waited for the (overlapped) gradient all-reduce - so that a poor 1 -> N line can be read (a slow rank? an exposed
    # collective?) and not just observed
    mine = torch.tensor([dt, trainer.comm_exposed_ms() if trainer.measure_comm else 0.0], device=dev, dtype=torch.float64)
    per_rank = [mine]
    if world > 1:
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(per_rank, mine)
    rank_s = [float(t[0].item()) for t in per_rank]
    exposed_ms = [float(t[1].item()) / args.steps for t in per_rank]
    dt = max(rank_s)                                        # the contract: MAX over ranks

    if rank == 0:
        # rows per clip whose backward the step really skipped (uvx_llm_bwd_train_from): read off the last d inputs_embeds, whose skipped rows are zeros
        s16 = 16      # synthetic_batch(audio_start = 16), already a multiple of the 16-row tiles
        d_last = model.__dict__.get("_last_d_embeds")
        bwd_skip = s16 if (d_last is not None and model.skip_prefix_backward and float(d_last[:, :s16].abs().max()) == 0.0) else 0
        fl = flops_per_sample(cfg, wl["seconds"], top_rows=bool(model._llm_top_rows), bwd_skip_rows=bwd_skip)    # what the measured step really skipped
        flavour_extra = (fl["kl_teacher"] if args.loss == "kl" else 0) + (fl["encoder_lora_bwd"] if args.audio_lora_r else 0)
        fl["step"] += flavour_extra          # (the recipe flavours do more work per step than the CE line: count it)
        fl["step_full_head"] += flavour_extra
        audio_s = B * world * wl["seconds"] * args.steps
        ms = dt / args.steps * 1e3
        out = {
            "metric": "audio-seconds/sec/node adapter-train step (Whisper-med + Llama-3-8B)" if args.workload == "c2"
                      else "audio-seconds/sec/node adapter-train step",
            "value": audio_s / dt, "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (seeded PCM + token ids; seeded random-init weights)",
            "config": {"workload": wl["name"], "clips_per_gpu": B, "clip_seconds": wl["seconds"], "text_tokens": 128,
                       "seq_len": T, "global_batch": B * world, "parallelism": ("SHARED-GPU TEST MODE " if share_gpu else "") + f"dp{world}" + (" (all-reduce overlapped with the next step's frozen encoder)" if trainer.overlap_comm else ""),
                       "audio_model": wl["audio"], "text_model": wl["text"], "optimizer": "AdamW bf16 state, clip 1.0",
                       "loss": "cross-entropy" if args.loss == "ce" else "KL distillation (teacher: text-only pass of the same LLM over 176 tokens)",
                       "trainable": "projector" + (f" + encoder LoRA r={args.audio_lora_r} (q_proj, k_proj)" if args.audio_lora_r else ""),
                       "supervised_tokens_per_clip": 32,
                       "loss_head": "LM head + CE on the supervised positions only (identical loss and gradients)",
                       "llm_backward_from_position": bwd_skip},
            "samples_per_sec": B * world * args.steps / dt,
            "step_tflops_algorithmic": fl["step"] * B / 1e12,
            "step_tflops_with_full_logits": fl["step_full_head"] * B / 1e12,
            "mfu": fl["step"] * B * world * args.steps / dt / (PEAK_BF16_TFLOPS * 1e12 * world),
            "loss": loss_val,
            **({"INVALID_timing_probe_skipped_kernel_mask": args.probe_skip} if args.probe_skip else {}),
            "world_size": world,
            "per_rank_ms": {"min": min(rank_s) / args.steps * 1e3, "max": max(rank_s) / args.steps * 1e3,
                            "all": [round(t / args.steps * 1e3, 3) for t in rank_s]},
            "allreduce_exposed_ms_per_step": (None if world == 1 else
                                              {"max": max(exposed_ms), "all": [round(x, 4) for x in exposed_ms],
                                               "note": "time the compute stream waited for the deferred all-reduce" if trainer.overlap_comm
                                                       else "sequential schedule: the collective is on the compute stream, not timed separately"}),
            "llm_schedule": ({"chosen": {str(k): v for k, v in trainer.llm_schedule.items()}, "trial_ms": {k: round(v, 3) for k, v in trainer.schedule_timings.items()},
                              "note": "uvx_set_option keys: 11 = LLM layer chains (streams), 13 = fused attention backward; picked "
                                      "from timed warm-up steps, results are bit-identical within a chain count"}
                             if tune and trainer.llm_schedule else
                             {"chosen": {"11": _opt_get(args.opt, 11, 1), "13": _opt_get(args.opt, 13, 1)}, "trial_ms": None}),
            "collective": (None if world == 1 else "gloo (shared-GPU test mode)" if share_gpu
                           else "RCCL " + ".".join(str(x) for x in torch.cuda.nccl.version()) + (" via uvx_comm_* (C ABI)" if comm else " via torch.distributed")
                                + " all-reduce(sum) of one flat f32 bucket, "
                                f"{trainer.model.proj_grad.numel() * 4 / 1e6:.0f} MB"),
        }
        out["parity"] = parity_record(args.workload)
        if not args.no_prof and prof[0] > 0:
            # denominator: the union of the GEMM launches' event intervals.  On one stream that is the sum of their durations;
            # with the two-stream LLM schedule launches of the two chains overlap (each interval then includes time the
            # kernel shared the chip), and the union is the time the GEMM family as a whole had the matrix cores.
            gemm_ms = gemm_union_ms if gemm_union_ms > 0 else prof[1]
            ach = prof[2] / (gemm_ms * 1e-3) / 1e12
            # memory-side traffic per GEMM launch from the PMC passes of the SAME command (tools/pmc_traffic.sh,
            # separate FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE x2 on gfx950), committed under profiles/
            traffic, traffic_source, traffic_detail = None, None, None
            if args.workload == "c2":
                plain = args.loss == "ce" and not args.audio_lora_r and not args.opt and not args.gemm_override and world == 1
                if plain and not args.no_live_traffic and not args.no_cpu_baseline:      # (the probes' A/B arms pass --no-cpu-baseline: no sub-runs there)
                    torch.cuda.synchronize()
                    torch.cuda.empty_cache()          # the sub-runs build their own model next to this process's: hand back every cached block first
                    traffic, traffic_detail = measure_traffic_live(["--workload", "c2"])
                    traffic_source = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE sub-runs of this command "
                                      "(2 + 1 steps each), gemm_nt_bf16_* dispatches, FETCH_SIZE x 2 (gfx950)") if traffic else None
                if traffic is None:
                    why = traffic_detail if isinstance(traffic_detail, str) else None
                    traffic, traffic_detail = pmc_traffic_per_launch(), None
                    traffic_source = "profiles/rNN_pmc_traffic.json (newest committed PMC summary of this command)" + (f"; live measurement skipped: {why}" if why else "")
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_nt_bf16_* (all tile variants)", "achieved": ach, "peak": PEAK_BF16_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                               "traffic_detail": traffic_detail,
                               "traffic_note": "memory-side (L2-miss) bytes per GEMM launch, Infinity-Cache hits included: an upper bound on HBM bytes",
                               "algorithmic_bytes_per_launch": prof[3] / prof[0],
                               "launches_per_step": prof[0] / profiled_steps, "gemm_ms_per_step": gemm_ms / profiled_steps,
                               "gemm_ms_per_step_summed_intervals": prof[1] / profiled_steps,
                               "profiled_steps": f"{profiled_steps} of the {args.steps} timed steps (every {max(1, args.prof_every)}th; HIP events attached to every GEMM dispatch of those steps)",
                               "avg_launch_us": gemm_ms / prof[0] * 1e3,
                               "algorithmic_gflop_per_launch": prof[2] / prof[0] / 1e9}
        if world == 1 and not args.no_cpu_baseline:
            # The oracle on THIS box's host cores.  Preferred: ONE WHOLE B = 1 step (all layers, nothing extrapolated; ~15 s of
            # CPU work + ~15 s of weight set-up) - needs ~45 GB of host memory for the f32 weights of an 8B-parameter LLM.
            # Otherwise the bounded sample (2 encoder layers + 1 LLM layer timed, scaled to full depth).
            try:
                # (with --parity-live, or by default on the BASELINE workload: free this process's model and let the leg also push its
                #  weights and batch through the HIP path - the `parity.live` object is then measured in this very run)
                live = (args.parity_live or (args.workload == "c2" and args.loss == "ce" and not args.audio_lora_r)) and not args.no_parity_live
                if live:
                    import gc
                    del trainer, model
                    gc.collect()
                    torch.cuda.synchronize()
                    torch.cuda.empty_cache()
                out["cpu_baseline"] = cpu_baseline_full(cfg, wl["seconds"], parity_dev=dev if live else None)
                lp = out["cpu_baseline"].pop("_live_parity", None)
                if lp is not None:
                    out.setdefault("parity", {})
                    if out["parity"] is None:
                        out["parity"] = {}
                    out["parity"]["live"] = lp
            except MemoryError as e:
                try:
                    out["cpu_baseline"] = cpu_baseline(cfg, wl["seconds"])
                    out["cpu_baseline"]["note"] = f"whole-step measurement skipped ({e})"
                except Exception as e2:  # the GPU number stands on its own; report why the CPU leg is missing
                    out["cpu_baseline"] = {"value": None, "error": f"{type(e2).__name__}: {e2}"}
            except Exception as e:
                out["cpu_baseline"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        if shapes:
            with open(args.gemm_table, "w") as f:
                f.write("# per-shape bf16 GEMM time inside the timed region (HIP events), C2 step\n")
                f.write(f"# {'M':>6s} {'N':>7s} {'K':>7s} {'batch':>5s} {'var':>3s} {'calls/step':>10s} {'ms/step':>9s} {'avg_us':>9s} {'TF/s':>8s}\n")
                for (m, n, k, bt, v), c, ms in shapes:
                    f.write(f"  {m:6d} {n:7d} {k:7d} {bt:5d} {v:3d} {c / profiled_steps:10.1f} {ms / profiled_steps:9.3f} "
                            f"{ms / c * 1e3:9.1f} {2.0 * m * n * k * bt * c / ms / 1e9:8.1f}\n")
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
