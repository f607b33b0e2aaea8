"""Synthetic inputs of the benchmark / smoke configurations: what one rank feeds one
adapter-training step when no dataset is reachable.  PCM = 0.1 * N(0,1) clipped to [-1, 1]
(seed 1234 + rank), 16 kHz; `n_text` token ids uniform in [0, V-2] (seed 4321 + rank) with the audio
placeholder run (the EOS id, ultravox_processing.py:338-352) inserted at `audio_start`; attention mask all
ones; labels = ids with everything but the last `n_supervised` tokens set to -100 (LAST_ASSISTANT
masking, ultravox_data_proc.py:106-110)."""
from __future__ import annotations

import torch

HOP = 160


def synthetic_batch(cfg, B: int, seconds: float, n_text: int = 128, audio_start: int = 16, n_supervised: int = 32,
                    rank: int = 0):
    g = torch.Generator().manual_seed(1234 + rank)
    L = int(round(seconds * 16000)) // HOP * HOP
    pcm = (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1)
    frames = L // HOP
    n_audio = -(-frames // (2 * cfg.stack_factor))  # ceil(frames / (encoder_ds_factor * stack_factor))
    if getattr(cfg.audio_config, "is_wav2vec2", False):     # raw-waveform tower: audio_lens = encoder frames, no 2x factor
        frames = cfg.audio_config.feat_extract_output_length(L)
        n_audio = -(-frames // cfg.stack_factor)
    g2 = torch.Generator().manual_seed(4321 + rank)
    V = cfg.text_config.vocab_size
    text = torch.randint(0, V - 1, (B, n_text), generator=g2)
    eos = cfg.text_config.eos_token_id
    ids = torch.cat([text[:, :audio_start], torch.full((B, n_audio), eos, dtype=torch.long), text[:, audio_start:]], 1)
    T = ids.shape[1]
    labels = ids.clone()
    labels[:, : T - n_supervised] = -100
    return {
        "pcm": pcm, "input_ids": ids, "attention_mask": torch.ones(B, T, dtype=torch.long), "labels": labels,
        "audio_token_start_idx": torch.full((B,), audio_start, dtype=torch.long),
        "audio_lens": torch.full((B,), frames, dtype=torch.long),
        "audio_token_len": torch.full((B,), n_audio, dtype=torch.int32),
        "audio_batch_size": torch.ones(B, dtype=torch.long),
    }
