"""Seeded weights / inputs shared by tests/golden/make_golden.py `forward_cases` (which fills the REFERENCE UltravoxModel with
them and records its outputs in tests/golden/forward_reference.npz) and the tests that replay the same case through the
oracle and the HIP path.  Every tensor is a function of its NAME and shape only (torch CPU randn with a crc32 seed), so the
fixture stores outputs, not megabytes of weights.  Dimensions are the smallest the HIP kernels take (head_dim 64, GEMM K % 64)."""
import zlib

import torch

TEXT = {"model_type": "llama", "hidden_size": 256, "intermediate_size": 512, "num_hidden_layers": 2, "num_attention_heads": 4,
        "num_key_value_heads": 2, "vocab_size": 512, "rms_norm_eps": 1e-5, "max_position_embeddings": 256, "rope_theta": 10000.0}
AUDIO = {"model_type": "whisper", "d_model": 64, "encoder_layers": 2, "encoder_attention_heads": 2, "encoder_ffn_dim": 128,
         "num_mel_bins": 80, "max_source_positions": 1500}
N_AUDIO, B, T = 4, 3, 44
# The REAL-tower fixtures (tests/golden/real_tower_reference.npz: the reference's own ModifiedWhisperEncoder.forward) use an
# encoder the HIP kernels accept (head_dim 64); everything else as above.
AUDIO_REAL = {"model_type": "whisper", "d_model": 128, "encoder_layers": 2, "encoder_attention_heads": 2, "encoder_ffn_dim": 256,
              "num_mel_bins": 80, "max_source_positions": 1500}
ENCODER_CASES = {            # name -> (mel frames, audio_len or None, audio_latency_block_size or None)
    "key_padding": (400, [400, 123, 250], None),
    "latency_and_padding": (400, [400, 123, 250], 50),
    "latency_only": (300, None, 100),
    "odd_frames": (77, [77, 40, 1], None),
}


def config_kwargs(ln_mid: bool) -> dict:
    return dict(text_config=dict(TEXT), audio_config=dict(AUDIO), hidden_size=256, stack_factor=8, projector_ln_mid=ln_mid)


def param(name: str, shape) -> torch.Tensor:
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    w = torch.randn(tuple(shape), generator=g) * (0.05 if len(shape) > 1 else 0.02)
    return w + 1.0 if (name.endswith("weight") and len(shape) == 1) else w      # norm weights around 1


def real_config_kwargs(ln_mid: bool = True, latency=None) -> dict:
    kw = dict(text_config=dict(TEXT), audio_config=dict(AUDIO_REAL), hidden_size=256, stack_factor=8, projector_ln_mid=ln_mid)
    if latency is not None:
        kw["audio_latency_block_size"] = latency
    return kw


def mel(n: int, frames: int, seed: int = 21) -> torch.Tensor:
    """Seeded stand-in for log-mel features (the Whisper range is roughly [-1, 1.5])."""
    return torch.randn(n, 80, frames, generator=torch.Generator().manual_seed(seed)) * 0.6


def tower_output() -> torch.Tensor:
    return torch.randn(N_AUDIO, 1500, AUDIO["d_model"], generator=torch.Generator().manual_seed(5)) * 0.5


def batch(mixed: bool = False) -> dict:
    """mixed: sample 0 is text-only, sample 1 carries three audio items, sample 2 one (audio_batch_size [0, 3, 1])."""
    ids = torch.randint(3, TEXT["vocab_size"], (B, T), generator=torch.Generator().manual_seed(3))
    labels = ids.clone()
    labels[:, :24] = -100
    mask = torch.ones(B, T, dtype=torch.long)
    mask[1, 38:] = 0                      # right padding (sample 1), labels ignored there
    labels[1, 38:] = -100
    mask[2, :5] = 0                       # left padding (sample 2)
    if mixed:
        return dict(input_ids=ids, labels=labels, attention_mask=mask,
                    audio_token_start_idx=torch.tensor([2, 12, 20, 8]), audio_token_len=torch.tensor([6, 5, 4, 9]),
                    audio_lens=torch.tensor([3000, 2600, 2000, 2999]), audio_batch_size=torch.tensor([0, 3, 1]))
    return dict(input_ids=ids, labels=labels, attention_mask=mask,
                audio_token_start_idx=torch.tensor([2, 13, 3, 8]), audio_token_len=torch.tensor([6, 5, 7, 9]),
                audio_lens=torch.tensor([3000, 2600, 2000, 2999]), audio_batch_size=torch.tensor([2, 1, 1]))


def generate_batch() -> dict:
    """Prompts for generate(): samples 0 (no padding, two audio items) and 2 (left padded, one item) of `batch()`."""
    b = batch()
    sel = [0, 2]
    return dict(input_ids=b["input_ids"][sel], attention_mask=b["attention_mask"][sel],
                audio_token_start_idx=torch.tensor([2, 13, 8]), audio_token_len=torch.tensor([6, 5, 9]),
                audio_lens=torch.tensor([3000, 2600, 2999]), audio_batch_size=torch.tensor([2, 1]))


GEN_AUDIO_ROWS = [0, 1, 3]        # rows of tower_output() behind generate_batch()'s three audio items


def alt_batch() -> dict:
    """The text-only (teacher) fields of the KL-distillation step for `batch()`: same number of supervised tokens per row as
    `labels` (20, 14, 20), a shorter sequence, row 1 right padded."""
    Ta = 36
    ids = torch.randint(3, TEXT["vocab_size"], (B, Ta), generator=torch.Generator().manual_seed(9))
    mask = torch.ones(B, Ta, dtype=torch.long)
    labels = torch.full((B, Ta), -100, dtype=torch.long)
    labels[0, 16:] = ids[0, 16:]
    labels[1, 18:32] = ids[1, 18:32]
    mask[1, 32:] = 0
    labels[2, 16:] = ids[2, 16:]
    return dict(alt_input_ids=ids, alt_attention_mask=mask, alt_labels=labels)


def speech_like_pcm(seconds: float = 30.0, seed: int = 77):
    """A 30 s harmonic, speech-like test signal for the log-mel frontend (white noise hides the accuracy of the bins near the
    per-clip `max - 8` floor: tonal audio has 6-8 decades of dynamic range between a harmonic and the gaps next to it):
    a glottal-like harmonic series (f0 gliding 90-220 Hz with vibrato, 1 / h roll-off up to 4 kHz) through three moving
    formant-like gains, syllable-rate amplitude modulation, pauses of exact silence, a faint noise floor in the voiced parts.
    float64 arithmetic from a seeded generator, rounded to float32 - reproducible, so fixtures store outputs only."""
    import numpy as np
    n = int(seconds * 16000)
    t = np.arange(n, dtype=np.float64) / 16000.0
    rng = np.random.RandomState(seed)
    f0 = 150.0 + 60.0 * np.sin(2 * np.pi * 0.31 * t) + 6.0 * np.sin(2 * np.pi * 5.3 * t)
    phase = 2 * np.pi * np.cumsum(f0) / 16000.0
    formants = [(500 + 250 * np.sin(2 * np.pi * 0.7 * t), 90.0), (1500 + 500 * np.sin(2 * np.pi * 0.43 * t + 1.0), 140.0),
                (2600 + 300 * np.sin(2 * np.pi * 0.2 * t + 2.0), 200.0)]
    x = np.zeros(n)
    for h in range(1, 27):
        fh = h * f0
        gain = sum(np.exp(-0.5 * ((fh - fc) / bw) ** 2) for fc, bw in formants) + 0.02
        x += np.where(fh < 4000.0, gain / h, 0.0) * np.sin(h * phase)
    env = np.clip(np.sin(2 * np.pi * 1.9 * t) * 1.5 + 0.4, 0.0, 1.0) ** 2
    x = x * env + 1e-4 * rng.randn(n) * (env > 0)
    for a, b in ((2.0, 2.6), (9.5, 10.4), (17.0, 17.3), (28.5, 30.0)):        # pauses: exact zeros
        x[int(a * 16000): int(b * 16000)] = 0.0
    x = 0.6 * x / np.abs(x).max()
    return x.astype(np.float32)
