"""Host-side tables for the on-device log-mel frontend (K1) and the feature-extractor call contract.

`WhisperFeatureExtractor` below takes the place of the [3P] transformers WhisperFeatureExtractor that
the reference calls at ultravox_processing.py:295-303 with
`padding="longest", pad_to_multiple_of=hop_length, truncation=False, return_attention_mask=True`;
it keeps that call's inputs/outputs (input_features [B, n_mels, F] f32, attention_mask [B, F]) but the
STFT + mel + log runs in libuvx (uvx_logmel) on the GPU.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

N_FFT, HOP, N_BINS, N_BINS_PAD = 400, 160, 201, 208


def hertz_to_mel_slaney(freq: np.ndarray) -> np.ndarray:
    f_sp = 200.0 / 3
    min_log_hertz, min_log_mel, logstep = 1000.0, 15.0, 27.0 / np.log(6.4)
    freq = np.asarray(freq, dtype=np.float64)
    mels = freq / f_sp
    log_region = freq >= min_log_hertz
    mels[log_region] = min_log_mel + np.log(freq[log_region] / min_log_hertz) * logstep
    return mels


def mel_to_hertz_slaney(mels: np.ndarray) -> np.ndarray:
    f_sp = 200.0 / 3
    min_log_hertz, min_log_mel, logstep = 1000.0, 15.0, np.log(6.4) / 27.0
    mels = np.asarray(mels, dtype=np.float64)
    freq = f_sp * mels
    log_region = mels >= min_log_mel
    freq[log_region] = min_log_hertz * np.exp(logstep * (mels[log_region] - min_log_mel))
    return freq


def mel_filter_bank(n_mels: int, n_bins: int = N_BINS, fmin: float = 0.0, fmax: float = 8000.0,
                    sampling_rate: int = 16000) -> np.ndarray:
    """Slaney-scale, slaney-normalised triangular filters, [n_bins, n_mels] float64 — the filterbank
    Whisper's feature extractor builds (norm="slaney", mel_scale="slaney", 0-8000 Hz)."""
    mel_pts = np.linspace(hertz_to_mel_slaney(np.array([fmin]))[0], hertz_to_mel_slaney(np.array([fmax]))[0], n_mels + 2)
    filter_freqs = mel_to_hertz_slaney(mel_pts)
    fft_freqs = np.linspace(0, sampling_rate // 2, n_bins)
    fdiff = np.diff(filter_freqs)
    slopes = filter_freqs[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (filter_freqs[2:n_mels + 2] - filter_freqs[:n_mels])
    return fb * enorm[None, :]


def logmel_tables(n_mels: int):
    """window[400], tw_cos/tw_sin [400, 208] (k fastest, zero padded), mel_fb [n_mels, 208], all f32."""
    n = np.arange(N_FFT, dtype=np.float64)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)  # torch.hann_window(400), periodic
    k = np.arange(N_BINS, dtype=np.float64)
    # exact argument reduction: (k*n) mod 400 before the multiply by 2*pi/400
    kn = (np.outer(n.astype(np.int64), k.astype(np.int64)) % N_FFT).astype(np.float64)
    ang = 2.0 * np.pi * kn / N_FFT
    tw_cos = np.zeros((N_FFT, N_BINS_PAD), np.float32)
    tw_sin = np.zeros((N_FFT, N_BINS_PAD), np.float32)
    tw_cos[:, :N_BINS] = np.cos(ang)
    tw_sin[:, :N_BINS] = -np.sin(ang)
    fb = np.zeros((n_mels, N_BINS_PAD), np.float32)
    fb[:, :N_BINS] = mel_filter_bank(n_mels).T.astype(np.float32)
    return window.astype(np.float32), tw_cos, tw_sin, fb


class WhisperFeatureExtractor:
    """Device log-mel with the HF feature extractor's call contract (the subset the reference uses)."""

    def __init__(self, feature_size: int = 80, sampling_rate: int = 16000, hop_length: int = HOP,
                 chunk_length: int = 30, n_fft: int = N_FFT, device: str = "cuda"):
        if hop_length != HOP or n_fft != N_FFT or sampling_rate != 16000:
            raise ValueError("the device frontend is built for Whisper's n_fft=400 / hop=160 / 16 kHz")
        self.feature_size = feature_size
        self.sampling_rate = sampling_rate
        self.hop_length = hop_length
        self.n_fft = n_fft
        self.chunk_length = chunk_length
        self.n_samples = chunk_length * sampling_rate
        self.nb_max_frames = self.n_samples // hop_length
        self.device = device
        self.model_input_names = ["input_features"]
        self._tables = None

    @property
    def feature_extractor(self):  # the reference reads audio_processor.feature_extractor.hop_length
        return self

    def _device_tables(self):
        import torch

        if self._tables is None:
            self._tables = tuple(torch.from_numpy(t).to(self.device) for t in logmel_tables(self.feature_size))
        return self._tables

    def logmel_device(self, pcm):
        """pcm: torch f32 [B, L] on the GPU, L % 160 == 0 -> [B, n_mels, L/160] f32 (device)."""
        import torch
        from . import _lib

        assert pcm.is_cuda and pcm.dtype == torch.float32 and pcm.is_contiguous()
        B, L = pcm.shape
        F = L // HOP
        window, tw_cos, tw_sin, fb = self._device_tables()
        out = torch.empty((B, self.feature_size, F), device=pcm.device, dtype=torch.float32)
        scratch = torch.empty((B * ((F + 31) // 32) + 1,), device=pcm.device, dtype=torch.float32)
        _lib.check(_lib.lib().uvx_logmel(_lib.stream_ptr(), _lib.ptr(pcm), _lib.ptr(window), _lib.ptr(tw_cos),
                                         _lib.ptr(tw_sin), _lib.ptr(fb), _lib.ptr(out), _lib.ptr(scratch),
                                         C.c_int32(B), C.c_int32(L), C.c_int32(self.feature_size), C.c_int32(F)),
                   "uvx_logmel")
        return out

    def __call__(self, raw_speech: Sequence[np.ndarray], sampling_rate: Optional[int] = None,
                 padding: str = "longest", pad_to_multiple_of: Optional[int] = None, truncation: bool = False,
                 return_attention_mask: bool = True, return_tensors: Optional[str] = None, **kwargs):
        import torch

        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(
                f"The model corresponding to this feature extractor: {self.__class__.__name__} was trained using a"
                f" sampling rate of {self.sampling_rate}. Please make sure that the provided `raw_speech` input"
                f" was sampled with {self.sampling_rate} and not {sampling_rate}.")
        if padding != "longest" or truncation:
            raise ValueError("only padding='longest', truncation=False (the reference's call) is built")
        if isinstance(raw_speech, np.ndarray) and raw_speech.ndim == 1:
            raw_speech = [raw_speech]
        raw: List[np.ndarray] = [np.asarray(x, dtype=np.float32) for x in raw_speech]
        lens = [len(x) for x in raw]
        L = max(lens)
        if pad_to_multiple_of:
            L = (L + pad_to_multiple_of - 1) // pad_to_multiple_of * pad_to_multiple_of
        batch = np.zeros((len(raw), L), np.float32)
        mask = np.zeros((len(raw), L), np.int32)
        for i, x in enumerate(raw):
            batch[i, :len(x)] = x
            mask[i, :len(x)] = 1
        feats = self.logmel_device(torch.from_numpy(batch).to(self.device))
        # sample mask -> frame mask exactly as the HF extractor rescales it
        fmask = mask[:, ::self.hop_length]
        if L % self.hop_length != 0:
            fmask = fmask[:, :-1]
        out = {"input_features": feats, "attention_mask": torch.from_numpy(np.ascontiguousarray(fmask))}
        return out


class Wav2Vec2FeatureExtractor:
    """Call contract of the [3P] transformers Wav2Vec2FeatureExtractor (feature_size 1, do_normalize True) for the raw-waveform
    tower of BASELINE.json config 5: every clip is normalised to zero mean / unit variance over its OWN samples
    (zero_mean_unit_var_norm: (x - mean) / sqrt(var + 1e-7)), padded with zeros to the longest clip, and returned as
    `input_values` [B, L] f32 - what ultravox_processing.py:308 falls back to when there is no `input_features`.
    wav2vec2-large-960h is a feat_extract_norm="group" model: `return_attention_mask` defaults to False there and the model
    runs without a mask; the sample-count mask is still returned on request because the processor needs the lengths.
    Host arithmetic (a mean and a variance per clip); no device work."""

    model_input_names = ["input_values", "attention_mask"]

    def __init__(self, sampling_rate: int = 16000, padding_value: float = 0.0, do_normalize: bool = True):
        self.sampling_rate, self.padding_value, self.do_normalize, self.feature_size = sampling_rate, padding_value, do_normalize, 1

    @property
    def feature_extractor(self):
        return self

    def __call__(self, raw_speech, sampling_rate: Optional[int] = None, padding="longest", pad_to_multiple_of: Optional[int] = None,
                 truncation: bool = False, return_attention_mask: bool = False, return_tensors=None, **kwargs):
        import torch
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(f"The model corresponding to this feature extractor was trained using a sampling rate of "
                             f"{self.sampling_rate}. Please make sure that the provided `raw_speech` input was sampled with "
                             f"{self.sampling_rate} and not {sampling_rate}.")
        clips = [np.asarray(x, dtype=np.float32).reshape(-1) for x in (raw_speech if isinstance(raw_speech, (list, tuple)) else [raw_speech])]
        lens = [len(x) for x in clips]
        width = max(lens)
        if pad_to_multiple_of:
            width = -(-width // pad_to_multiple_of) * pad_to_multiple_of
        out = np.full((len(clips), width), self.padding_value, dtype=np.float32)
        for i, x in enumerate(clips):
            if self.do_normalize:
                x = (x - x.mean()) / np.sqrt(x.var() + 1e-7)
            out[i, : len(x)] = x
        data = {"input_values": torch.from_numpy(out)}
        if return_attention_mask:
            data["attention_mask"] = torch.from_numpy((np.arange(width)[None, :] < np.asarray(lens)[:, None]).astype(np.int32))
        return data
