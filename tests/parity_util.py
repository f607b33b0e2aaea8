"""Helpers shared by the GPU parity tests: error measures and a small recorder so that the numbers a test measured on the
GPU box come back with gpurun (gpurun_out/parity/<name>.json) and can be committed under profiles/."""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_l2(a, b) -> float:
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def max_abs(a, b) -> float:
    return (a.float().cpu() - b.float().cpu()).abs().max().item()


def stage_errors(got, want) -> dict:
    return {"rel_l2": rel_l2(got, want), "max_abs": max_abs(got, want), "ref_rms": want.float().pow(2).mean().sqrt().item()}


def record(name: str, payload: dict) -> None:
    d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", "parity")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name + ".json"), "w") as f:
            json.dump(payload, f, indent=1, sort_keys=True)
    except OSError:
        pass


def width_config(text_id: str, audio_id: str, llm_layers: int, enc_layers: int, **kw):
    """A BASELINE.json configuration at full WIDTH and reduced depth (the f32 CPU oracle has to finish in seconds)."""
    from ultravox_amd.config import AUDIO_PRESETS, TEXT_PRESETS, UltravoxConfig
    tc = dict(TEXT_PRESETS[text_id], num_hidden_layers=llm_layers)
    ac = dict(AUDIO_PRESETS[audio_id], encoder_layers=enc_layers)
    return UltravoxConfig(text_config=tc, audio_config=ac, hidden_size=4096, stack_factor=8, projector_ln_mid=True,
                          torch_dtype="bfloat16", **kw)


def oracle_threads() -> int:
    """torch's CPU GEMM on the GPU boxes peaks around 32 threads (bench.py calibrates the same way)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = min(32, n)
    torch.set_num_threads(n)
    return n
