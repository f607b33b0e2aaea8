"""Data parallelism of the adapter-training step: one process per GPU (torchrun), frozen towers
replicated, ONE collective per step — the mean of the flat projector-gradient bucket.

Reference semantics: HF Trainer + accelerate wrap the model in torch DDP, which
all-reduces (sum) the gradients of the trainable parameters and divides by the world size; every rank's
loss is a mean over ITS OWN supervised tokens (accepts_loss_kwargs = False, ultravox_model.py:50-53), so the
DP result is the mean of per-rank means — never re-weighted by token counts.  The batch is dispatched by
slicing dim 0 of the global batch (accelerate split_batches=True, train.py:273-284).

On MI355X the collective is RCCL over xGMI (torch.distributed backend "nccl"); the same code runs on
gloo for the CPU tests.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def dp_mean_(flat: torch.Tensor) -> torch.Tensor:
    """In-place DDP gradient averaging of one flat bucket: all-reduce(sum) then * 1/W."""
    w = world_size()
    if w > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / w)
    return flat


def shard_batch(batch: Dict[str, torch.Tensor], rank_: int, world: int) -> Dict[str, torch.Tensor]:
    """Rank `rank_`'s slice of a global batch: dim-0 slicing of the per-sample tensors; the per-audio
    tensors (audio_values / audio_lens / audio_token_len / audio_token_start_idx) follow their owning
    samples through audio_batch_size."""
    B = batch["input_ids"].shape[0]
    if B % world != 0:
        raise ValueError(f"global batch {B} is not divisible by world size {world}")
    per = B // world
    lo, hi = rank_ * per, (rank_ + 1) * per
    out = {}
    per_audio = ("audio_values", "audio_lens", "audio_token_len", "audio_token_start_idx")
    if "audio_batch_size" in batch:
        counts = batch["audio_batch_size"].reshape(-1).tolist()
        a_lo, a_hi = int(sum(counts[:lo])), int(sum(counts[:hi]))
    for k, v in batch.items():
        if k in per_audio:
            out[k] = v[a_lo:a_hi]
        elif isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == B:
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out
