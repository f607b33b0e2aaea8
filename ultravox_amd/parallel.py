"""Data parallelism of the adapter-training step: one process per GPU (torchrun), frozen towers
replicated, ONE collective per step — the mean of the flat projector-gradient bucket.

Reference semantics: HF Trainer + accelerate wrap the model in torch DDP, which
all-reduces (sum) the gradients of the trainable parameters and divides by the world size; every rank's
loss is a mean over ITS OWN supervised tokens (accepts_loss_kwargs = False, ultravox_model.py:50-53), so the
DP result is the mean of per-rank means — never re-weighted by token counts.  The batch is dispatched by
slicing dim 0 of the global batch (accelerate split_batches=True, train.py:273-284).

On MI355X the collective is RCCL over xGMI: through torch.distributed (backend "nccl" IS RCCL on ROCm; the default, and what
the gloo CPU tests exercise) or through the library's own uvx_comm_* entry points (`UvxComm`: the same exchange behind the C
ABI, for hosts that drive libuvx.so without torch.distributed - include/uvx.h "data-parallel exchange").
"""
from __future__ import annotations

from typing import Dict, Optional

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


class UvxComm:
    """RCCL communicator owned by libuvx.so (uvx_comm_*): one per process, bound to the current HIP device.

    `UvxComm(rank, world, unique_id)` is the raw constructor (rank 0 obtains the id from `UvxComm.unique_id()` and ships it to
    the other ranks); `UvxComm.from_torch_distributed()` uses an initialised torch.distributed group only as that side channel
    (an object broadcast of 128 bytes) - the gradients themselves never pass through torch.distributed."""

    def __init__(self, rank_: int, world: int, unique_id: bytes):
        import ctypes as C
        from . import _lib
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes returned by UvxComm.unique_id()")
        self._C, self._lib = C, _lib
        self._h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.check(_lib.lib().uvx_comm_init(C.byref(self._h), int(rank_), int(world), buf), "uvx_comm_init")
        self.rank, self.world = int(rank_), int(world)

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib
        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.lib().uvx_comm_unique_id(buf), "uvx_comm_unique_id")
        return bytes(buf)

    @classmethod
    def from_torch_distributed(cls) -> "UvxComm":
        box = [cls.unique_id() if rank() == 0 else None]
        if world_size() > 1:
            dist.broadcast_object_list(box, src=0)
        return cls(rank(), world_size(), box[0])

    @staticmethod
    def rccl_version() -> int:
        from . import _lib
        return int(_lib.lib().uvx_comm_version())

    def all_reduce_mean_(self, flat: torch.Tensor, stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
        """In place: flat = sum over ranks / world (torch DDP's gradient averaging), asynchronous on `stream` (default: the
        current stream)."""
        if flat.dtype != torch.float32 or not flat.is_contiguous() or not flat.is_cuda:
            raise ValueError("all_reduce_mean_ takes a contiguous f32 device tensor (the flat gradient bucket)")
        st = torch.cuda.current_stream() if stream is None else stream
        C = self._C
        self._lib.check(self._lib.lib().uvx_comm_allreduce_f32(self._h, C.c_void_p(st.cuda_stream), C.c_void_p(flat.data_ptr()),
                                                               C.c_int64(flat.numel()), C.c_float(1.0 / self.world)),
                        "uvx_comm_allreduce_f32")
        return flat

    def close(self) -> None:
        if self._h:
            self._lib.check(self._lib.lib().uvx_comm_destroy(self._h), "uvx_comm_destroy")
            self._h = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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
