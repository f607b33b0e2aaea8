"""Validation loss over a dataset, sharded across the data-parallel ranks: the forward-only use of the hot path
(SURVEY.md §3.2).  Mirrors the reference's `validate_dataset` / `process_batch` (ultravox/evaluation/validate.py:22-114) and
`sharded_batch_iterator` (ultravox/training/ddp_utils.py:49-71):

  * rank r takes samples r, r + W, r + 2W, ... of the processed dataset in batches of `batch_size` (a short last batch is
    kept, not dropped);
  * each batch contributes `loss * n` and `n`, n = labels != -100 — the reference counts the UNSHIFTED labels although the
    loss is a mean over the shifted ones (one more than the terms in the mean per sequence); kept as is, it is the number
    the reference reports;
  * the two scalars are summed over ranks (the only collectives of this path: two scalar all-reduces over RCCL / gloo) and
    divided once: a token-weighted mean over the whole dataset, 0.0 for an empty one.
"""
from typing import Any, Callable, Dict, Generator, Iterable, List, Tuple

import torch
import torch.distributed as dist


def sharded_iterator(ds: Iterable, num_shards: int, shard_index: int):
    for i, sample in enumerate(ds):
        if i % num_shards == shard_index:
            yield sample


def sharded_batch_iterator(ds: Iterable, batch_size: int, num_shards: int,
                           shard_index: int) -> Generator[List[Tuple[int, Any]], None, None]:
    batch: List[Tuple[int, Any]] = []
    for idx, sample in enumerate(ds):
        if idx % num_shards != shard_index:
            continue
        batch.append((idx, sample))
        if len(batch) == batch_size:
            yield batch
            batch = []
    if batch:
        yield batch


def process_batch(model, data_collator: Callable[[List[Dict[str, Any]]], Dict[str, torch.Tensor]],
                  batch: List[Tuple[int, Dict[str, Any]]]) -> Tuple[float, int]:
    """-> (loss * number of valid labels, number of valid labels) of one batch of processed samples."""
    collated = data_collator([sample for _, sample in batch])
    collated = {k: v.to(model.device) for k, v in collated.items()}
    loss = model(**collated).loss
    n = int((collated["labels"] != -100).sum().item())
    return float(loss.item()) * n, n


def _dp() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def validate_dataset(model, data_collator, processed_dataset: Iterable, batch_size: int = 1) -> float:
    """`processed_dataset` yields what `UltravoxDataproc` produces (the reference wraps the raw dataset itself,
    `model.wrap_with_data_proc`, validate.py:62)."""
    rank, world = _dp()
    total_loss, total_labels = 0.0, 0
    if hasattr(model, "eval"):
        model.eval()
    with torch.no_grad():
        for batch in sharded_batch_iterator(processed_dataset, batch_size, world, rank):
            batch_loss, batch_labels = process_batch(model, data_collator, batch)
            total_loss += batch_loss
            total_labels += batch_labels
    if world > 1:
        dev = model.device if dist.get_backend() != "gloo" else "cpu"
        t_loss = torch.tensor(total_loss, device=dev, dtype=torch.float32)      # torch.tensor(python float): f32, as the reference
        t_labels = torch.tensor(total_labels, device=dev)
        dist.all_reduce(t_loss, op=dist.ReduceOp.SUM)
        dist.all_reduce(t_labels, op=dist.ReduceOp.SUM)
        total_loss, total_labels = float(t_loss.item()), int(t_labels.item())
    return total_loss / total_labels if total_labels > 0 else 0.0
