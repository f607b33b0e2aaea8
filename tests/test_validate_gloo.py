"""validate_dataset: the forward-only pass sharded over data-parallel ranks (reference evaluation/validate.py:45-114,
training/ddp_utils.py:49-71) — 2 processes, gloo, with a stub model whose loss is a known function of the batch."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ultravox_amd.validate import sharded_batch_iterator, sharded_iterator, validate_dataset


def test_sharded_batch_iterator_keeps_indices_and_the_short_tail():
    ds = [f"s{i}" for i in range(11)]
    got = list(sharded_batch_iterator(ds, 2, 2, 1))
    assert got == [[(1, "s1"), (3, "s3")], [(5, "s5"), (7, "s7")], [(9, "s9")]]
    assert list(sharded_batch_iterator(ds, 4, 1, 0))[-1] == [(8, "s8"), (9, "s9"), (10, "s10")]
    assert list(sharded_batch_iterator([], 2, 2, 0)) == [] and list(sharded_iterator(ds, 3, 2)) == ["s2", "s5", "s8"]


class StubModel:
    """loss = mean of the supervised label VALUES of the batch (any deterministic function of the batch would do)."""
    device = torch.device("cpu")

    def eval(self):
        return self

    def __call__(self, input_ids, labels, **kw):
        lab = labels[labels != -100].float()
        return type("Out", (), {"loss": lab.mean() if lab.numel() else torch.tensor(0.0)})()


def collate(feats):
    n = max(len(f["labels"]) for f in feats)
    pad = lambda x, v: torch.tensor([list(f[x]) + [v] * (n - len(f[x])) for f in feats])
    return {"input_ids": pad("input_ids", 0), "labels": pad("labels", -100)}


def dataset():
    g = torch.Generator().manual_seed(0)
    out = []
    for i in range(9):
        n = 3 + i % 4
        lab = torch.randint(1, 50, (n,), generator=g).tolist()
        for j in range(i % 3):
            lab[j] = -100
        out.append({"input_ids": list(range(n)), "labels": lab})
    return out


def expected(ds, batch_size, world):
    tot, cnt = 0.0, 0
    for r in range(world):
        for batch in sharded_batch_iterator(ds, batch_size, world, r):
            vals = [v for _, f in batch for v in f["labels"] if v != -100]
            tot += float(torch.tensor(vals, dtype=torch.float32).mean()) * len(vals)
            cnt += len(vals)
    return tot / cnt


def test_single_process_is_a_token_weighted_mean():
    ds = dataset()
    assert abs(validate_dataset(StubModel(), collate, ds, batch_size=2) - expected(ds, 2, 1)) < 1e-6
    assert validate_dataset(StubModel(), collate, [], batch_size=2) == 0.0


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank, validate_dataset(StubModel(), collate, dataset(), batch_size=2)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_the_dataset_and_agree_on_the_global_mean():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = expected(dataset(), 2, 2)
    assert abs(got[0] - want) < 1e-5 and got[0] == got[1]
