"""The N > 1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Checks the data-parallel
semantics the reference inherits from DDP: flat-bucket mean over ranks, each rank's loss
a mean over ITS OWN supervised tokens (mean of per-rank means, not a token-weighted global mean), batch
dispatch by dim-0 slicing including the per-audio tensors."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import reference_cpu as O
from ultravox_amd.config import UltravoxConfig
from ultravox_amd.parallel import dp_mean_, shard_batch
from ultravox_amd.weights import random_state_dict

CFG = dict(
    audio_config=dict(d_model=64, encoder_layers=1, encoder_attention_heads=2, encoder_ffn_dim=128),
    text_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2,
                     num_key_value_heads=1, vocab_size=128, eos_token_id=2),
    hidden_size=64, projector_ln_mid=True)


def _global_batch(cfg):
    b = O.synthetic_batch(cfg, 4, 1.0, n_text=12, audio_start=3, n_supervised=4)
    b["audio_values"] = O.logmel_ref(b.pop("pcm"), 80)
    b["labels"][0, :-1] = -100  # sample 0 supervises 1 token, the others 4: per-rank token counts differ
    return b


def _flat(grads, keys):
    return torch.cat([grads[k].reshape(-1) for k in keys])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg = UltravoxConfig(**CFG)
    om = O.OracleModel(cfg, random_state_dict(cfg, seed=3))
    shard = shard_batch(_global_batch(cfg), rank, world)
    out, grads, _ = om.train_step(shard)
    flat = _flat(grads, sorted(grads))
    dp_mean_(flat)
    q.put((rank, out["loss"].item(), flat))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_batch_follows_audio_items():
    cfg = UltravoxConfig(**CFG)
    b = _global_batch(cfg)
    b["audio_batch_size"] = torch.tensor([2, 0, 1, 1])          # 4 audio items: sample 0 owns two, sample 1 none
    s0, s1 = shard_batch(b, 0, 2), shard_batch(b, 1, 2)
    assert s0["input_ids"].shape[0] == s1["input_ids"].shape[0] == 2
    assert s0["audio_values"].shape[0] == 2 and s1["audio_values"].shape[0] == 2
    assert torch.equal(s1["audio_token_start_idx"], b["audio_token_start_idx"][2:4])
    with pytest.raises(ValueError, match="not divisible"):
        shard_batch(b, 0, 3)


def test_two_rank_gloo_gradient_mean_matches_ddp_semantics():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(res[0][2], res[1][2])                     # every rank holds the same averaged bucket

    # single-process restatement: gradient of mean_r(loss_r) where loss_r is rank r's own token mean
    cfg = UltravoxConfig(**CFG)
    om = O.OracleModel(cfg, random_state_dict(cfg, seed=3))
    b = _global_batch(cfg)
    losses = [om.forward(**shard_batch(b, r, world))["loss"] for r in range(world)]
    (sum(losses) / world).backward()
    want = _flat({k: om.sd[k].grad for k in om.trainable}, sorted(om.trainable))
    assert torch.allclose(res[0][2], want, rtol=1e-4, atol=1e-7)
    assert abs(res[0][1] - losses[0].item()) < 1e-5 and abs(res[1][1] - losses[1].item()) < 1e-5
    # ... which is NOT the token-weighted global mean whenever ranks hold different token counts
    om2 = O.OracleModel(cfg, random_state_dict(cfg, seed=3))
    om2.forward(**b)["loss"].backward()
    glob = _flat({k: om2.sd[k].grad for k in om2.trainable}, sorted(om2.trainable))
    assert (glob - want).norm() > 1e-3 * want.norm()
