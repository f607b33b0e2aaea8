"""Host logic of the optimizer step: learning-rate schedules pinned against the installed
transformers (the [3P] the reference's HF Trainer takes them from), gradient accumulation and step counting of
UltravoxTrainer with the device calls stubbed out."""
import math

import pytest
import torch

from ultravox_amd.model import UltravoxTrainer
from ultravox_amd.schedule import LRSchedule, lr_factor, resolve_warmup_steps


@pytest.mark.parametrize("name,kwargs", [("constant", {}), ("constant_with_warmup", {}), ("linear", {}), ("cosine", {}),
                                         ("cosine_with_min_lr", {"min_lr_rate": 0.1}), ("cosine_with_min_lr", {"min_lr": 2e-4})])
@pytest.mark.parametrize("warmup,total", [(0, 40), (7, 40), (1000, 3000)])
def test_schedules_match_transformers(name, kwargs, warmup, total):
    from transformers.optimization import get_scheduler
    base = 2e-3
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=base)
    sched = get_scheduler(name, opt, num_warmup_steps=warmup, num_training_steps=total, scheduler_specific_kwargs=kwargs or None)
    mine = LRSchedule(base, name, warmup, total, kwargs)
    steps = range(total) if total <= 100 else list(range(0, 1010, 37)) + list(range(1000, total, 113)) + [total - 1]
    k = 0
    for want_step in steps:
        while k < want_step:
            opt.step()
            sched.step()
            k += 1
        assert mine(k) == pytest.approx(opt.param_groups[0]["lr"], rel=1e-12, abs=1e-18), (name, k)


def test_warmup_ratio_and_the_release_recipe():
    assert resolve_warmup_steps(0.1, 995) == math.ceil(99.5) and resolve_warmup_steps(1000, 5) == 1000 and resolve_warmup_steps(0, 10) == 0
    s = LRSchedule(2e-3, "cosine_with_min_lr", 1000, 10000, {"min_lr_rate": 0.1})       # meta_config.yaml:28-31
    assert s(0) == 0.0 and s(500) == pytest.approx(1e-3) and s(1000) == pytest.approx(2e-3)
    assert s(10000) == pytest.approx(2e-4) and s(5500) == pytest.approx(2e-3 * (0.5 * 0.9 + 0.1))
    with pytest.raises(ValueError):
        LRSchedule(1e-3, "polynomial", 0, 10)
    with pytest.raises(ValueError):
        LRSchedule(1e-3, "cosine", 0, 0)
    assert lr_factor("constant", 123) == 1.0


class StubModel:
    """Stands in for UltravoxModel: forward_backward writes grad_scale * g_i into the flat gradient bucket."""
    dtype = torch.float32
    device = torch.device("cpu")

    def __init__(self, grads):
        self.proj_flat = torch.zeros(4)
        self.proj_grad = torch.zeros(4)
        self.grads, self.i, self._before_projector = grads, 0, None

    def train(self):
        pass

    def forward_backward(self, grad_scale=1.0, **batch):
        self.proj_grad.copy_(self.grads[self.i] * grad_scale)
        self.i += 1
        return torch.tensor(float(self.i))


def test_gradient_accumulation_steps_once_per_boundary_with_the_scheduled_lr():
    g = [torch.tensor([1.0, 2.0, 3.0, 4.0]) * (i + 1) for i in range(6)]
    model = StubModel(g)
    tr = UltravoxTrainer(model, lr=1e-2, gradient_accumulation_steps=3, lr_scheduler="linear", lr_warmup_steps=1, max_steps=4)
    seen = []
    tr._adamw = lambda lr: seen.append((lr, tr.step_count, model.proj_grad.clone()))
    for _ in range(6):
        tr.train_step(x=None)
    assert [s[1] for s in seen] == [1, 2]                                 # two optimizer steps for six micro-batches
    assert seen[0][0] == 0.0 and seen[1][0] == pytest.approx(1e-2)        # factor(0) = 0 (warm-up), factor(1) = 1
    assert torch.allclose(seen[0][2], (g[0] + g[1] + g[2]) / 3) and torch.allclose(seen[1][2], (g[3] + g[4] + g[5]) / 3)
    assert tr.last_lr == pytest.approx(1e-2) and tr._micro == 0


def test_default_trainer_keeps_a_constant_lr_and_steps_every_batch():
    model = StubModel([torch.ones(4)] * 3)
    tr = UltravoxTrainer(model, lr=3e-4)
    seen = []
    tr._adamw = lambda lr: seen.append(lr)
    for _ in range(3):
        tr.train_step(x=None)
    assert seen == [3e-4] * 3 and tr.step_count == 3


def test_generate_logit_policies_match_the_hf_processors():
    """Host policies generate() applies to the last-position logits, against the installed transformers' processors:
    repetition penalty (pipeline default 1.1) and the temperature / top-k / top-p warpers."""
    from transformers.generation.logits_process import (RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper,
                                                         TopKLogitsWarper, TopPLogitsWarper)
    from ultravox_amd.model import UltravoxModel
    g = torch.Generator().manual_seed(0)
    scores = torch.randn(3, 50, generator=g) * 3
    seen = torch.randint(0, 50, (3, 12), generator=g)
    want = RepetitionPenaltyLogitsProcessor(1.3)(seen, scores.clone())
    got = UltravoxModel._repetition_penalty(scores.clone(), seen, 1.3)
    assert torch.equal(got, want) and not torch.equal(got, scores)
    # the sampling distribution: same support and probabilities as HF's warper chain
    x = scores.clone()
    for w in (TemperatureLogitsWarper(0.7), TopKLogitsWarper(20), TopPLogitsWarper(0.8)):
        x = w(seen, x)
    p_want = torch.softmax(x, -1)
    counts = torch.zeros(3, 50)
    gen = torch.Generator().manual_seed(1)
    for _ in range(4000):
        counts[torch.arange(3), UltravoxModel._sample(scores, 0.7, 20, 0.8, gen)] += 1
    assert torch.all(counts[p_want == 0] == 0)                       # nothing outside the nucleus is ever drawn
    assert (counts / 4000 - p_want).abs().max() < 0.04
