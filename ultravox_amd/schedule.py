"""Learning-rate schedules of the reference's training recipes, as pure functions of the optimizer step.

The reference hands `lr_scheduler`, `lr_scheduler_kwargs` and `lr_warmup_steps` to the [3P] HF Trainer
(training/train.py:288-293; defaults config_base.py:150-153; the release recipes use `cosine_with_min_lr`,
`min_lr_rate` 0.1, 1000 warm-up steps: training/configs/meta_config.yaml:28-31).  HF builds a `LambdaLR` from them and steps it
AFTER each optimizer step, so optimizer step k (counting from 0) runs with `lr * factor(k)` — the very first step of a
warmed-up run has lr 0.  The factors below restate transformers.optimization's lambdas and are pinned against the installed
transformers in tests/test_trainer_cpu.py."""
import math
from typing import Any, Dict, Optional


def resolve_warmup_steps(lr_warmup_steps: float, num_training_steps: int) -> int:
    """train.py:292-293: a value below 1 is a RATIO of the run (HF: ceil(num_training_steps * warmup_ratio))."""
    if lr_warmup_steps < 1:
        return int(math.ceil(num_training_steps * lr_warmup_steps))
    return int(lr_warmup_steps)


def lr_factor(name: str, step: int, num_warmup_steps: int = 0, num_training_steps: int = 0, **kwargs: Any) -> float:
    if name == "constant":
        return 1.0
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    if name == "constant_with_warmup":
        return 1.0
    if num_training_steps <= 0:
        raise ValueError(f"lr_scheduler '{name}' needs the number of training steps (max_steps)")
    if name == "linear":
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    if name == "cosine":
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(kwargs.get("num_cycles", 0.5)) * 2.0 * progress)))
    if name == "cosine_with_min_lr":
        if kwargs.get("min_lr") is not None:
            raise ValueError("cosine_with_min_lr: pass min_lr_rate (min_lr needs the base lr; use LRSchedule)")
        rate = float(kwargs.get("min_lr_rate", 0.0))
        f = 0.5 * (1.0 + math.cos(math.pi * float(kwargs.get("num_cycles", 0.5)) * 2.0 * progress))
        return max(0.0, f * (1.0 - rate) + rate)
    raise ValueError(f"lr_scheduler '{name}' is not built (constant, constant_with_warmup, linear, cosine, cosine_with_min_lr)")


class LRSchedule:
    def __init__(self, base_lr: float, name: str = "constant", lr_warmup_steps: float = 0, num_training_steps: int = 0,
                 lr_scheduler_kwargs: Optional[Dict[str, Any]] = None):
        self.base_lr, self.name, self.total = float(base_lr), name, int(num_training_steps)
        self.warmup = resolve_warmup_steps(lr_warmup_steps, self.total)
        self.kwargs = dict(lr_scheduler_kwargs or {})
        if name == "cosine_with_min_lr" and self.kwargs.get("min_lr") is not None:      # HF: min_lr / base lr becomes the rate
            self.kwargs["min_lr_rate"] = float(self.kwargs.pop("min_lr")) / self.base_lr
        lr_factor(name, 0, self.warmup, self.total, **self.kwargs)      # reject unknown names / missing max_steps up front

    def __call__(self, optimizer_step: int) -> float:
        """lr of optimizer step `optimizer_step` (0-based: the number of optimizer steps already taken)."""
        return self.base_lr * lr_factor(self.name, optimizer_step, self.warmup, self.total, **self.kwargs)
