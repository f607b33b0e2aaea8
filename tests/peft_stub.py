"""A stand-in for `peft` (the reference pins ~0.11.1, pyproject.toml:15; it is not installed here and cannot be): the three
names the reference imports - `peft.LoraConfig`, `peft.get_peft_model`, `peft.peft_model.PeftModel`
(ultravox_model.py:6-7, :534-545, :690-709) - restated from peft's published LoRA algorithm, so that the REFERENCE's own
`apply_lora` can be run on HF towers in the build container and its result recorded as a fixture (tests/golden/make_golden.py).

What this makes checkable: which modules the reference's config adapts (LoraConfigSimplified.target_modules =
[k_proj, q_proj, linear_k, linear_q], suffix match), which parameters end up trainable, the state-dict key names a reference
checkpoint carries (`base_model.model.<path>.lora_A.default.weight`, wrapped linears under `.base_layer.`), and the forward /
gradients of the adapted tower.  What it cannot make checkable: peft's own arithmetic - that part is restated here
(peft/tuners/lora/layer.py `Linear.forward`, `LoraLayer.update_layer / reset_lora_parameters`; tuners_utils
`check_target_module_exists`; `mark_only_lora_as_trainable`) and says so.  Test infrastructure only.
"""
import math
import sys
import types

from torch import nn


class LoraConfig:
    def __init__(self, r=8, lora_alpha=8, target_modules=None, lora_dropout=0.0, bias="none", **kw):
        self.r, self.lora_alpha, self.target_modules, self.lora_dropout, self.bias = r, lora_alpha, target_modules, lora_dropout, bias
        self.extra = kw


class LoraLinear(nn.Module):
    """peft.tuners.lora.layer.Linear for one adapter named "default", dropout 0, no bias adaptation."""

    def __init__(self, base_layer: nn.Linear, r: int, lora_alpha: float):
        super().__init__()
        self.base_layer = base_layer
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base_layer.in_features, r, bias=False)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, base_layer.out_features, bias=False)})
        self.scaling = {"default": lora_alpha / r}
        nn.init.kaiming_uniform_(self.lora_A["default"].weight, a=math.sqrt(5))      # reset_lora_parameters
        nn.init.zeros_(self.lora_B["default"].weight)
        self.to(base_layer.weight.dtype)

    def forward(self, x):
        result = self.base_layer(x)
        a, b = self.lora_A["default"], self.lora_B["default"]
        return result + b(a(x.to(a.weight.dtype))) * self.scaling["default"]


class _LoraModel(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model


class PeftModel(nn.Module):
    def __init__(self, model, config):
        super().__init__()
        self.base_model = _LoraModel(model)
        self.peft_config = {"default": config}

    def forward(self, *a, **k):
        return self.base_model.model(*a, **k)

    def __getattr__(self, name):     # peft forwards unknown attributes to the wrapped model
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.base_model.model, name)


def get_peft_model(model, config):
    targets = list(config.target_modules)
    hit = [n for n, m in model.named_modules()
           if isinstance(m, nn.Linear) and (n in targets or any(n.endswith("." + t) for t in targets))]
    if not hit:
        raise ValueError(f"Target modules {targets} not found in the base model.")
    for name in hit:
        parent_name, _, leaf = name.rpartition(".")
        parent = model.get_submodule(parent_name) if parent_name else model
        setattr(parent, leaf, LoraLinear(getattr(parent, leaf), config.r, config.lora_alpha))
    for n, p in model.named_parameters():        # mark_only_lora_as_trainable, bias="none"
        p.requires_grad = "lora_" in n
    return PeftModel(model, config)


def install():
    """Put the stub into sys.modules under peft's names; returns a function that removes it again (a spec-less `peft` left
    behind breaks transformers' own find_spec("peft") probes)."""
    mod = types.ModuleType("peft")
    mod.LoraConfig, mod.PeftModel, mod.get_peft_model = LoraConfig, PeftModel, get_peft_model
    sub = types.ModuleType("peft.peft_model")
    sub.PeftModel = PeftModel
    mod.peft_model = sub
    had = {k: sys.modules.get(k) for k in ("peft", "peft.peft_model")}
    sys.modules["peft"], sys.modules["peft.peft_model"] = mod, sub

    def remove():
        for k, v in had.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return remove
