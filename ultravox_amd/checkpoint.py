"""Checkpoint I/O with the reference's "diff state dict" semantics.

Reference: UltravoxModel.diff_state_dict / save_pretrained / _pre_load_state_dict_hook
(ultravox/model/ultravox_model.py:565-594): a saved checkpoint holds ONLY the trainable parameters plus every key a
previously loaded checkpoint already carried (`keep_params`); the frozen towers are re-created from
`audio_model_id` / `text_model_id`.  The file layout is what `transformers.PreTrainedModel.save_pretrained` writes for
that state dict — `model.safetensors` + `config.json` (`UltravoxConfig.to_diff_dict`, ultravox_config.py:188-203) — so a
checkpoint written here loads in the reference and vice versa (same key names, same dtypes).

Trainer state (step, AdamW moments, optional f32 master weights) goes to `optimizer.safetensors` + `trainer_state.json`:
the HF Trainer's resume contract (optimizer.pt / trainer_state.json) restated for the one flat projector bucket.

Pure host code: works on plain {name: tensor} dicts, no GPU needed.
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, Iterable, Optional, Set, Tuple

import torch
from safetensors.torch import load_file, save_file

from .config import UltravoxConfig

SAFE_WEIGHTS_NAME = "model.safetensors"      # transformers.utils.SAFE_WEIGHTS_NAME
CONFIG_NAME = "config.json"
OPTIMIZER_NAME = "optimizer.safetensors"
TRAINER_STATE_NAME = "trainer_state.json"
FSDP_INFIX = "_fsdp_wrapped_module."         # ultravox_model.py:573-577 normalises this away


def diff_state_dict(state_dict: Dict[str, torch.Tensor], trainable_params: Iterable[str],
                    keep_params: Iterable[str] = ()) -> Dict[str, torch.Tensor]:
    """ultravox_model.py:565-584: keep k iff k is trainable (FSDP wrapper infix stripped) or in keep_params."""
    trainable = {k.replace(FSDP_INFIX, "") for k in trainable_params}
    keep = set(keep_params)
    return {k: v for k, v in state_dict.items() if k in keep or k in trainable}


def save_pretrained(save_directory: str, config: UltravoxConfig, state_dict: Dict[str, torch.Tensor],
                    trainable_params: Iterable[str], keep_params: Iterable[str] = ()) -> Dict[str, torch.Tensor]:
    """ultravox_model.py:586-591 + PreTrainedModel.save_pretrained: config.json + model.safetensors of the diff."""
    os.makedirs(save_directory, exist_ok=True)
    diff = diff_state_dict(state_dict, trainable_params, keep_params)
    tensors = {k: v.detach().to("cpu").contiguous() for k, v in diff.items()}
    save_file(tensors, os.path.join(save_directory, SAFE_WEIGHTS_NAME), metadata={"format": "pt"})
    with open(os.path.join(save_directory, CONFIG_NAME), "w") as f:
        json.dump(config.to_diff_dict(), f, indent=2, sort_keys=True, default=str)
    return diff


def load_pretrained(directory: str) -> Tuple[UltravoxConfig, Dict[str, torch.Tensor]]:
    """-> (config, the checkpoint's state dict).  The caller merges it over the base towers' weights; every key found
    here becomes a keep_param of the loaded model (_pre_load_state_dict_hook, ultravox_model.py:593-594)."""
    with open(os.path.join(directory, CONFIG_NAME)) as f:
        cd = json.load(f)
    cd.pop("model_type", None)
    cd.pop("vocab_size", None)
    cd.pop("initializer_range", None)
    config = UltravoxConfig(**cd)
    return config, load_file(os.path.join(directory, SAFE_WEIGHTS_NAME))


SAFE_WEIGHTS_INDEX_NAME = "model.safetensors.index.json"


def load_hf_weights(directory: str) -> Dict[str, torch.Tensor]:
    """All tensors of a local HF checkpoint directory: `model.safetensors`, or the shards `model.safetensors.index.json`
    names.  Stands in for the hub download behind `from_pretrained(audio_model_id / text_model_id)`
    (ultravox_model.py:439-526): there is no network here, the towers' weights come from disk."""
    single, index = os.path.join(directory, SAFE_WEIGHTS_NAME), os.path.join(directory, SAFE_WEIGHTS_INDEX_NAME)
    if os.path.exists(single):
        return load_file(single)
    if not os.path.exists(index):
        raise FileNotFoundError(f"{directory}: neither {SAFE_WEIGHTS_NAME} nor {SAFE_WEIGHTS_INDEX_NAME} "
                                "(only safetensors checkpoints are read)")
    with open(index) as f:
        weight_map = json.load(f)["weight_map"]
    out: Dict[str, torch.Tensor] = {}
    for shard in sorted(set(weight_map.values())):
        out.update(load_file(os.path.join(directory, shard)))
    missing = set(weight_map) - set(out)
    if missing:
        raise KeyError(f"{directory}: the index lists tensors no shard holds: {sorted(missing)[:5]}")
    return out


def audio_tower_state_dict(directory: str, prefix: str = "audio_tower.") -> Dict[str, torch.Tensor]:
    """The encoder of a Whisper checkpoint under the reference's key names.  `ModifiedWhisperEncoder.base_model_prefix` is
    "model.encoder" (ultravox_model.py:818-820): a full `WhisperForConditionalGeneration` / `WhisperModel` file contributes
    its `model.encoder.*` (or `encoder.*`) tensors, the decoder and `proj_out` are ignored (:820), an encoder-only file is
    taken as is."""
    out = {}
    for k, v in load_hf_weights(directory).items():
        for head in ("model.encoder.", "encoder."):
            if k.startswith(head):
                out[prefix + k[len(head):]] = v
                break
        else:
            if not k.startswith(("model.decoder.", "decoder.", "proj_out.", "model.")):
                out[prefix + k] = v
    if prefix + "conv1.weight" not in out:
        raise KeyError(f"{directory}: no Whisper encoder weights found (looked for model.encoder.conv1.weight)")
    return out


def language_model_state_dict(directory: str, prefix: str = "language_model.") -> Dict[str, torch.Tensor]:
    """A causal-LM checkpoint (`model.*`, `lm_head.weight`) under the reference's `language_model.` prefix; a checkpoint with
    tied embeddings carries no `lm_head.weight`, which then IS the embedding matrix (HF `tie_word_embeddings`)."""
    sd = {prefix + k: v for k, v in load_hf_weights(directory).items()}
    if prefix + "lm_head.weight" not in sd:
        sd[prefix + "lm_head.weight"] = sd[prefix + "model.embed_tokens.weight"]
    return sd


def merge_state_dict(base: Dict[str, torch.Tensor], checkpoint: Dict[str, torch.Tensor],
                     strict_shapes: bool = True) -> Tuple[Dict[str, torch.Tensor], Set[str]]:
    """base (towers from their own ids + freshly initialised projector) overlaid with the checkpoint's keys.
    Unknown keys raise like load_state_dict(strict=True) does for unexpected keys."""
    out = dict(base)
    unexpected = [k for k in checkpoint if k not in base]
    if unexpected:
        raise KeyError(f"unexpected key(s) in checkpoint: {unexpected[:5]}{' ...' if len(unexpected) > 5 else ''}")
    for k, v in checkpoint.items():
        if strict_shapes and tuple(v.shape) != tuple(base[k].shape):
            raise ValueError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(base[k].shape)}")
        out[k] = v
    return out, set(checkpoint.keys())


def save_trainer_state(directory: str, step: int, tensors: Dict[str, torch.Tensor], extra: Optional[Dict[str, Any]] = None):
    os.makedirs(directory, exist_ok=True)
    save_file({k: v.detach().to("cpu").contiguous() for k, v in tensors.items() if v is not None},
              os.path.join(directory, OPTIMIZER_NAME), metadata={"format": "pt"})
    with open(os.path.join(directory, TRAINER_STATE_NAME), "w") as f:
        json.dump({"global_step": int(step), **(extra or {})}, f, indent=2, sort_keys=True)


def load_trainer_state(directory: str) -> Tuple[int, Dict[str, torch.Tensor], Dict[str, Any]]:
    with open(os.path.join(directory, TRAINER_STATE_NAME)) as f:
        st = json.load(f)
    return int(st.pop("global_step")), load_file(os.path.join(directory, OPTIMIZER_NAME)), st
