"""The three names the reference's entry points import from `peft` — `LoraConfig`, `get_peft_model`, `prepare_model_for_kbit_training`
(`reason.py:25, 362-392`; `train_dna_qwen.py:16, 136-177`) — and `PeftModel.from_pretrained` (`reason.py:430-446`), for the HIP
text model.

The reference wraps `model.text_model` with PEFT.  `bioreason_amd.modeling.Qwen3ForCausalLM` keeps the module tree PEFT produces
(`q_proj.base_layer`, `q_proj.lora_A.default`, ...), but its arithmetic runs in `engine.QwenEngine` over a flat parameter arena: a real
`peft.get_peft_model` would wrap the `nn.Linear` shells, the engine would never read those wrappers, and the adapters would train
nothing — silently.  So the call is dispatched on the model type:
  * a HIP text model  -> `Qwen3ForCausalLM.apply_lora(...)` (the adapters the engine executes), the SAME object is returned, carrying
    the PEFT attributes the scripts read afterwards (`peft_config`, `active_adapter`, `base_model.model`, `print_trainable_parameters`,
    `merge_and_unload`, `disable_adapter`);
  * anything else      -> the real `peft` if it is installed, else an ImportError that says so.
`compat/peft/` is a package of that name for `PYTHONPATH`: it re-exports an installed peft and overrides these names,
so that the scripts' `from peft import get_peft_model, LoraConfig, prepare_model_for_kbit_training` needs no edit.
`DNALLMModel.__setattr__` refuses a foreign wrapper as `text_model`, so the silent form cannot happen by another route either."""
from __future__ import annotations

import importlib
import json
import os
import sys
from typing import Any, Iterable, Optional

import torch.nn as nn

from .modeling import LORA_TARGETS, Qwen3ForCausalLM


def _real_peft():
    """the installed peft package (not the shim under compat/), or None"""
    me = sys.modules.get("peft")
    shim_dir = None
    if me is not None and getattr(me, "__bioreason_amd_shim__", False):
        shim_dir = os.path.dirname(os.path.dirname(os.path.abspath(me.__file__)))
    elif me is not None:
        return me
    saved_mod, saved_path = sys.modules.pop("peft", None), list(sys.path)
    try:
        if shim_dir is not None:
            sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != shim_dir]
        try:
            real = importlib.import_module("peft")
        except ImportError:
            return None
        return None if getattr(real, "__bioreason_amd_shim__", False) else real
    finally:
        for k in [k for k in sys.modules if k == "peft" or k.startswith("peft.")]:
            if saved_mod is not None and getattr(saved_mod, "__bioreason_amd_shim__", False):
                sys.modules.setdefault("_real_" + k, sys.modules[k])
                del sys.modules[k]
        if saved_mod is not None:
            sys.modules["peft"] = saved_mod
        sys.path[:] = saved_path


class LoraConfig:
    """peft.LoraConfig's fields as the scripts set them (`reason.py:376-384`, `train_dna_qwen.py:155-163`); unknown keywords are kept"""

    def __init__(self, r: int = 8, lora_alpha: float = 8, lora_dropout: float = 0.0, target_modules: Optional[Iterable[str]] = None,
                 init_lora_weights: Any = True, bias: str = "none", task_type: Optional[str] = None, **extra):
        self.r, self.lora_alpha, self.lora_dropout = int(r), lora_alpha, float(lora_dropout)
        self.target_modules = None if target_modules is None else (target_modules if isinstance(target_modules, str) else list(target_modules))
        self.init_lora_weights, self.bias, self.task_type = init_lora_weights, bias, task_type
        self.peft_type, self.inference_mode = "LORA", False
        for k, v in extra.items():
            setattr(self, k, v)

    def to_dict(self):
        return {k: v for k, v in vars(self).items()}

    def __repr__(self):
        return f"LoraConfig({self.to_dict()})"


def is_hip_text_model(model) -> bool:
    return isinstance(model, Qwen3ForCausalLM)


def _cfg(peft_config, key, default):
    return peft_config.get(key, default) if isinstance(peft_config, dict) else getattr(peft_config, key, default)


class _BaseModelView:
    """`peft_model.base_model` as the scripts use it: `.model` is the wrapped model, `.config` its config (`reason.py:71-72`)"""

    def __init__(self, model):
        self.model, self.config = model, model.config


def prepare_model_for_kbit_training(model, use_gradient_checkpointing: bool = True, gradient_checkpointing_kwargs=None):
    """peft.prepare_model_for_kbit_training: every parameter frozen (the adapters come afterwards).  peft also widens bf16 / fp16
    parameters to fp32 and switches gradient checkpointing on; the HIP engine keeps the frozen base in bf16 by design (DESIGN.md,
    parity definition) and recomputes nothing it does not have to — both are statements about memory, not about the function."""
    if not is_hip_text_model(model):
        real = _real_peft()
        if real is None:
            raise ImportError("prepare_model_for_kbit_training: the model is not a bioreason_amd text model and peft is not installed")
        return real.prepare_model_for_kbit_training(model, use_gradient_checkpointing=use_gradient_checkpointing,
                                                    gradient_checkpointing_kwargs=gradient_checkpointing_kwargs)
    for p in model.parameters():
        p.requires_grad_(False)
    return model


def get_peft_model(model, peft_config, adapter_name: str = "default", **kwargs):
    """peft.get_peft_model(text_model, LoraConfig(...)) (`reason.py:388`, `train_dna_qwen.py:167`, `grpo_trainer.py:279`)"""
    if not is_hip_text_model(model):
        real = _real_peft()
        if real is None:
            raise ImportError("get_peft_model: the model is not a bioreason_amd text model and peft is not installed")
        return real.get_peft_model(model, peft_config, adapter_name=adapter_name, **kwargs)
    if adapter_name != "default":
        raise NotImplementedError("the HIP text model carries one adapter, named 'default' (as the reference's scripts do)")
    if getattr(model, "peft_config", None):
        raise ValueError("this text model already carries adapters (merge_and_unload() first, as reason.py:445 does)")
    bias = _cfg(peft_config, "bias", "none")
    if bias != "none":
        raise NotImplementedError(f"LoraConfig(bias={bias!r}): only 'none' (the scripts' value) is implemented")
    init = _cfg(peft_config, "init_lora_weights", True)
    if init not in (True, "gaussian"):
        raise NotImplementedError(f"LoraConfig(init_lora_weights={init!r}): True / 'gaussian' only")
    wanted = _cfg(peft_config, "target_modules", None)
    if wanted is None:
        targets = LORA_TARGETS
    else:
        wanted = [wanted] if isinstance(wanted, str) else list(wanted)
        # the scripts pass the last name component of EVERY nn.Linear of the text model plus attention patterns of other model
        # families ("out_proj", "query", ...: reason.py:331-357); what matches a Linear of Qwen3 is what peft would wrap
        targets = tuple(t for t in LORA_TARGETS if t in wanted)
        if not targets:
            raise ValueError(f"LoraConfig.target_modules {wanted} names no linear layer of the text model {LORA_TARGETS}")
    model.apply_lora(r=int(_cfg(peft_config, "r", 8)), alpha=float(_cfg(peft_config, "lora_alpha", 8)),
                     dropout=float(_cfg(peft_config, "lora_dropout", 0.0)), target_modules=targets)
    if init is True:
        # peft's default: A kaiming-uniform, B zero.  B = 0 makes the two initialisations the same function; A's distribution only
        # matters once training starts — keep apply_lora's gaussian (the value both scripts ask for) and say so
        pass
    model.peft_config = {adapter_name: peft_config}
    model.active_adapter = adapter_name
    model.base_model = _BaseModelView(model)
    return model


def print_trainable_parameters(model) -> None:
    tr = sum(p.numel() for p in model.parameters() if p.requires_grad)
    al = sum(p.numel() for p in model.parameters())
    print(f"trainable params: {tr:,d} || all params: {al:,d} || trainable%: {100 * tr / max(al, 1):.4f}")


class PeftModel:
    """`PeftModel.from_pretrained(text_model, adapter_dir, is_trainable=True)` (`reason.py:430-438`)"""

    @staticmethod
    def from_pretrained(model, model_id: str, adapter_name: str = "default", is_trainable: bool = False, **kwargs):
        if not is_hip_text_model(model):
            real = _real_peft()
            if real is None:
                raise ImportError("PeftModel.from_pretrained: the model is not a bioreason_amd text model and peft is not installed")
            return real.PeftModel.from_pretrained(model, model_id, adapter_name=adapter_name, is_trainable=is_trainable, **kwargs)
        from . import checkpoint
        with open(os.path.join(model_id, "adapter_config.json")) as fh:
            cfg = json.load(fh)
        get_peft_model(model, LoraConfig(r=cfg["r"], lora_alpha=cfg["lora_alpha"], lora_dropout=cfg.get("lora_dropout", 0.0),
                                         target_modules=cfg.get("target_modules"), init_lora_weights="gaussian"), adapter_name)
        checkpoint.load_adapter_dir(model, model_id)
        return model


def install_on(model_cls=Qwen3ForCausalLM) -> None:
    """the PEFT-model methods the scripts call on `model.text_model` after wrapping"""
    if not hasattr(model_cls, "print_trainable_parameters"):
        model_cls.print_trainable_parameters = print_trainable_parameters


install_on()


def refuse_foreign_wrapper(value) -> None:
    """`DNALLMModel.text_model = <something>`: a peft wrapper (or any module tree with LoRA layers) around a HIP text model that the
    engine would ignore must not be accepted silently"""
    if value is None or isinstance(value, Qwen3ForCausalLM) or not isinstance(value, nn.Module):
        return
    inner = getattr(getattr(value, "base_model", None), "model", None)
    names = [n for n, _ in value.named_modules()]
    looks_peft = type(value).__name__.startswith("Peft") or any(n.endswith("lora_A") or ".lora_A." in n for n in names)
    if looks_peft and (isinstance(inner, Qwen3ForCausalLM) or any(isinstance(m, Qwen3ForCausalLM) for m in value.modules())):
        raise TypeError(
            "DNALLMModel.text_model was given a PEFT wrapper around the HIP text model. The HIP engine executes its own adapters "
            "(Qwen3ForCausalLM.apply_lora) and would never run the wrapper's: they would train nothing. Use bioreason_amd.peft_compat."
            "get_peft_model (or put <repo>/compat first on PYTHONPATH so that `from peft import get_peft_model` resolves to it).")
