"""Checkpoint interchange with the reference.

* `load_pretrained_pair`  — what `DNALLMModel.__init__` does with two model names (dna_llm.py:62-84): here the names must
  be LOCAL directories (`config.json` + `*.safetensors` / `pytorch_model.bin`), the Hub being unreachable.
* `reference_state_dict` / `save_checkpoint` — the `pytorch_model.bin` the reference's `SaveWithPyTorchCallback` writes
  (reason.py:46-81): `model.state_dict()` of a `DNALLMModel` whose text model is PEFT-wrapped, i.e. keys
  `text_model.base_model.model.<hf name>` with `base_layer` / `lora_A.default` / `lora_B.default`, `dna_model.*`,
  `dna_projection.*`.
* `load_sft_checkpoint` — the three input forms `reason.py:422-540` accepts: a PEFT adapter directory (loaded, then merged
  into the base weights as `merge_and_unload` does), a `{"state_dict": ...}` (Lightning, `train_dna_qwen.py:963-970`) /
  `{"module": ...}` (DeepSpeed) / plain state-dict file with or without LoRA keys, with the reference's key re-mapping.

Everything here is host-side bookkeeping over named tensors; the arithmetic (the merge W + s B A) runs on the device
through the same GEMM the rollout uses."""
import json
import os
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch

PEFT_PREFIX = "text_model.base_model.model."
NEW_TOKENS = ["<|dna_start|>", "<|dna_pad|>", "<|dna_end|>"]          # dna_llm.py:72


# ----------------------------------------------------------------------------------------------- reading weight files
def read_weight_dir(path: str) -> Dict[str, torch.Tensor]:
    """every tensor of a HF model directory: model.safetensors, sharded *.safetensors (+ index), or pytorch_model.bin"""
    if not os.path.isdir(path):
        raise RuntimeError(f"bioreason_amd: '{path}' is not a local checkpoint directory and the HF Hub is unreachable here. "
                           "Pass config objects (bioreason_amd.configs) for random-init models, or a local directory with "
                           "config.json + *.safetensors.")
    out: Dict[str, torch.Tensor] = {}
    idx = os.path.join(path, "model.safetensors.index.json")
    files = []
    if os.path.exists(idx):
        with open(idx) as fh:
            files = sorted(set(json.load(fh)["weight_map"].values()))
    else:
        files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors") and not f.startswith("adapter_"))
    if files:
        from safetensors.torch import load_file
        for f in files:
            out.update(load_file(os.path.join(path, f)))
        return out
    binf = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(binf):
        return dict(torch.load(binf, map_location="cpu", weights_only=True))
    raise RuntimeError(f"bioreason_amd: no *.safetensors or pytorch_model.bin under '{path}'")


def _load_config(path: str):
    with open(os.path.join(path, "config.json")) as fh:
        return json.load(fh)


def _tokenizer(path: str):
    # transformers 5 builds an EMPTY tokenizer for a directory that only holds config.json: require the files
    if not any(os.path.exists(os.path.join(path, f)) for f in ("tokenizer.json", "vocab.json", "vocab.txt", "tokenizer.model")):
        return None
    try:
        from transformers import AutoTokenizer
        return AutoTokenizer.from_pretrained(path, trust_remote_code=False, local_files_only=True)
    except Exception:
        return None                                   # weights-only directory: the caller tokenises elsewhere


def _require_dir(n):
    if not (isinstance(n, str) and os.path.isdir(n)):
        raise RuntimeError(
            f"bioreason_amd: '{n}' is not a local checkpoint directory and the HF Hub is unreachable here. "
            "Pass config objects (bioreason_amd.configs) for random-init models, or a local directory with "
            "config.json + *.safetensors.")


def load_pretrained_text(text_model_name, cache_dir, device):
    """-> (Qwen3ForCausalLM with the directory's weights, its tokenizer prepared as dna_llm.py:67-73 does, or None)"""
    from . import configs
    from .modeling import Qwen3ForCausalLM
    _require_dir(text_model_name)
    tc = _load_config(text_model_name)
    tkeys = ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads",
             "head_dim", "rope_theta", "max_position_embeddings", "rms_norm_eps")
    text = Qwen3ForCausalLM(configs.qwen3_config(**{k: tc[k] for k in tkeys if k in tc}), device=device)
    sd = read_weight_dir(text_model_name)
    missing, unexpected = text.load_state_dict(sd, strict=False)
    missing = [k for k in missing if k != "lm_head.weight"]           # tied: absent from HF Qwen3 checkpoints
    if missing:
        raise RuntimeError(f"bioreason_amd: '{text_model_name}' lacks {len(missing)} tensors, e.g. {missing[:3]}")
    text.tie_weights()
    tt = _tokenizer(text_model_name)
    if tt is not None:                                                # dna_llm.py:68-73
        from .chat_template import CHAT_TEMPLATE
        tt.pad_token = tt.eos_token
        tt.chat_template = CHAT_TEMPLATE                              # dna_llm.py:69: {"type": "dna"} items -> placeholders
        tt.add_special_tokens({"additional_special_tokens": NEW_TOKENS})
    return text, tt


def check_nt_v2_key_shapes(sd: Dict[str, torch.Tensor], hidden: int, intermediate: int, n_layers: int, where: str = "") -> None:
    """The NT-v2 feed-forward of this package is RECALLED from the public hub file (gated SwiGLU without bias: one
    `intermediate.dense.weight` of shape [2F, H] whose halves are gate | up, `output.dense.weight` [H, F]; SURVEY §8c asks for
    exactly this check once weights exist).  A checkpoint whose tensors say otherwise must fail HERE, loudly, not run as a
    different network: plain ESM has [F, H] + biases, which `load_state_dict(strict=False)` would otherwise half-accept."""
    F, H = int(intermediate), int(hidden)
    problems = []
    for layer in range(n_layers):
        base = f"esm.encoder.layer.{layer}."
        for key, want in ((base + "intermediate.dense.weight", (2 * F, H)), (base + "output.dense.weight", (H, F))):
            t = sd.get(key)
            if t is None:
                problems.append(f"{key}: missing")
            elif tuple(t.shape) != want:
                problems.append(f"{key}: {tuple(t.shape)} instead of {want}")
        for key in (base + "intermediate.dense.bias", base + "output.dense.bias"):
            if key in sd:
                problems.append(f"{key}: present, but the gated FFN of NT-v2 has no bias (config add_bias_fnn=false)")
    if problems:
        raise RuntimeError(f"bioreason_amd: '{where}' is not an NT-v2 checkpoint as this package models it "
                           f"({len(problems)} feed-forward tensors disagree, e.g. {problems[:3]}); the encoder's gated FFN was "
                           "restated from the public hub code without a checkpoint to verify against (SURVEY §8c)")


def load_pretrained_dna(dna_model_name, cache_dir, device):
    """-> (NTEncoderForMaskedLM with the directory's weights, its tokenizer or None)"""
    from . import configs
    from .modeling import NTEncoderForMaskedLM
    _require_dir(dna_model_name)
    dc = _load_config(dna_model_name)
    dkeys = ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "max_position_embeddings")
    cfg = configs.nt_v2_config(**{k: dc[k] for k in dkeys if k in dc})
    dna = NTEncoderForMaskedLM(cfg, device=device)
    sd = read_weight_dir(dna_model_name)
    check_nt_v2_key_shapes(sd, cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, dna_model_name)
    # the hub NT-v2 names equal the installed ESM's except for the gated FFN, which this package names as the oracle does
    missing, unexpected = dna.load_state_dict(sd, strict=False)
    if missing:
        raise RuntimeError(f"bioreason_amd: '{dna_model_name}' lacks {len(missing)} tensors, e.g. {missing[:3]} "
                           "(NT-v2 key names could not be verified offline: SURVEY §8c)")
    return dna, _tokenizer(dna_model_name)


def load_pretrained_pair(text_model_name, dna_model_name, cache_dir, device):
    """-> (text_model, dna_model, (text_tokenizer, dna_tokenizer, processor)) from two local directories"""
    for n in (text_model_name, dna_model_name):
        _require_dir(n)
    text, tt = load_pretrained_text(text_model_name, cache_dir, device)
    dna, dt = load_pretrained_dna(dna_model_name, cache_dir, device)
    processor = None
    if tt is not None and dt is not None:                             # dna_llm.py:100: DLProcessor(tokenizer, dna_tokenizer)
        from .processing import DLProcessor
        processor = DLProcessor(tokenizer=tt, dna_tokenizer=dt)
    return text, dna, (tt, dt, processor)


# ----------------------------------------------------------------------------------------------- reference key names
def reference_state_dict(model) -> "OrderedDict[str, torch.Tensor]":
    """`DNALLMModel.state_dict()` under the reference's names: the text model PEFT-wrapped when adapters exist"""
    has_lora = any("lora_" in k for k in model.text_model.state_dict())
    out = OrderedDict()
    for k, v in model.state_dict().items():
        if k.startswith("text_model.") and has_lora:
            k = PEFT_PREFIX + k[len("text_model."):]
        out[k] = v.detach()
    return out


def save_checkpoint(model, folder: str) -> str:
    """reason.py:46-81: `<folder>/pytorch_model.bin` (torch.save of the state dict) + the text model's config.json"""
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, "pytorch_model.bin")
    torch.save(OrderedDict((k, v.cpu().clone()) for k, v in reference_state_dict(model).items()), path)
    cfg = model.text_model.config
    with open(os.path.join(folder, "config.json"), "w") as fh:
        json.dump(cfg.to_dict() if hasattr(cfg, "to_dict") else dict(vars(cfg)), fh, indent=1, default=str)
    return path


def _strip(k: str) -> str:
    """reason.py:455-458 strips "=model." (sic) and "_forward_module."; DeepSpeed adds "module.", and Lightning's
    DNALLMFineTuner keeps the DNALLMModel under its attribute `model` (train_dna_qwen.py:963-970)"""
    changed = True
    while changed:
        changed = False
        for pre in ("_forward_module.", "module.", "=model."):
            if k.startswith(pre):
                k, changed = k[len(pre):], True
    if k.startswith("model.") and k[len("model."):].startswith(("text_model.", "dna_model.", "dna_projection.")):
        k = k[len("model."):]
    return k


def _to_own_key(k: str, own: set) -> str:
    if k in own:
        return k
    if k.startswith(PEFT_PREFIX):                                     # reason.py:492-496
        c = "text_model." + k[len(PEFT_PREFIX):]
        if c in own:
            return c
        c2 = c.replace(".base_layer.", ".")                           # PEFT-wrapped file into a model without adapters
        if c2 in own:
            return c2
    if k.startswith("text_model."):                                   # plain file into an adapter-carrying model
        parts = k.rsplit(".", 1)
        c = parts[0] + ".base_layer." + parts[1]
        if c in own:
            return c
    return k


def load_state_dict_tensors(model, tensors: Dict[str, torch.Tensor]) -> Tuple[list, list]:
    """copy every tensor whose (re-mapped) name the model owns; -> (missing, unexpected) like load_state_dict(strict=False)"""
    own = dict(model.state_dict())
    seen = set()
    unexpected = []
    for k, v in tensors.items():
        kk = _to_own_key(_strip(k), set(own))
        if kk not in own:
            unexpected.append(k)
            continue
        dst = own[kk]
        if tuple(dst.shape) != tuple(v.shape):
            raise RuntimeError(f"bioreason_amd: shape mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(dst.shape)}")
        dst.copy_(v.to(device=dst.device, dtype=dst.dtype))
        seen.add(kk)
    if model.arena is not None:
        model.arena.pack()
    skip = ("lm_head.weight",)                                         # tied to embed_tokens
    missing = [k for k in own if k not in seen and not k.endswith(skip)]
    return missing, unexpected


def load_adapter_dir(text_model, path: str) -> Tuple[list, list]:
    """the tensors of a PEFT adapter directory (`adapter_model.safetensors` / `.bin`: `base_model.model.<hf name>.lora_A.weight`, adapter
    name elided) into a text model that already carries adapters of that shape (`PeftModel.from_pretrained`, reason.py:430-438);
    -> (missing, unexpected)"""
    f = os.path.join(path, "adapter_model.safetensors")
    if os.path.exists(f):
        from safetensors.torch import load_file
        tensors = load_file(f)
    else:
        tensors = torch.load(os.path.join(path, "adapter_model.bin"), map_location="cpu", weights_only=True)
    own = {k: v for k, v in text_model.state_dict().items() if "lora_" in k}
    seen, unexpected = set(), []
    for k, v in tensors.items():
        kk = k.replace(".lora_A.weight", ".lora_A.default.weight").replace(".lora_B.weight", ".lora_B.default.weight")
        if kk.startswith("base_model.model."):
            kk = kk[len("base_model.model."):]
        if kk not in own:
            unexpected.append(k)
            continue
        if tuple(own[kk].shape) != tuple(v.shape):
            raise RuntimeError(f"bioreason_amd: shape mismatch for {k}: adapter {tuple(v.shape)} vs model {tuple(own[kk].shape)}")
        own[kk].copy_(v.to(device=own[kk].device, dtype=own[kk].dtype))
        seen.add(kk)
    if text_model.arena is not None:
        text_model.arena.pack()
    return [k for k in own if k not in seen], unexpected


def _infer_lora_r(tensors) -> Optional[int]:
    for k, v in tensors.items():
        if "lora_A" in k:
            return int(v.shape[0])
    return None


def _torch_load(path: str, trust_checkpoint: bool):
    """weights_only first; genuine Lightning / DeepSpeed files also pickle non-tensor objects (hyper-parameter Namespaces,
    callback and optimiser state) that the safe unpickler refuses — those need the caller's explicit `trust_checkpoint=True`
    (arbitrary-code unpickling, exactly what the reference's `torch.load(path)` does, reason.py:449)."""
    import pickle
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except (pickle.UnpicklingError, RuntimeError, AttributeError) as e:
        # only a refusal of the SAFE unpickler (non-tensor globals in the file) is an invitation to trust_checkpoint; a truncated
        # or corrupt file must not steer the user towards arbitrary-code unpickling
        refused = isinstance(e, pickle.UnpicklingError) or "weights_only" in str(e).lower() or "unsupported global" in str(e).lower()
        if not refused:
            raise
        if not trust_checkpoint:
            raise RuntimeError(f"bioreason_amd: '{path}' holds pickled objects besides tensors (a Lightning / DeepSpeed checkpoint). "
                               "Pass trust_checkpoint=True to load it with the full unpickler if you trust the file.") from e
        return torch.load(path, map_location="cpu", weights_only=False)


def load_sft_checkpoint(model, path: str, lora_alpha: float = 64.0, lora_dropout: float = 0.05,
                        trust_checkpoint: bool = False) -> Tuple[list, list]:
    """reason.py:422-540.  Directory = PEFT adapter (adapter_config.json + adapter_model.*): adapters are loaded and merged
    into the base weights.  File = torch.save'd dict in one of the three layouts; LoRA keys present -> adapters are created
    (if absent) and filled; absent -> base weights are loaded and fresh adapters are left to the caller."""
    if os.path.isdir(path):
        with open(os.path.join(path, "adapter_config.json")) as fh:
            ac = json.load(fh)
        f = os.path.join(path, "adapter_model.safetensors")
        if os.path.exists(f):
            from safetensors.torch import load_file
            tensors = load_file(f)
        else:
            tensors = torch.load(os.path.join(path, "adapter_model.bin"), map_location="cpu", weights_only=True)
        r, alpha = int(ac.get("r", _infer_lora_r(tensors) or 32)), float(ac.get("lora_alpha", lora_alpha))
        if not any("lora_" in k for k in model.text_model.state_dict()):
            kw = {"target_modules": tuple(ac["target_modules"])} if ac.get("target_modules") else {}
            model.text_model.apply_lora(r=r, alpha=alpha, dropout=float(ac.get("lora_dropout", 0.0)), arena=model.arena, **kw)
        else:                                                         # the merge below uses the existing groups' r / alpha
            for L in model.text_model.ensure_packed().layers:
                for G in L.lora.values():
                    if G is not None and (G.r != r or abs(G.scaling - alpha / r) > 1e-9):
                        raise ValueError(f"adapter_config.json says r={r}, lora_alpha={alpha} but the model's adapters were "
                                         f"created with r={G.r}, scaling={G.scaling}")
        # PEFT saves `base_model.model.<hf name>.lora_A.weight` (adapter name elided)
        renamed = {}
        for k, v in tensors.items():
            k = k.replace(".lora_A.weight", ".lora_A.default.weight").replace(".lora_B.weight", ".lora_B.default.weight")
            renamed["text_model." + k if k.startswith("base_model.model.") else k] = v
        res = load_state_dict_tensors(model, renamed)
        model.text_model.merge_and_unload()                           # reason.py:441-444
        return res
    ck = _torch_load(path, trust_checkpoint)
    if "state_dict" in ck:
        tensors = ck["state_dict"]
    elif "module" in ck:
        tensors = ck["module"]
    elif isinstance(ck, dict) and all(isinstance(k, str) for k in ck):
        tensors = ck
    else:
        raise ValueError(f"Unsupported checkpoint format: {path}")
    if any("lora" in k for k in tensors):
        if not any("lora_" in k for k in model.text_model.state_dict()):
            model.text_model.apply_lora(r=_infer_lora_r(tensors) or 32, alpha=lora_alpha, dropout=lora_dropout, arena=model.arena)
    return load_state_dict_tensors(model, tensors)
