"""CPU ORACLE for the DNA-LLM hot path — TEST INFRASTRUCTURE, never the thing measured or shipped.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

What it is: a plain-PyTorch restatement of the reference's algorithm for the path
  DNA encoder -> dna_projection -> scatter over <|dna_pad|> rows -> Qwen3 causal LM (+ LoRA) -> loss / logits
  / generate, plus the GRPO arithmetic (oracle/grpo_math.py).
The arithmetic of the reference lives in its unpinned third-party dependency `transformers`
(installed here: 5.15.0); the oracle therefore CALLS the installed HF Qwen3 / ESM modules — exactly what
bioreason/models/dna_llm.py:64-66,79-81 instantiates — and restates only the reference's own glue:
  process_dna_embeddings  -> bioreason/models/dna_llm.py:103-179
  forward                 -> bioreason/models/dna_llm.py:181-244
  generate                -> bioreason/models/dna_llm.py:246-306
NT-v2's SwiGLU feed-forward comes from hub-hosted code that is not on disk; `make_nt_v2` restates it
(SURVEY §8c): intermediate.dense = Linear(H, 2F, bias=False), silu(x1) * x2, output.dense = Linear(F, H, bias=False).

Pinning: oracle/make_golden.py (run in the build container, where /root/reference exists) imports the
reference's UNMODIFIED DNALLMModel class, checks this restatement against it bit for bit on seeded inputs, and
writes tests/golden/*.pt; tests/test_oracle.py re-checks the restatement against those fixtures anywhere.
The reference itself ships no tests or golden vectors, so these fixtures are the pin.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- model builders
def make_qwen3(cfg: dict, attn_implementation: str = "eager"):
    """HF Qwen3ForCausalLM with random init (TF:models/qwen3/modeling_qwen3.py:448-507)."""
    from transformers import Qwen3Config, Qwen3ForCausalLM

    c = Qwen3Config(
        vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
        num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
        num_key_value_heads=cfg["num_key_value_heads"], head_dim=cfg["head_dim"],
        max_position_embeddings=cfg.get("max_position_embeddings", 4096), rms_norm_eps=cfg.get("rms_norm_eps", 1e-6),
        tie_word_embeddings=True, attention_bias=False,
        rope_parameters={"rope_type": "default", "rope_theta": cfg.get("rope_theta", 1e6)},
    )
    c._attn_implementation = attn_implementation
    return Qwen3ForCausalLM(c)


class _NTIntermediate(nn.Module):
    """NT-v2 hub EsmIntermediate: GLU with SiLU, no bias."""

    def __init__(self, hidden: int, inter: int):
        super().__init__()
        self.dense = nn.Linear(hidden, 2 * inter, bias=False)

    def forward(self, hidden_states):
        x = self.dense(hidden_states)
        x1, x2 = x.split(x.size(-1) // 2, dim=-1)
        return F.silu(x1) * x2


def make_nt_v2(cfg: dict, attn_implementation: str = "eager"):
    """HF EsmForMaskedLM shaped like InstaDeepAI/nucleotide-transformer-v2 (rotary, pre-LN, SwiGLU FFN)."""
    from transformers import EsmConfig, EsmForMaskedLM

    c = EsmConfig(
        vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
        num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
        max_position_embeddings=cfg.get("max_position_embeddings", 2050), position_embedding_type="rotary",
        pad_token_id=1, mask_token_id=2, token_dropout=False, emb_layer_norm_before=False,
        layer_norm_eps=1e-12, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
    )
    c._attn_implementation = attn_implementation
    m = EsmForMaskedLM(c)
    for layer in m.esm.encoder.layer:
        layer.intermediate = _NTIntermediate(c.hidden_size, c.intermediate_size)
        layer.output.dense = nn.Linear(c.intermediate_size, c.hidden_size, bias=False)
    for mod in m.modules():          # same init HF applies (normal(0, initializer_range))
        if isinstance(mod, nn.Linear):
            mod.weight.data.normal_(0.0, c.initializer_range)
    return m


class LoraLinear(nn.Module):
    """PEFT LoRA layer restated (peft is absent): y = W x + (alpha/r) * B(A(dropout(x))).
    Parameter names follow PEFT (base_layer / lora_A.default / lora_B.default)."""

    def __init__(self, base: nn.Linear, r: int, alpha: float, dropout: float = 0.0):
        super().__init__()
        self.base_layer = base
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, r, bias=False)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, base.out_features, bias=False)})
        self.scaling = alpha / r
        self.dropout = nn.Dropout(dropout) if dropout > 0 else nn.Identity()
        self.enabled = True
        nn.init.normal_(self.lora_A["default"].weight, std=1.0 / r)   # init_lora_weights="gaussian"
        nn.init.zeros_(self.lora_B["default"].weight)
        for p in base.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        y = self.base_layer(x)
        if self.enabled:
            a, b = self.lora_A["default"], self.lora_B["default"]
            y = y + (b(a(self.dropout(x).to(a.weight.dtype))) * self.scaling).to(y.dtype)
        return y


LORA_TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


def apply_lora(text_model, r: int = 32, alpha: float = 64.0, dropout: float = 0.0):
    """LoRA on every text-model linear except lm_head (train_dna_qwen.py:103-134,155-167)."""
    for layer in text_model.model.layers:
        for holder in (layer.self_attn, layer.mlp):
            for name in LORA_TARGETS:
                if hasattr(holder, name) and isinstance(getattr(holder, name), nn.Linear):
                    setattr(holder, name, LoraLinear(getattr(holder, name), r, alpha, dropout))
    for n, p in text_model.named_parameters():
        if "lora_" not in n:
            p.requires_grad_(False)
    return text_model


def set_adapters(text_model, enabled: bool):
    for m in text_model.modules():
        if isinstance(m, LoraLinear):
            m.enabled = enabled


# --------------------------------------------------------------------------- the reference's glue, restated
class OracleDNALLM(nn.Module):
    """Same attributes and call signatures as the reference DNALLMModel, built from given sub-modules."""

    def __init__(self, text_model, dna_model, dna_token_id: int):
        super().__init__()
        self.text_model = text_model
        self.dna_model = dna_model
        self.text_hidden_size = text_model.config.hidden_size
        self.dna_hidden_size = dna_model.config.hidden_size
        self.dna_projection = nn.Linear(self.dna_hidden_size, self.text_hidden_size)
        self.dna_token_id = dna_token_id

    # dna_llm.py:103-179 (HF-encoder branch)
    def process_dna_embeddings(self, dna_tokenized: Dict[str, torch.Tensor], batch_idx_map: List[int], batch_size: int):
        with torch.no_grad():        # dna_llm.py:121 — the encoder is frozen at run time whatever the flags say
            enc = self.dna_model(input_ids=dna_tokenized["input_ids"], attention_mask=dna_tokenized["attention_mask"],
                                 output_hidden_states=True)
            hidden = enc.hidden_states[-1]
        w = self.dna_projection.weight
        projected = self.dna_projection(hidden.to(device=w.device, dtype=w.dtype))
        lengths = dna_tokenized["attention_mask"].sum(dim=1).tolist()      # keeps the FIRST `len` rows (:168-169)
        per_sample: List[List[torch.Tensor]] = [[] for _ in range(batch_size)]
        for s, b in enumerate(batch_idx_map):
            per_sample[b].append(projected[s, : int(lengths[s])])
        out = []
        for chunks in per_sample:
            out.append(torch.cat(chunks, dim=0) if chunks else torch.zeros((0, self.text_hidden_size)))
        return out

    def _inputs_embeds(self, input_ids, dna_tokenized, batch_idx_map):
        embeds = self.text_model.get_input_embeddings()(input_ids)              # :211
        if dna_tokenized is not None and batch_idx_map:
            rows = torch.cat(self.process_dna_embeddings(dna_tokenized, batch_idx_map, input_ids.shape[0]), dim=0)
            where = input_ids == self.dna_token_id                             # :216
            n_tok, n_feat = int(where.sum().item()), rows.shape[0]
            if n_tok != n_feat:                                                # :222-225
                raise ValueError(f"DNA features and DNA tokens do not match: features {n_feat}, tokens: {n_tok}")
            embeds[where] = rows.to(embeds.dtype)                              # :228-229 (row-major order)
        return embeds

    # dna_llm.py:181-244
    def forward(self, input_ids=None, attention_mask=None, dna_tokenized=None, batch_idx_map=None, labels=None, **kw):
        if input_ids is None or attention_mask is None:
            raise ValueError("Either 'inputs' or 'input_ids'/'attention_mask' must be provided")
        embeds = self._inputs_embeds(input_ids, dna_tokenized, batch_idx_map)
        return self.text_model(inputs_embeds=embeds, attention_mask=attention_mask, labels=labels, **kw)

    # dna_llm.py:246-306
    def generate(self, input_ids=None, attention_mask=None, dna_tokenized=None, batch_idx_map=None, **gen_kw):
        if input_ids is None or attention_mask is None:
            raise ValueError("Either 'inputs' or 'input_ids'/'attention_mask' must be provided")
        embeds = self._inputs_embeds(input_ids, dna_tokenized, batch_idx_map)
        with torch.no_grad():
            return self.text_model.generate(inputs_embeds=embeds, attention_mask=attention_mask, use_cache=True, **gen_kw)


# --------------------------------------------------------------------------- synthetic inputs (SURVEY §8d shape)
def synth_batch(seed: int, B: int, n_dna_per_sample: int, Sd: int, text_len: int, vocab_text: int, vocab_dna: int,
                dna_token_id: int, left_pad: Optional[List[int]] = None, dna_pad: Optional[Dict[int, int]] = None,
                label_tail: int = 8):
    """DNA+prompt batch: every sample = `n_dna_per_sample` DNA sequences of Sd tokens (optionally right-padded with
    id 1) and a text prompt of `text_len` ordinary tokens with one run of <|dna_pad|> per sequence; optional left
    padding of the text (attention_mask = 0), labels = -100 except the last `label_tail` positions."""
    g = torch.Generator().manual_seed(seed)
    nseq = B * n_dna_per_sample
    dna_ids = torch.randint(6, vocab_dna, (nseq, Sd), generator=g)
    dna_ids[:, 0] = 3                                           # <cls>
    dna_mask = torch.ones(nseq, Sd, dtype=torch.long)
    for s, keep in (dna_pad or {}).items():
        dna_ids[s, keep:] = 1
        dna_mask[s, keep:] = 0
    batch_idx_map = [b for b in range(B) for _ in range(n_dna_per_sample)]
    valid = dna_mask.sum(1).tolist()
    rows = []
    for b in range(B):
        toks = torch.randint(0, dna_token_id - 3, (text_len,), generator=g).tolist()
        head, tail = toks[: min(8, text_len // 2)], toks[min(8, text_len // 2):]
        mid = []
        for s in range(nseq):
            if batch_idx_map[s] == b:
                mid += [dna_token_id - 1] + [dna_token_id] * int(valid[s]) + [dna_token_id + 1]
        rows.append(head + mid + tail)
    lp = left_pad or [0] * B
    P = max(len(r) + lp[b] for b, r in enumerate(rows))
    ids = torch.zeros(B, P, dtype=torch.long)
    mask = torch.zeros(B, P, dtype=torch.long)
    for b, r in enumerate(rows):
        ids[b, P - len(r):] = torch.tensor(r)
        mask[b, P - len(r):] = 1
    labels = torch.full((B, P), -100, dtype=torch.long)
    labels[:, P - label_tail:] = ids[:, P - label_tail:]
    return {"input_ids": ids, "attention_mask": mask, "labels": labels,
            "dna_tokenized": {"input_ids": dna_ids, "attention_mask": dna_mask}, "batch_idx_map": batch_idx_map}
