"""nn.Module shells with the reference's (HF / PEFT) module and parameter names around the HIP engines.

The reference instantiates `AutoModelForCausalLM` (Qwen3) and `AutoModelForMaskedLM` (NT-v2 ESM)
(bioreason/models/dna_llm.py:64-66, 79-81) and wraps the text model with PEFT-LoRA
(train_dna_qwen.py:155-167, reason.py:376-388).  These classes keep the same attribute tree — so
`state_dict()` keys, `named_modules()` scans for `nn.Linear` LoRA targets and checkpoints interchange —
but their arithmetic is the kernel library.  Nothing here calls torch math on the hot path.
"""
from __future__ import annotations

import contextlib
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .arena import TrainableArena
from .engine import BF16, EsmEngine, LoraGroup, QwenEngine, SeqMeta

LORA_TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


# ---------------------------------------------------------------------------------------------- leaf shells
class HipLinear(nn.Linear):
    """Parameter container with nn.Linear's interface; standalone calls run the MFMA GEMM."""

    def forward(self, x):
        w = self.weight if self.weight.dtype == BF16 else self.weight.to(BF16)
        b = None if self.bias is None else self.bias.to(BF16)
        y = ops.gemm_nt(x.reshape(-1, x.shape[-1]).to(BF16).contiguous(), w.contiguous(), bias=b)
        return y.view(*x.shape[:-1], self.out_features)


class HipRMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-6, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype))
        self.variance_epsilon = eps

    def forward(self, x):
        return ops.rmsnorm_fwd(x.contiguous(), self.weight.to(BF16), self.variance_epsilon)


class LoraLinear(nn.Module):
    """PEFT-named LoRA wrapper: base_layer + lora_A.default / lora_B.default (views into the TrainableArena)."""

    def __init__(self, base: HipLinear, group: LoraGroup, j: int, r: int, alpha: float, dropout: float):
        super().__init__()
        self.base_layer = base
        self.in_features, self.out_features = base.in_features, base.out_features
        self.r, self.lora_alpha, self.scaling = r, alpha, alpha / r
        self.lora_dropout_p = dropout
        self.lora_A = nn.ModuleDict({"default": nn.Module()})
        self.lora_B = nn.ModuleDict({"default": nn.Module()})
        self._group, self._j = group, j
        group.on_views.append(self._bind)

    def _bind(self):
        A, B, gA, gB = self._group.target_views(self._j)
        a_mod, b_mod = self.lora_A["default"], self.lora_B["default"]
        if "weight" in a_mod._parameters:
            a_mod.weight.data, b_mod.weight.data = A, B
        else:
            a_mod.weight, b_mod.weight = nn.Parameter(A), nn.Parameter(B)
        a_mod.weight.grad, b_mod.weight.grad = gA, gB

    @property
    def weight(self):
        return self.base_layer.weight


# ---------------------------------------------------------------------------------------------- Qwen3
class Qwen3Attention(nn.Module):
    def __init__(self, c, dev, dt):
        super().__init__()
        H, hd = c.hidden_size, c.head_dim
        self.q_proj = HipLinear(H, c.num_attention_heads * hd, bias=False, device=dev, dtype=dt)
        self.k_proj = HipLinear(H, c.num_key_value_heads * hd, bias=False, device=dev, dtype=dt)
        self.v_proj = HipLinear(H, c.num_key_value_heads * hd, bias=False, device=dev, dtype=dt)
        self.o_proj = HipLinear(c.num_attention_heads * hd, H, bias=False, device=dev, dtype=dt)
        self.q_norm = HipRMSNorm(hd, c.rms_norm_eps, dev, dt)
        self.k_norm = HipRMSNorm(hd, c.rms_norm_eps, dev, dt)


class Qwen3MLP(nn.Module):
    def __init__(self, c, dev, dt):
        super().__init__()
        self.gate_proj = HipLinear(c.hidden_size, c.intermediate_size, bias=False, device=dev, dtype=dt)
        self.up_proj = HipLinear(c.hidden_size, c.intermediate_size, bias=False, device=dev, dtype=dt)
        self.down_proj = HipLinear(c.intermediate_size, c.hidden_size, bias=False, device=dev, dtype=dt)


class Qwen3DecoderLayer(nn.Module):
    def __init__(self, c, dev, dt):
        super().__init__()
        self.self_attn = Qwen3Attention(c, dev, dt)
        self.mlp = Qwen3MLP(c, dev, dt)
        self.input_layernorm = HipRMSNorm(c.hidden_size, c.rms_norm_eps, dev, dt)
        self.post_attention_layernorm = HipRMSNorm(c.hidden_size, c.rms_norm_eps, dev, dt)


class Qwen3Model(nn.Module):
    def __init__(self, c, dev, dt):
        super().__init__()
        self.embed_tokens = nn.Embedding(c.vocab_size, c.hidden_size, device=dev, dtype=dt)
        self.layers = nn.ModuleList([Qwen3DecoderLayer(c, dev, dt) for _ in range(c.num_hidden_layers)])
        self.norm = HipRMSNorm(c.hidden_size, c.rms_norm_eps, dev, dt)


def _unwrap(mod):
    return mod.base_layer if isinstance(mod, LoraLinear) else mod


class _StackFn(torch.autograd.Function):
    """The whole decoder stack as one autograd node: forward keeps the activation tape, backward runs the
    hand-written layer backward (engine.layer_bwd) and deposits LoRA gradients in the arena."""

    @staticmethod
    def forward(ctx, x, anchor, model, meta, grad_mode=True):
        eng = model.engine
        # `needs_input_grad` reflects the inputs' requires_grad, NOT the caller's grad mode (and grad mode is always off inside a Function's
        # forward): the caller passes torch.is_grad_enabled() — a no-grad pass keeps no tape and may take the no-grad kernels
        # (fused SwiGLU epilogue, the fp8 path under engine.use_fp8)
        need = bool(grad_mode and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]))
        hid, tape = eng.forward_hidden(x, meta, save=need)
        ctx.model, ctx.meta, ctx.tape = model, meta, tape
        return hid

    @staticmethod
    def backward(ctx, dhid):
        eng = ctx.model.engine
        eng.ensure_transposed()
        dx = eng.backward_hidden(dhid.contiguous(), ctx.tape, ctx.meta)
        ctx.tape = None
        return dx, None, None, None, None


class _SharedStackFn(torch.autograd.Function):
    """The decoder stack over (distinct prompts, per-copy completions) as one autograd node (engine.forward_hidden_shared)."""

    @staticmethod
    def forward(ctx, xp, xc, anchor, model, mp, mc, copies, side, grad_mode=True):
        eng = model.engine
        need = bool(grad_mode and (ctx.needs_input_grad[0] or ctx.needs_input_grad[2]))
        hid_last, hid_c, tape = eng.forward_hidden_shared(xp, mp, xc, mc, copies, save=need, side=side)
        ctx.model, ctx.mp, ctx.mc, ctx.copies, ctx.tape, ctx.side = model, mp, mc, copies, tape, side
        return hid_last, hid_c

    @staticmethod
    def backward(ctx, dlast, dc):
        eng = ctx.model.engine
        eng.ensure_transposed()
        dxp, dxc = eng.backward_hidden_shared(dlast.contiguous(), dc.contiguous(), ctx.tape, ctx.mp, ctx.mc, ctx.copies, side=ctx.side)
        ctx.tape = None
        return dxp, None, None, None, None, None, None, None, None


class _ExpandGroupsFn(torch.autograd.Function):
    """row r -> `copies` consecutive copies of it; backward = sum over the copies of each group (bra_group_sum)"""

    @staticmethod
    def forward(ctx, x, copies):
        ctx.copies = copies
        return x.repeat_interleave(copies, dim=0)

    @staticmethod
    def backward(ctx, dy):
        return ops.group_sum(dy.contiguous(), ctx.copies), None


class _LogProbFn(torch.autograd.Function):
    """log p(target | hidden row) through the tied lm_head, fused (no [rows, V] logits in HBM on the forward)."""

    @staticmethod
    def forward(ctx, h, model, tgt):
        eng = model.engine
        logp, lse = ops.lmhead_logprob(h, eng.E, tgt)
        ctx.save_for_backward(h, tgt, lse)
        ctx.model = model
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        h, tgt, lse = ctx.saved_tensors
        eng = ctx.model.engine
        eng.ensure_transposed()
        dlogits = ops.lmhead_dlogits(h, eng.E, tgt, lse, dlogp.contiguous().float())
        dh = ops.gemm_nt(dlogits, eng.ET[:, :eng.V] if eng.ET.shape[1] != eng.V else eng.ET)
        return dh, None, None


class _LogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, model):
        ctx.model = model
        return ops.gemm_nt(h, model.engine.E)

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.model.engine
        eng.ensure_transposed()
        return ops.gemm_nt(dlogits.contiguous().to(BF16), eng.ET), None


class Qwen3ForCausalLM(nn.Module):
    """HIP-backed stand-in for transformers.Qwen3ForCausalLM (TF:models/qwen3/modeling_qwen3.py:430-507)."""

    def __init__(self, config, device=None, dtype=BF16):
        super().__init__()
        self.config = config
        self.model = Qwen3Model(config, device, dtype)
        self.lm_head = HipLinear(config.hidden_size, config.vocab_size, bias=False, device=device, dtype=dtype)
        self.lm_head.weight = self.model.embed_tokens.weight          # tie_word_embeddings (all Qwen3 sizes used)
        self.engine: Optional[QwenEngine] = None
        self.arena: Optional[TrainableArena] = None
        self._packed_sig = None
        self._lora_enabled = True
        self.warnings_issued: Dict[str, bool] = {}
        self.generation_config = SimpleNamespace()

    # ---- HF surface used by the reference -----------------------------------------------------------------
    def get_input_embeddings(self):
        return self.model.embed_tokens

    def gradient_checkpointing_enable(self, *a, **k):   # activations are kept (288 GB HBM): nothing to do
        return None

    def enable_input_require_grads(self):
        return None

    def tie_weights(self):
        self.lm_head.weight = self.model.embed_tokens.weight

    @property
    def device(self):
        return self.model.embed_tokens.weight.device

    @torch.no_grad()
    def init_weights(self, std: float = 0.02, seed: int = 0):
        g = torch.Generator(device=self.device).manual_seed(seed)
        for n, p in self.named_parameters():
            if "lora_" in n:
                continue
            if p.dim() >= 2:
                p.normal_(0.0, std, generator=g)
            else:
                p.fill_(1.0)
        self._packed_sig = None

    # ---- packing --------------------------------------------------------------------------------------------
    def _base_params(self):
        out = [self.model.embed_tokens.weight, self.model.norm.weight]
        for l in self.model.layers:
            a, m = l.self_attn, l.mlp
            out += [_unwrap(a.q_proj).weight, _unwrap(a.k_proj).weight, _unwrap(a.v_proj).weight, _unwrap(a.o_proj).weight,
                    a.q_norm.weight, a.k_norm.weight, _unwrap(m.gate_proj).weight, _unwrap(m.up_proj).weight,
                    _unwrap(m.down_proj).weight, l.input_layernorm.weight, l.post_attention_layernorm.weight]
        return out

    def ensure_packed(self):
        """(Re)build the fused bf16 weight images when a base parameter changed identity / version / device."""
        if self.arena is not None and self.arena.params is not None:
            self.arena.pack_if_stale()
        params = self._base_params()
        sig = tuple((p.data_ptr(), p._version, str(p.device), p.dtype) for p in params)
        if sig == self._packed_sig:
            return self.engine
        dev = self.device
        c = self.config
        eng = self.engine
        if eng is None or eng.device != dev:
            old = eng
            eng = QwenEngine(c, dev)
            if old is not None:
                for ln, lo in zip(eng.layers, old.layers):
                    ln.lora = lo.lora
            self.engine = eng

        def bf(p):
            return p.data if p.dtype == BF16 else p.data.to(BF16)

        eng.E = bf(self.model.embed_tokens.weight).contiguous()
        eng.ET = None
        eng.norm_w = bf(self.model.norm.weight)
        for l, L in zip(self.model.layers, eng.layers):
            a, m = l.self_attn, l.mlp
            qs, ks, vs = (_unwrap(a.q_proj).weight, _unwrap(a.k_proj).weight, _unwrap(a.v_proj).weight)
            L.Wqkv = torch.cat([bf(qs), bf(ks), bf(vs)], dim=0).contiguous()
            gs, us = _unwrap(m.gate_proj).weight, _unwrap(m.up_proj).weight
            L.Wgu = torch.cat([bf(gs), bf(us)], dim=0).contiguous()
            L.Wo = bf(_unwrap(a.o_proj).weight).contiguous()
            L.Wd = bf(_unwrap(m.down_proj).weight).contiguous()
            L.WqkvT = L.WoT = L.WguT = L.WdT = None
            # the fused buffers become the parameters' storage (no second copy of the base weights)
            if qs.dtype == BF16:
                nq, nk = qs.shape[0], ks.shape[0]
                qs.data, ks.data, vs.data = L.Wqkv[:nq], L.Wqkv[nq:nq + nk], L.Wqkv[nq + nk:]
                gs.data, us.data = L.Wgu[:gs.shape[0]], L.Wgu[gs.shape[0]:]
            L.ln1, L.ln2 = bf(l.input_layernorm.weight), bf(l.post_attention_layernorm.weight)
            L.qn, L.kn = bf(a.q_norm.weight), bf(a.k_norm.weight)
        params = self._base_params()
        self._packed_sig = tuple((p.data_ptr(), p._version, str(p.device), p.dtype) for p in params)
        return eng

    # ---- LoRA ------------------------------------------------------------------------------------------------
    def apply_lora(self, r: int = 32, alpha: float = 64.0, dropout: float = 0.0, target_modules=LORA_TARGETS,
                   arena: Optional[TrainableArena] = None, init_seed: int = 0):
        """PEFT get_peft_model(text_model, LoraConfig(r, lora_alpha, lora_dropout, target_modules,
        init_lora_weights="gaussian")) on the text model (train_dna_qwen.py:155-167)."""
        if dropout < 0.0 or dropout >= 1.0:
            raise ValueError("lora_dropout must be in [0, 1)")
        if dropout > 0.0 and r != 32:
            raise NotImplementedError("lora_dropout > 0 needs r = 32 in the HIP path (one mask stream per 32-column rank block)")
        self.lora_dropout_p = float(dropout)      # active in training mode with adapters enabled (nn.Dropout of PEFT's LoraLayer)
        dev = self.device
        self.arena = arena or self.arena or TrainableArena(dev)
        self.ensure_packed()
        c = self.config
        groups = []
        for li, (l, L) in enumerate(zip(self.model.layers, self.engine.layers)):
            a, m = l.self_attn, l.mlp
            spec = {"qkv": (a, ["q_proj", "k_proj", "v_proj"], c.hidden_size),
                    "o": (a, ["o_proj"], c.num_attention_heads * c.head_dim),
                    "gu": (m, ["gate_proj", "up_proj"], c.hidden_size),
                    "d": (m, ["down_proj"], c.intermediate_size)}
            for gname, (holder, names, K) in spec.items():
                if not any(n in target_modules for n in names):
                    continue
                sizes = [getattr(holder, n).out_features for n in names]
                G = LoraGroup(self.arena, f"text.layers.{li}.{gname}", K, sizes, r, alpha)
                G.active = [n in target_modules for n in names]
                L.lora[gname] = G
                groups.append((G, holder, names))
        self.arena.commit()                       # allocates and calls every group's materialise()
        gen = torch.Generator(device="cpu").manual_seed(init_seed)
        for G, holder, names in groups:
            for j, n in enumerate(names):
                if not G.active[j]:
                    am, bm = self.arena.mask_view(G.key + ".A"), self.arena.mask_view(G.key + ".B")
                    am[j * r:(j + 1) * r] = 0
                    off = sum(G.n_sizes[:j])
                    bm[off:off + G.n_sizes[j]] = 0
                    continue
                base = getattr(holder, n)
                wrap = LoraLinear(base, G, j, r, alpha, dropout)
                wrap._bind()
                A, B, _, _ = G.target_views(j)
                A.copy_((torch.randn(A.shape, generator=gen) * (1.0 / r)).to(dev))   # init_lora_weights="gaussian"
                B.zero_()
                setattr(holder, n, wrap)
        for n, p in self.named_parameters():
            if "lora_" not in n:
                p.requires_grad_(False)
        self.arena.pack()
        return self

    @torch.no_grad()
    def merge_and_unload(self, reinit_seed: int = 0):
        """PEFT `merge_and_unload()` (reason.py:441-444): W <- W + (alpha / r) B A for every target, delta rounded to the
        weight dtype and added, as PEFT's `merge` does.  The adapter slots stay allocated (one flat arena) and restart from
        PEFT's initial state — A gaussian, B = 0, a branch that adds nothing — which is exactly what the reference builds
        next (`_prep_for_training` -> `get_peft_model`, reason.py:533-536)."""
        eng = self.ensure_packed()
        if self.arena is None:
            return self
        self.arena.pack()
        gen = torch.Generator(device="cpu").manual_seed(reinit_seed)
        merged = {}
        for L in eng.layers:
            for gname, wname in (("qkv", "Wqkv"), ("o", "Wo"), ("gu", "Wgu"), ("d", "Wd")):
                G = L.lora[gname]
                if G is None:
                    continue
                W = getattr(L, wname)
                W.copy_(ops.gemm_nt(G.B, G.AT, alpha=G.scaling, res=W))           # [N, K] = s * B A + W
                merged[(id(L), wname)] = W
                for j in range(len(G.n_sizes)):
                    A, B, _, _ = G.target_views(j)
                    if getattr(G, "active", None) is None or G.active[j]:
                        A.copy_((torch.randn(A.shape, generator=gen) * (1.0 / G.r)).to(A.device))
                    B.zero_()
        self.arena.pack()
        # the packed images alias the parameters only when those are bf16; otherwise write the merge back into the
        # parameters themselves so the repack below (and any later one) starts from the merged weights
        for l, L in zip(self.model.layers, eng.layers):
            a, m = l.self_attn, l.mlp
            for wname, mods in (("Wqkv", (a.q_proj, a.k_proj, a.v_proj)), ("Wo", (a.o_proj,)), ("Wgu", (m.gate_proj, m.up_proj)),
                                ("Wd", (m.down_proj,))):
                W = merged.get((id(L), wname))
                if W is None:
                    continue
                off = 0
                for mod in mods:
                    p = _unwrap(mod).weight
                    n = p.shape[0]
                    if p.dtype != BF16:
                        p.data.copy_(W[off:off + n].to(p.dtype))
                    off += n
        self._packed_sig = None                                                   # transposed weight images are stale
        self.ensure_packed()
        return self

    @contextlib.contextmanager
    def disable_adapter(self):
        """PEFT's `with model.disable_adapter():` — the reference policy of GRPO (grpo_trainer.py:636-640)."""
        prev = self._lora_enabled
        self._lora_enabled = False
        try:
            yield
        finally:
            self._lora_enabled = prev

    # ---- forward --------------------------------------------------------------------------------------------
    def set_dropout_seed(self, seed: int) -> None:
        """base seed of the LoRA dropout masks (per-pass seeds derive from it and a call counter, which is reset)"""
        self._dropout_seed = int(seed) & 0xFFFFFFFF
        self._dropout_calls = 0

    def hidden_states(self, inputs_embeds: torch.Tensor, attention_mask: Optional[torch.Tensor],
                      position_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[B,S,H] embeddings -> final-normed hidden states [B*S, H] (differentiable)."""
        eng = self.ensure_packed()
        if self.arena is not None:
            self.arena.pack_if_stale()
        B, S, H = inputs_embeds.shape
        dev = inputs_embeds.device
        if position_ids is None:      # Qwen3Model.forward: arange regardless of padding (TF:qwen3:391-394)
            pos = torch.arange(S, dtype=torch.int32, device=dev).repeat(B)
            max_pos = S
        else:
            pos = position_ids.to(torch.int32).reshape(-1).contiguous()
            max_pos = self.config.max_position_embeddings
        kmask = None if attention_mask is None else attention_mask.to(torch.uint8).contiguous()
        meta = SeqMeta(B=B, S=S, pos=pos, kmask=kmask, lora_on=self._lora_enabled, max_pos=max_pos)
        p_drop = getattr(self, "lora_dropout_p", 0.0)
        if p_drop > 0.0 and self.training and self._lora_enabled:
            # a fresh set of masks per forward pass, as nn.Dropout draws; the backward of THIS pass regenerates them from
            # the seed kept in `meta` (counter-based hash, no stored masks)
            self._dropout_calls = getattr(self, "_dropout_calls", 0) + 1
            meta.drop_p = p_drop
            meta.drop_seed = (getattr(self, "_dropout_seed", 0x5EED) * 0x9E3779B1 + self._dropout_calls * 0x85EBCA6B) & 0xFFFFFFFF
        x = inputs_embeds.reshape(B * S, H)
        if x.dtype != BF16:
            x = x.to(BF16)
        anchor = self.arena.anchor if self.arena is not None else x.new_zeros(1, dtype=torch.float32)
        return _StackFn.apply(x.contiguous(), anchor, self, meta, torch.is_grad_enabled())

    def hidden_states_shared(self, prompt_embeds: torch.Tensor, prompt_mask: torch.Tensor, completion_ids: torch.Tensor,
                             completion_mask: torch.Tensor, copies: int, side=None):
        """The hidden states a [B, P + C] pass would produce at the positions GRPO keeps (grpo_trainer.py:510-520 + :779: the last
        prompt position and the completion positions), for B = R * copies rows made of R distinct prompts whose `copies` rollouts
        are consecutive: prompt_embeds [R, P, H] (differentiable), prompt_mask [R, P], completion_ids / completion_mask [B, C].
        -> (hidden of the last prompt row [R, H], hidden of the completion rows [B * C, H]), both final-normed, differentiable.
        LoRA dropout (training mode): one mask stream for the shared prompt rows, one for the completion rows."""
        eng = self.ensure_packed()
        if self.arena is not None:
            self.arena.pack_if_stale()
        R, P, H = prompt_embeds.shape
        B, C = completion_ids.shape
        assert B == R * copies
        dev = prompt_embeds.device
        S = P + C
        kfull = torch.cat([prompt_mask.to(torch.uint8).repeat_interleave(copies, dim=0), completion_mask.to(torch.uint8)], dim=1).contiguous()
        mp = SeqMeta(B=R, S=P, pos=torch.arange(P, dtype=torch.int32, device=dev).repeat(R), kmask=prompt_mask.to(torch.uint8).contiguous(),
                     lora_on=self._lora_enabled, max_pos=S)
        mc = SeqMeta(B=B, S=C, pos=(torch.arange(C, dtype=torch.int32, device=dev) + P).repeat(B), kmask=kfull,
                     lora_on=self._lora_enabled, max_pos=S)
        p_drop = getattr(self, "lora_dropout_p", 0.0)
        if p_drop > 0.0 and self.training and self._lora_enabled:
            self._dropout_calls = getattr(self, "_dropout_calls", 0) + 1
            seed = (getattr(self, "_dropout_seed", 0x5EED) * 0x9E3779B1 + self._dropout_calls * 0x85EBCA6B) & 0xFFFFFFFF
            mp.drop_p = mc.drop_p = p_drop
            mp.drop_seed = seed
            mc.drop_seed = (seed * 0x2C1B3C6D + 0x5BD1E995) & 0xFFFFFFFF      # the two segments must not share mask streams
        xp = prompt_embeds.reshape(R * P, H)
        if xp.dtype != BF16:
            xp = xp.to(BF16)
        xc = torch.empty((B * C, H), dtype=BF16, device=dev)
        ops.embed_scatter_fwd(completion_ids.to(torch.int32).reshape(-1).contiguous(), None, eng.E, None, xc)
        anchor = self.arena.anchor if self.arena is not None else xp.new_zeros(1, dtype=torch.float32)
        return _SharedStackFn.apply(xp.contiguous(), xc, anchor, self, mp, mc, copies, side, torch.is_grad_enabled())

    def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, labels=None, position_ids=None,
                return_logits: bool = True, **unused):
        """CausalLMOutputWithPast(loss, logits): shifted cross-entropy with ignore_index -100
        (TF:qwen3:482-499, TF:loss/loss_utils.py:49-71).  logits are bf16 like the reference's lm_head output."""
        from transformers.modeling_outputs import CausalLMOutputWithPast
        if inputs_embeds is None:
            eng = self.ensure_packed()
            ids32 = input_ids.to(torch.int32).reshape(-1).contiguous()
            x = torch.empty((ids32.numel(), eng.H), dtype=BF16, device=ids32.device)
            ops.embed_scatter_fwd(ids32, None, eng.E, None, x)
            inputs_embeds = x.view(*input_ids.shape, eng.H)
        B, S, H = inputs_embeds.shape
        rows = tgt = None
        if labels is not None:       # row selection first: it only depends on the labels (one host sync, up front)
            shift = labels[:, 1:].reshape(-1)
            valid = torch.nonzero(shift != -100, as_tuple=False).reshape(-1)
            b_idx, s_idx = valid // (S - 1), valid % (S - 1)
            rows = (b_idx * S + s_idx).to(torch.int32)
            tgt = shift[valid].to(torch.int32)
        hid = self.hidden_states(inputs_embeds, attention_mask, position_ids)
        loss = None
        if labels is not None and rows.numel() == 0:
            # every label is -100 (e.g. the assistant span fell to truncation): HF's mean over zero positions is NaN
            # (TF:loss/loss_utils.py:32-46) and so is this loss; its backward sends zeros (an all-ignored batch teaches nothing)
            loss = _NanLossFn.apply(hid)
        elif labels is not None:
            hsel = _GatherRowsFn.apply(hid, rows)
            logp = _LogProbFn.apply(hsel, self, tgt)
            loss = _neg_mean(logp)
        logits = None
        if return_logits:
            logits = _LogitsFn.apply(hid, self).view(B, S, -1)
        return CausalLMOutputWithPast(loss=loss, logits=logits)

    def token_logprobs(self, hid: torch.Tensor, rows: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        """log-probabilities of `targets` at hidden rows `rows` (int32) — the fused form of
        `_get_per_token_logps` (grpo_trainer.py:510-520) restricted to the rows the caller keeps."""
        return _LogProbFn.apply(_GatherRowsFn.apply(hid, rows), self, targets)


class _GatherRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rows):
        ctx.save_for_backward(rows)
        ctx.n = x.shape[0]
        return ops.gather_rows(rows, x)

    @staticmethod
    def backward(ctx, dy):
        (rows,) = ctx.saved_tensors
        return ops.scatter_rows(rows, dy.contiguous(), ctx.n), None


class _NanLossFn(torch.autograd.Function):
    """the loss of a batch without a single supervised position: NaN forward, zero gradient"""

    @staticmethod
    def forward(ctx, hid):
        ctx.meta = (hid.shape, hid.dtype, hid.device)
        return torch.full((), float("nan"), dtype=torch.float32, device=hid.device)

    @staticmethod
    def backward(ctx, g):
        shape, dt, dev = ctx.meta
        return torch.zeros(shape, dtype=dt, device=dev)


class _NegMeanFn(torch.autograd.Function):
    """-mean(x) over a small fp32 vector (the CE reduction); backward is a constant fill."""

    @staticmethod
    def forward(ctx, x):
        ctx.n = x.numel()
        return ops.vec_sum(x.contiguous(), -1.0 / x.numel())[0]

    @staticmethod
    def backward(ctx, g):
        return (-g / ctx.n).expand(ctx.n).contiguous()


def _neg_mean(x):
    return _NegMeanFn.apply(x)


# ---------------------------------------------------------------------------------------------- NT-v2 (ESM)
class _EsmSelfAttention(nn.Module):
    def __init__(self, H, dev, dt):
        super().__init__()
        self.query = HipLinear(H, H, device=dev, dtype=dt)
        self.key = HipLinear(H, H, device=dev, dtype=dt)
        self.value = HipLinear(H, H, device=dev, dtype=dt)


class _EsmSelfOutput(nn.Module):
    def __init__(self, H, dev, dt):
        super().__init__()
        self.dense = HipLinear(H, H, device=dev, dtype=dt)


class _EsmAttention(nn.Module):
    def __init__(self, c, dev, dt):
        super().__init__()
        self.self = _EsmSelfAttention(c.hidden_size, dev, dt)
        self.output = _EsmSelfOutput(c.hidden_size, dev, dt)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps, device=dev, dtype=dt)


class _EsmFF(nn.Module):
    def __init__(self, i, o, dev, dt):
        super().__init__()
        self.dense = HipLinear(i, o, bias=False, device=dev, dtype=dt)


class _EsmLayer(nn.Module):
    def __init__(self, c, dev, dt):
        super().__init__()
        self.attention = _EsmAttention(c, dev, dt)
        self.intermediate = _EsmFF(c.hidden_size, 2 * c.intermediate_size, dev, dt)    # NT-v2 GLU: [2F, H], no bias
        self.output = _EsmFF(c.intermediate_size, c.hidden_size, dev, dt)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps, device=dev, dtype=dt)


class _EsmEncoder(nn.Module):
    def __init__(self, c, dev, dt):
        super().__init__()
        self.layer = nn.ModuleList([_EsmLayer(c, dev, dt) for _ in range(c.num_hidden_layers)])
        self.emb_layer_norm_after = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps, device=dev, dtype=dt)


class _EsmEmbeddings(nn.Module):
    def __init__(self, c, dev, dt):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=c.pad_token_id, device=dev, dtype=dt)


class _EsmBody(nn.Module):
    def __init__(self, c, dev, dt):
        super().__init__()
        self.embeddings = _EsmEmbeddings(c, dev, dt)
        self.encoder = _EsmEncoder(c, dev, dt)


class NTEncoderForMaskedLM(nn.Module):
    """HIP-backed stand-in for the NT-v2 `EsmForMaskedLM` the reference loads with trust_remote_code
    (dna_llm.py:79-81).  Only what the reference uses is computed: `outputs.hidden_states[-1]` (dna_llm.py:150-156);
    the MLM head the reference evaluates and discards is not run."""

    def __init__(self, config, device=None, dtype=BF16):
        super().__init__()
        self.config = config
        self.esm = _EsmBody(config, device, dtype)
        self.engine: Optional[EsmEngine] = None
        self._packed_sig = None

    @property
    def device(self):
        return self.esm.embeddings.word_embeddings.weight.device

    @torch.no_grad()
    def init_weights(self, std: float = 0.02, seed: int = 0):
        g = torch.Generator(device=self.device).manual_seed(seed)
        for n, p in self.named_parameters():
            if p.dim() >= 2:
                p.normal_(0.0, std, generator=g)
            elif n.endswith("weight"):
                p.fill_(1.0)
            else:
                p.zero_()
        self._packed_sig = None

    def ensure_packed(self):
        params = list(self.parameters())
        sig = tuple((p.data_ptr(), p._version, str(p.device), p.dtype) for p in params)
        if sig == self._packed_sig:
            return self.engine
        dev = self.device
        eng = EsmEngine(self.config, dev)

        def bf(p):
            return (p.data if p.dtype == BF16 else p.data.to(BF16)).contiguous()

        eng.E = bf(self.esm.embeddings.word_embeddings.weight)
        for l, L in zip(self.esm.encoder.layer, eng.layers):
            s = l.attention.self
            L.Wqkv = torch.cat([bf(s.query.weight), bf(s.key.weight), bf(s.value.weight)], 0).contiguous()
            L.bqkv = torch.cat([bf(s.query.bias), bf(s.key.bias), bf(s.value.bias)], 0).contiguous()
            L.Wo, L.bo = bf(l.attention.output.dense.weight), bf(l.attention.output.dense.bias)
            L.ln1_w, L.ln1_b = bf(l.attention.LayerNorm.weight), bf(l.attention.LayerNorm.bias)
            L.ln2_w, L.ln2_b = bf(l.LayerNorm.weight), bf(l.LayerNorm.bias)
            L.Wup, L.Wdown = bf(l.intermediate.dense.weight), bf(l.output.dense.weight)
        eng.lnf_w, eng.lnf_b = bf(self.esm.encoder.emb_layer_norm_after.weight), bf(self.esm.encoder.emb_layer_norm_after.bias)
        self.engine = eng
        self._packed_sig = sig
        return eng

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, output_hidden_states: bool = True, **unused):
        eng = self.ensure_packed()
        n, S = input_ids.shape
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        hid = eng.forward(input_ids.to(torch.int32).contiguous(), attention_mask.to(torch.uint8).contiguous())
        hid = hid.view(n, S, -1)
        return SimpleNamespace(hidden_states=(hid,), last_hidden_state=hid, logits=None)
