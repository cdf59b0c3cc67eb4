"""Kernel-level runtimes of the two networks on the hot path.

`QwenEngine`   — Qwen3 decoder stack (TF:models/qwen3/modeling_qwen3.py:294-427): forward, hand-written backward
                 (input gradient + LoRA weight gradients written straight into the TrainableArena), KV-cache
                 prefill and single-token decode.
`EsmEngine`    — NT-v2 / ESM encoder forward (TF:models/esm/modeling_esm.py:466-555 + NT-v2 SwiGLU FFN), no
                 backward: the reference always runs the DNA encoder under no_grad (dna_llm.py:121).

Both consume "packed" bf16 weights: q/k/v fused into one [Nq+2Nkv, H] matrix, gate/up fused into [2F, H], plus
pre-transposed copies of the frozen weights so that every dgrad is the same K-contiguous NT GEMM.  The packed
buffers are the storage of the nn.Parameters the callers see (they are re-aliased after .to()/.load_state_dict()).
All arithmetic is libbioreason_hip.so; torch provides memory, streams and autograd bookkeeping only.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .arena import TrainableArena

BF16 = torch.bfloat16


def rope_tables(npos: int, hd: int, theta: float, device, round_bf16: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin [npos, hd/2] exactly as HF builds them (fp32 inv_freq x position, TF:qwen3:126-138, TF:esm:144-159);
    HF then casts them to the activation dtype (bf16) — `round_bf16` reproduces that."""
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.arange(npos, dtype=torch.float32)[:, None] * inv[None, :]
    c, s = fr.cos(), fr.sin()
    if round_bf16:
        c, s = c.to(BF16).float(), s.to(BF16).float()
    return c.contiguous().to(device), s.contiguous().to(device)


# =============================================================================================== LoRA groups
class LoraGroup:
    """Adapters of one fused projection: targets j = 0..n-1 with output row blocks `n_sizes`.
    Master layout (fp32, in the arena): Acat [r_pad, K] (row block j = A_j), Bcat [N, r_pad] (block (rows_j, cols_j) = B_j).
    bf16 images: A, AT [K, r_pad], B, BT [r_pad, N]."""

    def __init__(self, arena: TrainableArena, key: str, K: int, n_sizes: List[int], r: int, alpha: float):
        self.key, self.K, self.n_sizes, self.r = key, K, n_sizes, r
        self.N = sum(n_sizes)
        self.r_pad = (len(n_sizes) * r + 63) // 64 * 64
        self.scaling = alpha / r
        self.arena = arena
        arena.add(key + ".A", self.r_pad, K)
        arena.add(key + ".B", self.N, self.r_pad)
        self.on_views: List = []          # callbacks of the module shells that alias the per-target views
        arena.on_rebind(self.materialise)

    def materialise(self):
        a, dev = self.arena, self.arena.device
        self.A_master, self.B_master = a.param(self.key + ".A"), a.param(self.key + ".B")
        self.A_grad, self.B_grad = a.grad(self.key + ".A"), a.grad(self.key + ".B")
        am, bm = a.mask_view(self.key + ".A"), a.mask_view(self.key + ".B")
        off = 0
        active = getattr(self, "active", None)
        for j, n in enumerate(self.n_sizes):
            if active is None or active[j]:
                am[j * self.r:(j + 1) * self.r, :] = 1
                bm[off:off + n, j * self.r:(j + 1) * self.r] = 1
            off += n
        self.A = torch.zeros(self.r_pad, self.K, dtype=BF16, device=dev)
        self.AT = torch.zeros(self.K, self.r_pad, dtype=BF16, device=dev)
        self.B = torch.zeros(self.N, self.r_pad, dtype=BF16, device=dev)
        self.BT = torch.zeros(self.r_pad, self.N, dtype=BF16, device=dev)
        a.register_pack(self.A_master, self.A, False)
        a.register_pack(self.A_master, self.AT, True)
        a.register_pack(self.B_master, self.B, False)
        a.register_pack(self.B_master, self.BT, True)
        for fn in self.on_views:
            fn()

    def target_views(self, j: int):
        """(A_j param, B_j param, A_j grad, B_j grad) — what lora_A.default.weight / lora_B.default.weight alias"""
        off = sum(self.n_sizes[:j])
        n = self.n_sizes[j]
        rs = slice(j * self.r, (j + 1) * self.r)
        return (self.A_master[rs, :], self.B_master[off:off + n, rs], self.A_grad[rs, :], self.B_grad[off:off + n, rs])


# =============================================================================================== Qwen3
@dataclass
class QwenLayerW:
    ln1: torch.Tensor = None
    ln2: torch.Tensor = None
    qn: torch.Tensor = None
    kn: torch.Tensor = None
    Wqkv: torch.Tensor = None
    Wo: torch.Tensor = None
    Wgu: torch.Tensor = None
    Wd: torch.Tensor = None
    WqkvT: torch.Tensor = None
    WoT: torch.Tensor = None
    WguT: torch.Tensor = None
    WdT: torch.Tensor = None
    lora: Dict[str, Optional[LoraGroup]] = field(default_factory=lambda: {"qkv": None, "o": None, "gu": None, "d": None})


@dataclass
class SeqMeta:
    B: int
    S: int
    pos: torch.Tensor                 # int32 [B*S]
    kmask: Optional[torch.Tensor]     # uint8 [B, S] or None
    lora_on: bool = True
    max_pos: int = 0                  # largest rotary position + 1 (0 -> S)
    drop_p: float = 0.0               # LoRA dropout probability of this pass (training mode only) ...
    drop_seed: int = 0                # ... and the 32-bit seed its mask streams derive from


def _mix32(a: int, b: int) -> int:
    h = (a * 0x9E3779B1 + b * 0x85EBCA6B + 0x165667B1) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0x2C1B3C6D) & 0xFFFFFFFF
    h ^= h >> 12
    h = (h * 0x297A2D39) & 0xFFFFFFFF
    h ^= h >> 15
    return h


import contextlib as _contextlib
import os as _os
# no-grad passes: SwiGLU in the epilogue of the gate/up projection (ops.gemm_swiglu); BRA_FUSE_SWIGLU=0 keeps the two launches (A/B runs)
FUSE_SWIGLU = _os.environ.get("BRA_FUSE_SWIGLU", "1") == "1"

_nullctx = _contextlib.nullcontext

LORA_GROUP_INDEX = {"qkv": 0, "o": 1, "gu": 2, "d": 3}


def lora_drop_seeds(pass_seed: int, layer: int, group: str, n_targets: int):
    """mask-stream seeds of the targets of one fused projection (PEFT: one nn.Dropout per target module)"""
    base = (layer * 4 + LORA_GROUP_INDEX[group]) * 4
    return [_mix32(pass_seed, base + j) for j in range(n_targets)]


class QwenEngine:
    def __init__(self, cfg, device):
        self.cfg = cfg
        self.device = device
        self.H = cfg.hidden_size
        self.Hq, self.Hkv, self.hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        self.F = cfg.intermediate_size
        self.L = cfg.num_hidden_layers
        self.V = cfg.vocab_size
        self.eps = cfg.rms_norm_eps
        self.Nq, self.Nkv = self.Hq * self.hd, self.Hkv * self.hd
        self.scale = self.hd ** -0.5
        self.layers: List[QwenLayerW] = [QwenLayerW() for _ in range(self.L)]
        self.norm_w: torch.Tensor = None
        self.E: torch.Tensor = None          # [V, H] tied embedding / lm_head
        self.ET: Optional[torch.Tensor] = None
        self._rope = None
        self._rope_len = 0
        self.layer_done_hook = None          # called with the layer index when a layer's backward (its LoRA grads) is complete

    # ------------------------------------------------------------------ helpers
    def rope(self, npos: int):
        if self._rope is None or self._rope_len < npos:
            n = max(npos, 512)
            theta = self.cfg.rope_parameters["rope_theta"] if hasattr(self.cfg, "rope_parameters") else self.cfg.rope_theta
            self._rope = rope_tables(n, self.hd, theta, self.device)
            self._rope_len = n
        return self._rope

    def ensure_transposed(self):
        for L in self.layers:
            if L.WqkvT is None:
                L.WqkvT = ops.transpose2d(L.Wqkv)
                L.WoT = ops.transpose2d(L.Wo)
                L.WguT = ops.transpose2d(L.Wgu)
                L.WdT = ops.transpose2d(L.Wd)
        if self.ET is None:
            self.ET = ops.transpose2d(self.E)

    # below this many rows a [rows, K] x [r_pad, K]^T product goes to the row-block kernel of k_lora.hip (32 rows per workgroup, p = 0)
    # instead of the tiled GEMM: a 128-row tile grid over N = 64 / 128 columns is 16 workgroups at 2048 rows (measured: 182 us vs ~28 us
    # for dy [2048, 12288] x B^T); at 19 488 rows the tiled GEMM fills the chip
    SMALL_M = 8192

    @staticmethod
    def _down(x2d, A, scaling, live_rows):
        """s x A^T -> [rows, r_pad] (A = the adapter image [r_pad, K]; its first `live_rows` = targets x rank rows are adapters, the
        rest padding)"""
        if x2d.shape[0] < QwenEngine.SMALL_M and A.shape[0] in (32, 64, 128):
            # p = 0: every 32-row block that holds adapter rows is live, whatever the rank (r = 64: one target spans two blocks;
            # r = 16 x 3 targets: 48 rows in two blocks); the kernel zeroes the blocks past the seeds it is given — the all-padding
            # blocks (r = 32 x 3 targets in a 128-row image: the fourth) are not multiplied
            return ops.lora_down_drop(x2d, A, scaling, 0.0, [0] * min(A.shape[0] // 32, (live_rows + 31) // 32))
        return ops.gemm_nt(x2d, A, alpha=scaling)

    @staticmethod
    def _lora_fwd(x2d, W, G: Optional[LoraGroup], on: bool, res=None, out=None, drop=None):
        """y = x W^T (+ (s dropout(x) A^T) B^T) (+res); returns (y, t).  drop = (p, seeds per target) in training mode"""
        if G is not None and on:
            if drop is not None:
                t = ops.lora_down_drop(x2d, G.A, G.scaling, drop[0], drop[1])
            else:
                t = QwenEngine._down(x2d, G.A, G.scaling, len(G.n_sizes) * G.r)
            return ops.gemm_nt(x2d, W, a2=t, b2=G.B, res=res, out=out), t
        return ops.gemm_nt(x2d, W, res=res, out=out), None

    @staticmethod
    def _lora_bwd(dy, WT, G: Optional[LoraGroup], on: bool, x2d, t, drop=None):
        """dx = dy W (+ dropout'(s (dy B) A)); accumulates dA, dB into the arena."""
        if G is not None and on:
            dts = QwenEngine._down(dy, G.BT, G.scaling, len(G.n_sizes) * G.r)   # [T, r_pad] = s * dy B
            if drop is not None:
                dxl = ops.lora_up_drop(dts, G.AT, drop[0], drop[1])      # the branch's input gradient, masked per target
                dx = ops.gemm_nt(dy, WT, res=dxl)
            else:
                dx = ops.gemm_nt(dy, WT, a2=dts, b2=G.AT)
            # weight gradients straight from the row-major activations (k_wgrad.hip): no transposed copies in HBM
            ops.wgrad_tn(dy, t, G.B_grad)                                # dB [N, r] += dy^T t      (t already holds s)
            ops.wgrad_tn(x2d, dts, G.A_grad, transposed_out=True, drop=drop)   # dA [r, K] += (s dy B)^T dropout(x)
            return dx
        return ops.gemm_nt(dy, WT)

    def _drop(self, m: "SeqMeta", li: int, group: str):
        G = self.layers[li].lora[group]
        if G is None or not m.lora_on or m.drop_p <= 0.0:
            return None
        if G.r != 32:
            raise NotImplementedError("LoRA dropout needs r = 32 (one mask stream per 32-column rank block)")
        return (m.drop_p, lora_drop_seeds(m.drop_seed, li, group, len(G.n_sizes)))

    # ------------------------------------------------------------------ fp8 x fp8 projections (opt-in; BASELINE config 5)
    def use_fp8(self, W8):
        """context: every no-grad layer forward issued inside runs its four projections on the fp8 MFMA path (ops.gemm_fp8_nt) over
        `W8` = one record per layer {"Wqkv": (q, s), "Wo": .., "Wgu": .., "Wd": ..} of row-major e4m3 images + row scales
        (fp8_weight_images) — the prompt pass of an fp8 rollout (merged policy weights) and, with GRPOConfig.ref_fp8, the reference
        pass (base weights).  None = the bf16 path."""
        eng = self

        class _Ctx:
            def __enter__(self_):
                self_.prev = getattr(eng, "_fp8", None)
                eng._fp8 = W8

            def __exit__(self_, *a):
                eng._fp8 = self_.prev
        return _Ctx()

    def fp8_supported(self) -> bool:
        """the fp8 GEMM takes K % 128 == 0 (one MFMA per 128-byte K-tile) and row strides of whole 16-byte chunks"""
        return self.H % 128 == 0 and self.F % 128 == 0 and self.Nq % 128 == 0

    def fp8_weight_images(self, merged=None):
        """row-major e4m3 images of the layers' projections: q[n, k] = e4m3(W[n, k] w[k] / s[n]) with the input's RMSNorm weight w
        folded into Wqkv / Wgu (bra_dec_pack_weights_fp8's rule: the token loop's fragment-ordered image holds the same bytes).
        merged: per-layer dicts {"Wqkv", "Wo", "Wgu", "Wd"} of LoRA-merged weights ([gate; up] rows NOT interleaved); None = base."""
        out = []
        for li, L in enumerate(self.layers):
            W = merged[li] if merged is not None else {"Wqkv": L.Wqkv, "Wo": L.Wo, "Wgu": L.Wgu, "Wd": L.Wd}
            out.append({"Wqkv": ops.quant_rows_fp8(W["Wqkv"], colw=L.ln1), "Wo": ops.quant_rows_fp8(W["Wo"]),
                        "Wgu": ops.quant_rows_fp8(W["Wgu"], colw=L.ln2), "Wd": ops.quant_rows_fp8(W["Wd"])})
        return out

    def _layer_fwd_fp8(self, li: int, x: torch.Tensor, m: SeqMeta, kv_out, R):
        """layer_fwd(save=False) with W8A8 projections: activations quantised per token (e4m3, absmax / 448), the RMSNorm row factor on
        the activation scale (y = rstd (x (W w)^T), as the token loop computes it), fp32 accumulation, bf16 outputs; attention,
        rotary, q / k norms and the residual stream stay bf16.  No LoRA branch: `R` holds merged (or base) weights."""
        L = self.layers[li]
        B, S, T = m.B, m.S, m.B * m.S
        cosT, sinT = self.rope(m.max_pos or m.S)
        xq, xs = ops.quant_rows_fp8(x, rms_eps=self.eps)
        qkv = ops.gemm_fp8_nt(xq, xs, R["Wqkv"][0], R["Wqkv"][1])
        q = torch.empty((B, S, self.Hq, self.hd), dtype=BF16, device=x.device)
        if kv_out is None:
            k = torch.empty((B, S, self.Hkv, self.hd), dtype=BF16, device=x.device)
            v = torch.empty((B, S, self.Hkv, self.hd), dtype=BF16, device=x.device)
            s_off = 0
        else:
            kc, vc, s_off = kv_out[:3]
            k, v = kc.permute(0, 2, 1, 3), vc.permute(0, 2, 1, 3)
        ops.qk_norm_rope_fwd(qkv, L.qn, L.kn, cosT, sinT, m.pos, S, self.Hq, self.Hkv, self.hd, self.eps, 1.0, q, k, v, s_off)
        if kv_out is not None:
            k, v = k[:, :s_off + S], v[:, :s_off + S]
        vt = ops.head_transpose(v)
        if kv_out is not None and len(kv_out) > 3 and kv_out[3] is not None:
            kv_out[3].append(vt)
        o, _ = ops.attn_fwd(q, k, vt, m.kmask, True, self.scale, need_lse=False)
        oq, os_ = ops.quant_rows_fp8(o.view(T, self.Nq))
        h = ops.gemm_fp8_nt(oq, os_, R["Wo"][0], R["Wo"][1], res=x)
        hq, hs = ops.quant_rows_fp8(h, rms_eps=self.eps)
        gu = ops.gemm_fp8_nt(hq, hs, R["Wgu"][0], R["Wgu"][1])
        aq, as_ = ops.swiglu_quant_fp8(gu)
        return ops.gemm_fp8_nt(aq, as_, R["Wd"][0], R["Wd"][1], res=h)

    # ------------------------------------------------------------------ one decoder layer
    def layer_fwd(self, li: int, x: torch.Tensor, m: SeqMeta, save: bool, kv_out=None):
        """x [T, H] bf16 -> y [T, H].  kv_out = (kcache, vcache, s_off): also write K/V rows into a cache
        [B, Hkv, Smax, hd] (prefill) and attend over the cache view."""
        w8 = getattr(self, "_fp8", None)
        if w8 is not None and not save:
            return self._layer_fwd_fp8(li, x, m, kv_out, w8[li]), None
        L = self.layers[li]
        B, S, T = m.B, m.S, m.B * m.S
        cosT, sinT = self.rope(m.max_pos or m.S)
        xn = ops.rmsnorm_fwd(x, L.ln1, self.eps)
        qkv, t1 = self._lora_fwd(xn, L.Wqkv, L.lora["qkv"], m.lora_on, drop=self._drop(m, li, "qkv"))
        q = torch.empty((B, S, self.Hq, self.hd), dtype=BF16, device=x.device)
        if kv_out is None:
            k = torch.empty((B, S, self.Hkv, self.hd), dtype=BF16, device=x.device)
            v = torch.empty((B, S, self.Hkv, self.hd), dtype=BF16, device=x.device)
            s_off = 0
        else:
            kc, vc, s_off = kv_out[:3]
            k, v = kc.permute(0, 2, 1, 3), vc.permute(0, 2, 1, 3)        # [B, Smax, Hkv, hd] views
        ops.qk_norm_rope_fwd(qkv, L.qn, L.kn, cosT, sinT, m.pos, S, self.Hq, self.Hkv, self.hd, self.eps, 1.0, q, k, v, s_off)
        if kv_out is not None:
            k, v = k[:, :s_off + S], v[:, :s_off + S]
        vt = ops.head_transpose(v)
        if kv_out is not None and len(kv_out) > 3 and kv_out[3] is not None:
            kv_out[3].append(vt)                                          # the prompt's V^T image, kept for shared-prefix decode
        o, lse = ops.attn_fwd(q, k, vt, m.kmask, True, self.scale, need_lse=save)
        o2 = o.view(T, self.Nq)
        h, t2 = self._lora_fwd(o2, L.Wo, L.lora["o"], m.lora_on, res=x, drop=self._drop(m, li, "o"))
        hn = ops.rmsnorm_fwd(h, L.ln2, self.eps)
        gu = t3 = act = None
        if not save and FUSE_SWIGLU:
            # nothing is kept for a backward: the SwiGLU rides in the epilogue of the gate/up projection (ops.gemm_swiglu: the same
            # values in one launch, no [T, 2 F] intermediate); None where the fused kernel does not apply
            G3, d3 = L.lora["gu"], self._drop(m, li, "gu")
            if G3 is not None and m.lora_on:
                t3 = ops.lora_down_drop(hn, G3.A, G3.scaling, d3[0], d3[1]) if d3 is not None else self._down(hn, G3.A, G3.scaling, len(G3.n_sizes) * G3.r)
                act = ops.gemm_swiglu(hn, L.Wgu, a2=t3, b2=G3.B)
            else:
                act = ops.gemm_swiglu(hn, L.Wgu)
        if act is None:
            if t3 is not None:
                gu = ops.gemm_nt(hn, L.Wgu, a2=t3, b2=L.lora["gu"].B)
            else:
                gu, t3 = self._lora_fwd(hn, L.Wgu, L.lora["gu"], m.lora_on, drop=self._drop(m, li, "gu"))
            act = ops.swiglu_fwd(gu)
        y, t4 = self._lora_fwd(act, L.Wd, L.lora["d"], m.lora_on, res=h, drop=self._drop(m, li, "d"))
        saved = (x, xn, t1, qkv, q, k, v, o, lse, t2, h, hn, t3, gu, act, t4) if save else None
        return y, saved

    def layer_bwd(self, li: int, dy: torch.Tensor, saved, m: SeqMeta, prefix: int = 0, extra_dkv=None):
        """prefix > 0: the layer's queries attended to `prefix` key rows that belong to ANOTHER segment (a shared prompt) in front
        of their own; their dK / dV rows are returned as the second value instead of entering this segment's projection backward.
        extra_dkv = (dk_views, dv_views, copies): gradients other segments sent to THIS segment's K / V rows ([R * copies, S, Hkv, hd]
        views), summed over the copies of each row group and added to its own."""
        L = self.layers[li]
        (x, xn, t1, qkv, q, k, v, o, lse, t2, h, hn, t3, gu, act, t4) = saved
        B, S, T = m.B, m.S, m.B * m.S
        cosT, sinT = self.rope(m.max_pos or m.S)
        on = m.lora_on
        dact = self._lora_bwd(dy, L.WdT, L.lora["d"], on, act, t4, drop=self._drop(m, li, "d"))
        dgu = ops.swiglu_bwd(gu, dact)
        dhn = self._lora_bwd(dgu, L.WguT, L.lora["gu"], on, hn, t3, drop=self._drop(m, li, "gu"))
        dh = ops.rmsnorm_bwd(dhn, h, L.ln2, self.eps, dres=dy)             # + residual branch
        do = self._lora_bwd(dh, L.WoT, L.lora["o"], on, o.view(T, self.Nq), t2, drop=self._drop(m, li, "o"))
        dq, dk, dv = ops.attn_bwd(q, k, v, o, do.view(B, S, self.Hq, self.hd), lse, m.kmask, True, self.scale)
        pre = None
        if prefix > 0:
            pre = (dk[:, :prefix], dv[:, :prefix])
            dk, dv = dk[:, prefix:], dv[:, prefix:]
        if extra_dkv is not None:
            dk = ops.group_sum(extra_dkv[0], extra_dkv[2], add=dk)
            dv = ops.group_sum(extra_dkv[1], extra_dkv[2], add=dv)
        dqkv = ops.qk_norm_rope_bwd(qkv, L.qn, L.kn, cosT, sinT, m.pos, S, self.Hq, self.Hkv, self.hd, self.eps, 1.0, dq, dk, dv)
        dxn = self._lora_bwd(dqkv, L.WqkvT, L.lora["qkv"], on, xn, t1, drop=self._drop(m, li, "qkv"))
        dx = ops.rmsnorm_bwd(dxn, x, L.ln1, self.eps, dres=dh)
        return (dx, pre) if prefix > 0 else dx

    # ------------------------------------------------------------------ stack
    def forward_hidden(self, x: torch.Tensor, m: SeqMeta, save: bool):
        """x [T,H] -> final-normed hidden [T,H]; returns (hidden, tape)"""
        tape = []
        for li in range(self.L):
            x, saved = self.layer_fwd(li, x, m, save)
            tape.append(saved)
        hid = ops.rmsnorm_fwd(x, self.norm_w, self.eps)
        return hid, (tape, x) if save else None

    def backward_hidden(self, dhid: torch.Tensor, tape_x, m: SeqMeta):
        tape, xlast = tape_x
        dx = ops.rmsnorm_bwd(dhid, xlast, self.norm_w, self.eps)
        for li in reversed(range(self.L)):
            dx = self.layer_bwd(li, dx, tape[li], m)
            tape[li] = None
            if self.layer_done_hook is not None:
                self.layer_done_hook(li)
        return dx


    # ------------------------------------------------------------------ stack over a shared prompt + per-copy completions
    def forward_hidden_shared(self, xp: torch.Tensor, mp: SeqMeta, xc: torch.Tensor, mc: SeqMeta, copies: int, save: bool, side=None):
        """GRPO's G rollouts of a prompt (grpo_trainer.py:107-116) as TWO row segments: the R distinct prompts [R * P, H] run once,
        the B = R * copies completions [B * C, H] attend to [their prompt's K / V | their own K / V].  Rows of a batched forward are
        independent and causal attention never lets a prompt position see a completion, so every hidden state equals the one of
        the full [B, P + C] pass.  Returns (final-normed hidden of the LAST prompt row of every prompt [R, H], of the completion
        rows [B * C, H], tape).
        `side` (a second HIP stream): the completion chain is issued there, one event per layer behind the prompt chain's K / V —
        at one prompt x 8 rollouts each chain alone fills about half of the chip (grids of 128 - 144 workgroups)."""
        R, P, B, C = mp.B, mp.S, mc.B, mc.S
        assert B == R * copies
        dev = xp.device
        tape_p, tape_c = [], []
        main = torch.cuda.current_stream(dev) if side is not None else None
        if side is not None:
            side.wait_stream(main)
            xc.record_stream(side)
        for li in range(self.L):
            kc_r = torch.empty((R, self.Hkv, P, self.hd), dtype=BF16, device=dev)
            vc_r = torch.empty((R, self.Hkv, P, self.hd), dtype=BF16, device=dev)
            xp, sp = self.layer_fwd(li, xp, mp, save, kv_out=(kc_r, vc_r, 0))
            tape_p.append(sp)
            if side is not None:
                ev = torch.cuda.Event()
                ev.record(main)                              # (the whole prompt layer; its K / V rows are what the other chain needs)
                kc_r.record_stream(side)
                vc_r.record_stream(side)
            with (torch.cuda.stream(side) if side is not None else _nullctx()):
                if side is not None:
                    side.wait_event(ev)
                # every copy gets its prompt's K / V rows in front of its own (one broadcast copy per tensor)
                kc = torch.empty((B, self.Hkv, P + C, self.hd), dtype=BF16, device=dev)
                vc = torch.empty((B, self.Hkv, P + C, self.hd), dtype=BF16, device=dev)
                ops.group_broadcast(kc_r, kc, copies)
                ops.group_broadcast(vc_r, vc, copies)
                xc, sc = self.layer_fwd(li, xc, mc, save, kv_out=(kc, vc, P))
                tape_c.append(sc)
        last = torch.arange(R, device=dev, dtype=torch.int32) * P + (P - 1)
        xp_last = ops.gather_rows(last, xp)
        hid_last = ops.rmsnorm_fwd(xp_last, self.norm_w, self.eps)
        with (torch.cuda.stream(side) if side is not None else _nullctx()):
            hid_c = ops.rmsnorm_fwd(xc, self.norm_w, self.eps)
        if side is not None:
            main.wait_stream(side)
            for t_ in (hid_c, xc):
                t_.record_stream(main)
        return hid_last, hid_c, ((tape_p, tape_c, xp_last, xc, last) if save else None)

    def backward_hidden_shared(self, dhid_last: torch.Tensor, dhid_c: torch.Tensor, tape, mp: SeqMeta, mc: SeqMeta, copies: int, side=None):
        tape_p, tape_c, xp_last, xc, last = tape
        dev = dhid_c.device
        main = torch.cuda.current_stream(dev) if side is not None else None
        if side is not None:
            side.wait_stream(main)
            dhid_c.record_stream(side)
        with (torch.cuda.stream(side) if side is not None else _nullctx()):
            dxc = ops.rmsnorm_bwd(dhid_c, xc, self.norm_w, self.eps)
        dxp = ops.scatter_rows(last, ops.rmsnorm_bwd(dhid_last, xp_last, self.norm_w, self.eps), mp.B * mp.S)
        for li in reversed(range(self.L)):
            with (torch.cuda.stream(side) if side is not None else _nullctx()):
                dxc, pre = self.layer_bwd(li, dxc, tape_c[li], mc, prefix=mp.S)
                tape_c[li] = None
                if side is not None:
                    ev = torch.cuda.Event()
                    ev.record(side)
            if side is not None:
                main.wait_event(ev)                          # the prompt rows' K / V gradients from the copies
                pre[0].record_stream(main)
                pre[1].record_stream(main)
            dxp = self.layer_bwd(li, dxp, tape_p[li], mp, extra_dkv=(pre[0], pre[1], copies))
            tape_p[li] = None
            if self.layer_done_hook is not None:
                self.layer_done_hook(li)                     # (main has waited for the completion chain's layer: its LoRA gradients are in)
        if side is not None:
            main.wait_stream(side)
            dxc.record_stream(main)
        return dxp, dxc


# =============================================================================================== NT-v2 / ESM encoder
@dataclass
class EsmLayerW:
    ln1_w: torch.Tensor = None
    ln1_b: torch.Tensor = None
    Wqkv: torch.Tensor = None
    bqkv: torch.Tensor = None
    Wo: torch.Tensor = None
    bo: torch.Tensor = None
    ln2_w: torch.Tensor = None
    ln2_b: torch.Tensor = None
    Wup: torch.Tensor = None      # [2F, H]
    Wdown: torch.Tensor = None    # [H, F]


class EsmEngine:
    def __init__(self, cfg, device):
        self.cfg, self.device = cfg, device
        self.H = cfg.hidden_size
        self.nh = cfg.num_attention_heads
        self.hd = self.H // self.nh
        self.F = cfg.intermediate_size
        self.L = cfg.num_hidden_layers
        self.eps = cfg.layer_norm_eps
        self.layers: List[EsmLayerW] = [EsmLayerW() for _ in range(self.L)]
        self.E: torch.Tensor = None
        self.lnf_w = self.lnf_b = None
        self._rope = None
        self._rope_len = 0

    def rope(self, npos):
        if self._rope is None or self._rope_len < npos:
            theta = getattr(self.cfg, "rope_theta", 10000.0) or 10000.0
            self._rope = rope_tables(max(npos, 64), self.hd, theta, self.device)
            self._rope_len = max(npos, 64)
        return self._rope

    @torch.no_grad()
    def forward(self, ids32: torch.Tensor, mask_u8: torch.Tensor) -> torch.Tensor:
        """ids32 [n, S] int32, mask_u8 [n, S] -> last hidden state [n*S, H] (after emb_layer_norm_after,
        = outputs.hidden_states[-1] of EsmForMaskedLM)."""
        n, S = ids32.shape
        T = n * S
        dev = ids32.device
        cosT, sinT = self.rope(S)
        pos = torch.arange(S, dtype=torch.int32, device=dev).repeat(n)      # positions ignore padding (TF:esm:733-737)
        # word embeddings * attention_mask (TF:esm:239,267-268): padded rows are zero vectors
        # (a token with mask 0 reads row 0 of a one-row zero table instead of its embedding; no host sync)
        x = torch.empty((T, self.H), dtype=BF16, device=dev)
        zero_row = torch.zeros((1, self.H), dtype=BF16, device=dev)
        tok_src = (mask_u8.reshape(-1).to(torch.int32) - 1).clamp_(max=0).neg_().sub_(1)   # mask 1 -> -1, mask 0 -> 0
        ops.embed_scatter_fwd(ids32.reshape(-1), tok_src, self.E, zero_row, x)
        for L in self.layers:
            xn = ops.layernorm_fwd(x, L.ln1_w, L.ln1_b, self.eps)
            qkv = ops.gemm_nt(xn, L.Wqkv, bias=L.bqkv)
            q = torch.empty((n, S, self.nh, self.hd), dtype=BF16, device=dev)
            k = torch.empty_like(q)
            v = torch.empty_like(q)
            ops.qk_norm_rope_fwd(qkv, None, None, cosT, sinT, pos, S, self.nh, self.nh, self.hd, 0.0, self.hd ** -0.5, q, k, v, 0)
            vt = ops.head_transpose(v)
            o, _ = ops.attn_fwd(q, k, vt, mask_u8, False, 1.0, need_lse=False)     # scaling 1.0: q was pre-scaled (TF:esm:345,374)
            x = ops.gemm_nt(o.view(T, self.H), L.Wo, bias=L.bo, res=x)
            xn = ops.layernorm_fwd(x, L.ln2_w, L.ln2_b, self.eps)
            act = ops.gemm_swiglu(xn, L.Wup) if FUSE_SWIGLU else None
            if act is None:
                act = ops.swiglu_fwd(ops.gemm_nt(xn, L.Wup))
            x = ops.gemm_nt(act, L.Wdown, res=x)
        return ops.layernorm_fwd(x, self.lnf_w, self.lnf_b, self.eps)
