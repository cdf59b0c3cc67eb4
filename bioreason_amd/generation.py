"""KV-cache generation for the HIP Qwen3: prefill + single-token decode + device-side sampling.

Replaces `text_model.generate(inputs_embeds=..., attention_mask=..., use_cache=True, **kw)` as called by
DNALLMModel.generate (bioreason/models/dna_llm.py:297-304), i.e. HF GenerationMixin.generate/_sample
(TF:generation/utils.py:2261, 2783-2925) with a DynamicCache:
  * rotary positions = cumsum(attention_mask) - 1 (TF:generation/utils.py:751-773), continuing +1 per new token;
  * left-padded prompts are handled by the key-validity mask; new positions are always attendable;
  * finished rows emit pad_token_id; a row finishes when it emits eos_token_id; generation stops when every row
    has finished (checked every few steps so the host never blocks the launch queue per token) or at
    max_new_tokens; with `inputs_embeds` only the NEW tokens are returned (grpo_trainer.py:588-596).
The whole step is kernel launches on the current stream; the sampler keeps tokens, finished flags and the RNG
counter on the device.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import ops
from ._lib import current_stream, get_lib
from .engine import BF16, QwenEngine, SeqMeta, _nullctx


class KVCache:
    def __init__(self, eng: QwenEngine, B: int, Smax: int, device):
        self.k = [torch.zeros((B, eng.Hkv, Smax, eng.hd), dtype=BF16, device=device) for _ in range(eng.L)]
        self.v = [torch.zeros((B, eng.Hkv, Smax, eng.hd), dtype=BF16, device=device) for _ in range(eng.L)]
        self.Smax = Smax


@torch.no_grad()
def prefill(model, inputs_embeds: torch.Tensor, attention_mask: torch.Tensor, cache: KVCache, pos: torch.Tensor,
            vt_sink=None, decode_rows: int = 8):
    """Runs the prompt through the stack writing K/V into the cache; returns the last-position hidden rows [B, H].
    In an fp8 rollout (rollout_fp8_enabled) the projections run on the fp8 MFMA path over the quantised weights the token loop streams
    (decode_rows: the batch of that loop — its weight set is the one consulted)."""
    eng = model.ensure_packed()
    w8 = prefill_fp8_weights(model, decode_rows)
    if w8 is not None and getattr(eng, "_fp8", None) is None:
        with eng.use_fp8(w8):
            return prefill(model, inputs_embeds, attention_mask, cache, pos, vt_sink=vt_sink, decode_rows=decode_rows)
    B, S, H = inputs_embeds.shape
    kmask = attention_mask.to(torch.uint8).contiguous()
    dev = inputs_embeds.device
    # A few prompts (GRPO: ONE distinct prompt per GPU) give every kernel of the chain a grid of ~144 workgroups on 256 CUs.  The
    # rows are then run as TWO chunks — [0, Sa) and [Sa, S) — on two HIP streams, chunk B one layer behind chunk A: B's queries attend
    # to A's K / V rows of the same layer (one event per layer), both chains are in flight together and fill the chip.  Same kernels
    # on the same rows: every row's result is the one-chunk result (rows of a GEMM are independent, a query visits its key tiles in
    # the same order).  Measured at one prompt of 2180 rows (round 4, tools/decode_probe.py): 16.0 ms against 15.6 ms for one chain —
    # the half-height GEMMs lose what the overlap gains — so it is OPT-IN (BRA_PREFILL_CHUNKS=2), kept for wider prompts.
    two = (dev.type == "cuda" and os.environ.get("BRA_PREFILL_CHUNKS", "1") == "2" and B * S <= 4096 and S >= 1024)
    if two or (os.environ.get("BRA_PREFILL_CHUNKS_FORCE") == "2" and S >= 2):
        Sa = (S // 2 + 63) // 64 * 64 if S >= 256 else max(1, S // 2)
        posm = pos.reshape(B, S)
        ma = SeqMeta(B=B, S=Sa, pos=posm[:, :Sa].reshape(-1).contiguous(), kmask=kmask[:, :Sa].contiguous(), lora_on=model._lora_enabled,
                     max_pos=cache.Smax + 1)
        mb = SeqMeta(B=B, S=S - Sa, pos=posm[:, Sa:].reshape(-1).contiguous(), kmask=kmask, lora_on=model._lora_enabled,
                     max_pos=cache.Smax + 1)
        xe = inputs_embeds.to(BF16)
        xa = xe[:, :Sa].reshape(B * Sa, H).contiguous()
        xb = xe[:, Sa:].reshape(B * (S - Sa), H).contiguous()
        side = None
        if dev.type == "cuda":
            side = getattr(eng, "_prefill_side", None)
            if side is None:
                side = eng._prefill_side = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)
            xb.record_stream(side)
        for li in range(eng.L):
            xa, _ = eng.layer_fwd(li, xa, ma, save=False, kv_out=(cache.k[li], cache.v[li], 0))
            if side is not None:
                ev = torch.cuda.Event()
                ev.record(main)
            with (torch.cuda.stream(side) if side is not None else _nullctx()):
                if side is not None:
                    side.wait_event(ev)
                xb, _ = eng.layer_fwd(li, xb, mb, save=False, kv_out=(cache.k[li], cache.v[li], Sa, vt_sink))
        with (torch.cuda.stream(side) if side is not None else _nullctx()):
            last = (torch.arange(B, device=dev, dtype=torch.int32) * (S - Sa) + (S - Sa - 1))
            out = ops.rmsnorm_fwd(ops.gather_rows(last, xb), eng.norm_w, eng.eps)
        if side is not None:
            main.wait_stream(side)
            out.record_stream(main)
            if vt_sink is not None:
                for t_ in vt_sink:
                    t_.record_stream(main)
        return out
    meta = SeqMeta(B=B, S=S, pos=pos.reshape(-1).contiguous(), kmask=kmask, lora_on=model._lora_enabled,
                   max_pos=cache.Smax + 1)
    x = inputs_embeds.reshape(B * S, H).to(BF16).contiguous()
    for li in range(eng.L):
        x, _ = eng.layer_fwd(li, x, meta, save=False, kv_out=(cache.k[li], cache.v[li], 0, vt_sink))
    last = (torch.arange(B, device=x.device, dtype=torch.int32) * S + (S - 1))
    xl = ops.gather_rows(last, x)
    return ops.rmsnorm_fwd(xl, eng.norm_w, eng.eps)


class _LayerDesc(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_void_p) for n in ("ln1", "ln2", "qn", "kn", "Wqkv", "Wo", "Wgu", "Wd", "A_qkv", "B_qkv", "A_o",
                                                "B_o", "A_gu", "B_gu", "A_d", "B_d")]
                + [(n, ctypes.c_int) for n in ("r_qkv", "r_o", "r_gu", "r_d")]
                + [(n, ctypes.c_float) for n in ("s_qkv", "s_o", "s_gu", "s_d")]
                + [("kc", ctypes.c_void_p), ("vc", ctypes.c_void_p), ("kp", ctypes.c_void_p), ("vtp", ctypes.c_void_p)]
                + [("flags", ctypes.c_int), ("pad_", ctypes.c_int), ("head_packed", ctypes.c_void_p), ("rope_rows", ctypes.c_void_p),
                   ("head_tmax", ctypes.c_void_p)]
                + [(n, ctypes.c_void_p) for n in ("sc_qkv", "sc_o", "sc_gu", "sc_d", "sc_head")])


def _layer_table(n: int):
    """host array of `n` layer records; the record layout is the library's (k_decode.hip `Layer`): sizes must agree"""
    assert ctypes.sizeof(_LayerDesc) == get_lib()._dll.bra_qwen_layer_desc_size(), "layer record out of sync with libbioreason_hip.so"
    return (_LayerDesc * n)()


class DecodeState:
    """Host descriptor table + device workspaces for `bra_qwen_decode_step` (one native call per token)."""

    def __init__(self, model, cache: KVCache, B: int):
        eng: QwenEngine = model.engine
        dev = eng.device
        self.eng, self.cache, self.B = eng, cache, B
        arr = _layer_table(eng.L)
        for i, L in enumerate(eng.layers):
            d = arr[i]
            d.ln1, d.ln2, d.qn, d.kn = L.ln1.data_ptr(), L.ln2.data_ptr(), L.qn.data_ptr(), L.kn.data_ptr()
            d.Wqkv, d.Wo, d.Wgu, d.Wd = L.Wqkv.data_ptr(), L.Wo.data_ptr(), L.Wgu.data_ptr(), L.Wd.data_ptr()
            for g in ("qkv", "o", "gu", "d"):
                G = L.lora[g]
                setattr(d, "A_" + g, G.A.data_ptr() if G is not None else None)
                setattr(d, "B_" + g, G.B.data_ptr() if G is not None else None)
                setattr(d, "r_" + g, G.r_pad if G is not None else 0)
                setattr(d, "s_" + g, G.scaling if G is not None else 0.0)
            d.kc, d.vc = cache.k[i].data_ptr(), cache.v[i].data_ptr()
        self.arr = arr
        nq, nqkv = eng.Nq, eng.Nq + 2 * eng.Nkv

        def buf(n):
            return torch.empty((B, n), dtype=BF16, device=dev)

        self.x, self.xn, self.h, self.hn, self.hid = buf(eng.H), buf(eng.H), buf(eng.H), buf(eng.H), buf(eng.H)
        self.qkv, self.q, self.o = buf(nqkv), buf(nq), buf(nq)
        self.gu, self.act, self.t = buf(2 * eng.F), buf(eng.F), buf(128)
        nch = (cache.Smax + 127) // 128
        self.part_o = torch.empty((B, eng.Hq, nch, eng.hd), dtype=torch.float32, device=dev)
        self.part_ml = torch.empty((B, eng.Hq, nch, 2), dtype=torch.float32, device=dev)
        self.cosT, self.sinT = eng.rope(cache.Smax + 1)

    def step(self, tok, pos, kmask, cur_len: int, lora_on: bool):
        e = self.eng
        get_lib().call("bra_qwen_decode_step", ctypes.addressof(self.arr), e.L, self.B, e.H, e.Hq, e.Hkv, e.hd, e.F,
                       self.cache.Smax, e.eps, e.scale, e.E, e.norm_w, self.cosT, self.sinT, tok, pos, kmask, cur_len,
                       int(lora_on), self.x, self.xn, self.qkv, self.q, self.o, self.h, self.hn, self.gu, self.act, self.t,
                       self.part_o, self.part_ml, self.hid, current_stream(self.x))
        return self.hid


@torch.no_grad()
def rollout_weights(model, rows: int = 8):
    """Per-layer weight set of the fused decode step: LoRA merged into the base weight (W + s B A, rounded to bf16 —
    the merge PEFT's merge_and_unload performs, reason.py:428-446) when adapters are enabled, and gate/up rows
    interleaved in blocks of 8 for the SwiGLU epilogue.  Rebuilt when the adapters or base weights changed."""
    eng: QwenEngine = model.ensure_packed()
    arena = model.arena
    pack = os.environ.get("BRA_DEC_PACK", "1") == "1"
    fold = pack and os.environ.get("BRA_DEC_FOLD", "1") == "1"
    wide = rows > 8             # 9 .. 16 sequences: every projection is packed for 16-column tiles (ops.dec_pack_weights(rows=16))
    fp8 = rollout_fp8_enabled(model) and pack and fold and not wide
    key = (model._packed_sig, model._lora_enabled, pack, fold, wide, fp8, None if arena is None or arena.params is None else (arena.step_count, arena.params._version))
    cached = getattr(eng, "_rollout", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    on = model._lora_enabled
    out = []
    F = eng.F
    for L in eng.layers:
        def merged(W, G):
            if G is None or not on:
                return W
            return ops.gemm_nt(G.B, G.AT, alpha=G.scaling, res=W)
        wgu = merged(L.Wgu, L.lora["gu"])
        wgu_il = wgu.view(2, F // 8, 8, wgu.shape[1]).permute(1, 0, 2, 3).reshape(2 * F, wgu.shape[1]).contiguous()
        rec = {"Wqkv": merged(L.Wqkv, L.lora["qkv"]), "Wo": merged(L.Wo, L.lora["o"]), "Wgu": wgu_il,
               "Wd": merged(L.Wd, L.lora["d"])}
        if pack:
            # fragment order of the M <= 8 streaming projections: contiguous KiB per wave-instruction (k_decgemm.hip)
            # (and, `fold`, the RMSNorm weight of the projection's input multiplied in: the kernel then applies rstd to the reduced
            # products and loads no norm weights / statistics ahead of its MFMAs)
            nws = {"Wqkv": L.ln1 if fold else None, "Wgu": L.ln2 if fold else None, "Wo": None, "Wd": None}
            if fp8:
                # opt-in fp8 rollout weights (BASELINE config 5): e4m3 + one fp32 scale per output row of the MERGED, norm-folded weight;
                # a layer whose shapes the fp8 kernel does not take keeps bf16 (all four projections together)
                q8 = {nm: ops.dec_pack_weights_fp8(rec[nm], act=(nm == "Wgu"), norm_w=nws[nm]) for nm in ("Wqkv", "Wo", "Wgu", "Wd")}
                if all(v is not None for v in q8.values()):
                    for nm, (q, sc) in q8.items():
                        rec[nm + "_q"], rec[nm + "_s"] = q, sc
                    rec["fp8"], rec["folded"] = True, True
                    if fp8_prefill_enabled() and eng.fp8_supported():
                        # the same quantised weights, row-major, for the prompt pass on the fp8 MFMA path (engine._layer_fwd_fp8):
                        # same rule and rounding as the fragment-ordered image above ([gate; up] rows not interleaved there)
                        rec["rm8"] = {"Wqkv": ops.quant_rows_fp8(rec["Wqkv"], colw=L.ln1), "Wo": ops.quant_rows_fp8(rec["Wo"]),
                                      "Wgu": ops.quant_rows_fp8(wgu, colw=L.ln2), "Wd": ops.quant_rows_fp8(rec["Wd"])}
                    for nm in ("Wqkv", "Wo", "Wgu", "Wd"):
                        rec[nm + "_p"] = rec[nm + "_q"]            # (one pointer set for `_packed_ok` / the descriptor table)
                    out.append(rec)
                    continue
            pk = {nm: ops.dec_pack_weights(rec[nm], act=(nm == "Wgu"), norm_w=nws[nm], rows=16 if wide else 8)
                  for nm in ("Wqkv", "Wo", "Wgu", "Wd")}
            if all(v is not None for v in pk.values()):
                rec.update({nm + "_p": v for nm, v in pk.items()})
                rec["folded"] = fold
        out.append(rec)
    eng._rollout = (key, out)
    return out


def fp8_prefill_enabled() -> bool:
    """the prompt pass of an fp8 rollout on the fp8 MFMA path (default with rollout_fp8; BRA_FP8_PREFILL=0: bf16 prompt pass, round 5's form)"""
    return os.environ.get("BRA_FP8_PREFILL", "1") == "1"


def prefill_fp8_weights(model, decode_rows: int):
    """per-layer row-major e4m3 images of the merged policy weights, or None (bf16 rollout, wide decode, a layer the fp8 kernels refuse)"""
    if not (rollout_fp8_enabled(model) and fp8_prefill_enabled()):
        return None
    rw = rollout_weights(model, rows=decode_rows)
    if not all(r.get("rm8") is not None for r in rw):
        return None
    return [r["rm8"] for r in rw]


def rollout_fp8_enabled(model) -> bool:
    """fp8 (e4m3) weights in the token loop: `model.rollout_fp8 = True` (GRPOConfig.rollout_fp8, bench --rollout-fp8) or BRA_ROLLOUT_FP8=1.
    Never the default: the bf16 rollout is the parity-tested reference path."""
    return bool(getattr(model, "rollout_fp8", False)) or os.environ.get("BRA_ROLLOUT_FP8") == "1"


def packed_head_fp8(model):
    """fp8 image of the tied lm_head with the final RMSNorm weight folded in -> (q, scale) or None (311 MB instead of 622 at Qwen3-1.7B)"""
    eng: QwenEngine = model.ensure_packed()
    key = (eng.E.data_ptr(), eng.E._version, tuple(eng.E.shape), eng.norm_w.data_ptr(), eng.norm_w._version)
    cached = getattr(eng, "_head_fp8", None)
    if cached is None or cached[0] != key:
        eng._head_fp8 = (key, ops.dec_pack_weights_fp8(eng.E, out_f32=True, norm_w=eng.norm_w))
    return eng._head_fp8[1]


def packed_head(model):
    """fragment-packed copy of the tied lm_head / embedding matrix for the decode step's logits projection (622 MB at
    Qwen3-1.7B); the matrix is frozen under LoRA training, so the copy is rebuilt only when its storage changes"""
    eng: QwenEngine = model.ensure_packed()
    fold = os.environ.get("BRA_DEC_FOLD", "1") == "1"
    key = (eng.E.data_ptr(), eng.E._version, tuple(eng.E.shape), eng.norm_w.data_ptr(), eng.norm_w._version, fold)
    cached = getattr(eng, "_head_packed", None)
    if cached is None or cached[0] != key:
        eng._head_packed = (key, ops.dec_pack_weights(eng.E, out_f32=True, norm_w=eng.norm_w if fold else None), fold)
    return eng._head_packed[1], eng._head_packed[2]


class FusedDecodeState:
    """Descriptor table + workspaces for `bra_qwen_decode_step_fused` (6 launches per layer, logits included)."""

    def __init__(self, model, cache: KVCache, B: int):
        eng: QwenEngine = model.engine
        dev = eng.device
        self.eng, self.cache, self.B = eng, cache, B
        self.rw = rollout_weights(model, rows=B)
        self.ss_ws, self.nss = _norm_stat_ws(eng, eng.device, rows=B)
        use_packed = self.ss_ws is not None and _packed_ok(B, self.rw)
        arr = _layer_table(eng.L)
        for i, (L, R) in enumerate(zip(eng.layers, self.rw)):
            d = arr[i]
            d.ln1, d.ln2, d.qn, d.kn = L.ln1.data_ptr(), L.ln2.data_ptr(), L.qn.data_ptr(), L.kn.data_ptr()
            _set_proj(d, R, use_packed)
            d.kc, d.vc = cache.k[i].data_ptr(), cache.v[i].data_ptr()
        if use_packed and (B > 8 or os.environ.get("BRA_DEC_PACK_HEAD", "1") == "1"):      # (above 8 rows only the packed head streams)
            h8 = packed_head_fp8(model) if (B <= 8 and all(R_.get("fp8") for R_ in self.rw)) else None
            if h8 is not None:
                self.head_p, self.head_s = h8
                arr[0].head_packed, arr[0].sc_head = self.head_p.data_ptr(), self.head_s.data_ptr()
                arr[0].flags |= 4 | 16
            else:
                self.head_p, head_folded = packed_head(model)
                if self.head_p is not None:
                    arr[0].head_packed = self.head_p.data_ptr()
                    arr[0].flags |= 4 if head_folded else 0
        self.arr = arr

        def buf(n):
            return torch.empty((B, n), dtype=BF16, device=dev)

        self.x, self.h, self.qkv, self.o, self.act = buf(eng.H), buf(eng.H), buf(eng.Nq + 2 * eng.Nkv), buf(eng.Nq), buf(eng.F)
        nch = (cache.Smax + 63) // 64
        self.part_o = torch.empty((B, eng.Hq, nch, eng.hd), dtype=torch.float32, device=dev)
        self.part_ml = torch.empty((B, eng.Hq, nch, 2), dtype=torch.float32, device=dev)
        self.cosT, self.sinT = eng.rope(cache.Smax + 1)

    def step(self, tok, pos, kmask, cur_len: int, logits: torch.Tensor, len_dev=None, embed_done: bool = False):
        """`len_dev` (device int32 [1] holding cur_len): the kernels read the length from memory and `cur_len` only
        sizes the grids, so the call can be captured once and replayed."""
        e = self.eng
        get_lib().call("bra_qwen_decode_step_fused", ctypes.addressof(self.arr), e.L, self.B, e.H, e.Hq, e.Hkv, e.hd, e.F,
                       self.cache.Smax, e.V, e.eps, e.scale, e.E, e.norm_w, self.cosT, self.sinT, tok, pos, kmask, cur_len,
                       len_dev, int(embed_done), self.x, self.qkv, self.o, self.h, self.act, self.ss_ws, self.nss, self.part_o,
                       self.part_ml,
                       logits, current_stream(self.x))


def decode_slots(P: int, C: int) -> int:
    """partial-result slots per (sequence, q-head) of the shared-prefix decode attention: 64-key chunks of the prompt and of the
    completion + the new key"""
    return (P + 63) // 64 + (C + 63) // 64 + 1


class SharedDecodeState:
    """`bra_qwen_decode_step_shared`: R prompts x `copies` sequences; prompt K / V^T are held once per prompt, each
    sequence owns only a completion cache [Hkv, C, hd]."""

    def persist_timed_out(self) -> bool:
        """(host sync) True if a barrier of the persistent step gave up (GridSync::err != 0): the rollout's tokens are invalid"""
        return self.persist is not None and bool(self.persist["sync"].view(torch.int32)[272].item() != 0)

    def __init__(self, model, cache_r: KVCache, vtp, R: int, copies: int, P: int, C: int):
        eng: QwenEngine = model.engine
        dev = eng.device
        B = R * copies
        self.eng, self.R, self.copies, self.P, self.C, self.B = eng, R, copies, P, C, B
        self.rw = rollout_weights(model, rows=B)
        self.kp, self.vtp = cache_r.k, vtp                       # [R,Hkv,P,hd], [R,Hkv,hd,pitch]
        self.kc = [torch.zeros((B, eng.Hkv, C, eng.hd), dtype=BF16, device=dev) for _ in range(eng.L)]
        # attention of the step: "one" = bra_dec_attn_one (k_decattn.hip: MFMA for the completion keys too, transposed completion
        # V cache [B, Hkv, hd, cp], the new key as its own partial);  "both" = bra_dec_attn_both + bra_attn_decode_merge
        # (first generation: row-layout V cache, completion keys on the VALU)
        self.attn_impl = os.environ.get("BRA_DEC_ATTN", "one")
        nslot = decode_slots(P, C)
        if self.attn_impl == "one" and nslot > 256:
            self.attn_impl = "both"
        if self.attn_impl == "one":
            self.cp = (C + 63) // 64 * 64
            self.vc = [torch.zeros((B, eng.Hkv, eng.hd, self.cp), dtype=BF16, device=dev) for _ in range(eng.L)]
        else:
            self.vc = [torch.zeros((B, eng.Hkv, C, eng.hd), dtype=BF16, device=dev) for _ in range(eng.L)]
        self.vt_pitch = vtp[0].shape[-1]
        self.ss_ws, self.nss = _norm_stat_ws(eng, dev, rows=B)
        use_packed = self.ss_ws is not None and _packed_ok(B, self.rw)
        arr = _layer_table(eng.L)
        for i, (L, Rw) in enumerate(zip(eng.layers, self.rw)):
            d = arr[i]
            d.ln1, d.ln2, d.qn, d.kn = L.ln1.data_ptr(), L.ln2.data_ptr(), L.qn.data_ptr(), L.kn.data_ptr()
            _set_proj(d, Rw, use_packed)
            d.kc, d.vc = self.kc[i].data_ptr(), self.vc[i].data_ptr()
            d.kp, d.vtp = self.kp[i].data_ptr(), self.vtp[i].data_ptr()
        if use_packed and (B > 8 or os.environ.get("BRA_DEC_PACK_HEAD", "1") == "1"):
            h8 = packed_head_fp8(model) if (B <= 8 and all(R_.get("fp8") for R_ in self.rw)) else None
            if h8 is not None:
                self.head_p, self.head_s = h8
                arr[0].head_packed, arr[0].sc_head = self.head_p.data_ptr(), self.head_s.data_ptr()
                arr[0].flags |= 4 | 16
            else:
                self.head_p, head_folded = packed_head(model)
                if self.head_p is not None:
                    arr[0].head_packed = self.head_p.data_ptr()
                    arr[0].flags |= 4 if head_folded else 0
        # (cos | sin) rows of every sequence's current position, refreshed once per token by bra_advance_counters
        self.rope_rows = torch.zeros((B, eng.hd), dtype=torch.float32, device=dev) if os.environ.get("BRA_DEC_ROPE_ROWS", "1") == "1" else None
        if self.rope_rows is not None:
            arr[0].rope_rows = self.rope_rows.data_ptr()
        self.arr = arr

        def buf(n):
            return torch.empty((B, n), dtype=BF16, device=dev)

        self.x, self.h, self.qkv, self.o, self.act = buf(eng.H), buf(eng.H), buf(eng.Nq + 2 * eng.Nkv), buf(eng.Nq), buf(eng.F)
        nch = (P + 63) // 64 + (C + 63) // 64 + 1
        self.part_o = torch.empty((B, eng.Hq, nch, eng.hd), dtype=torch.float32, device=dev)
        self.part_ml = torch.empty((B, eng.Hq, nch, 2), dtype=torch.float32, device=dev)
        self.cosT, self.sinT = eng.rope(P + C + 1)
        # persistent form of the layer loop (k_persist.hip): ONE launch for all decoder layers, bit-identical to the launched path.
        # BRA_DEC_PERSIST: "0" (default) launched kernels; "1" / "2" / "3" opt into the persistent step with prefetch level 0 / 1 / 2
        # where the kernel is instantiated for the shape (<= 8 sequences, packed + folded weights, one CU per workgroup) — measured
        # slower than the launched step.  `sync` is zeroed here, once per rollout: the kernel's error word is
        # sticky across the token steps, so persist_timed_out() after the loop sees a timeout of ANY step
        self.persist = None
        mode = os.environ.get("BRA_DEC_PERSIST", "0")
        if mode != "0" and not getattr(get_lib(), "debug", False):
            raise RuntimeError("BRA_DEC_PERSIST needs the debug library (the persistent decode step is not part of the product ABI): "
                               "call bioreason_amd._lib.use_debug_library() first (`make -C bioreason_amd/csrc debug`)")
        if mode != "0" and self.attn_impl == "one" and use_packed and B <= 8 and all(Rw.get("folded") and not Rw.get("fp8") for Rw in self.rw) and dev.type == "cuda":
            tab = torch.tensor([[Rw["Wqkv_p"].data_ptr(), Rw["Wo_p"].data_ptr(), Rw["Wgu_p"].data_ptr(), Rw["Wd_p"].data_ptr(),
                                 L.qn.data_ptr(), L.kn.data_ptr(), self.kp[i].data_ptr(), self.vtp[i].data_ptr(),
                                 self.kc[i].data_ptr(), self.vc[i].data_ptr()] for i, (L, Rw) in enumerate(zip(eng.layers, self.rw))],
                               dtype=torch.int64)
            assert tab.shape[1] * 8 == get_lib()._dll.bra_persist_layer_desc_size()
            self.persist = {"table": tab.contiguous(), "sync": torch.zeros(2048, dtype=torch.uint8, device=dev),       # (host table)
                            "prefetch": max(0, int(mode) - 1), "ok": None}

    def step(self, tok, pos, pmask, t: int, logits: torch.Tensor, t_dev=None, embed_done: bool = False):
        e = self.eng
        if self.persist is not None and self.persist["ok"] is not False:
            ps = self.persist
            rc = get_lib().call_rc("bra_qwen_decode_step_persist", ctypes.addressof(self.arr), ps["table"].data_ptr(), e.L, self.R, self.copies, e.H,
                                   e.Hq, e.Hkv, e.hd, e.F, self.P, self.vt_pitch, self.C, self.cp, e.V, e.eps, e.scale, e.E, e.norm_w,
                                   self.cosT, self.sinT, tok, pos, pmask, t, t_dev, int(embed_done), self.x, self.qkv, self.o, self.h,
                                   self.act, self.ss_ws, self.nss, self.part_o, self.part_ml, self.part_o.shape[2], logits, ps["sync"],
                                   ps["prefetch"], int(os.environ.get("BRA_DEC_PERSIST_STOP", "0")), 50000, current_stream(self.x))
            if rc == 0:
                ps["ok"] = True
                return
            ps["ok"] = False            # BRA_ERR_UNSUPPORTED (shape / CU count): launched kernels from here on
        if self.attn_impl == "one":
            get_lib().call("bra_qwen_decode_step_one", ctypes.addressof(self.arr), e.L, self.R, self.copies, e.H, e.Hq, e.Hkv,
                           e.hd, e.F, self.P, self.vt_pitch, self.C, self.cp, e.V, e.eps, e.scale, e.E, e.norm_w, self.cosT,
                           self.sinT, tok, pos, pmask, t, t_dev, int(embed_done), self.x, self.qkv, self.o, self.h, self.act,
                           self.ss_ws, self.nss, self.part_o, self.part_ml, self.part_o.shape[2], logits, current_stream(self.x))
            return
        get_lib().call("bra_qwen_decode_step_shared", ctypes.addressof(self.arr), e.L, self.R, self.copies, e.H, e.Hq, e.Hkv,
                       e.hd, e.F, self.P, self.vt_pitch, self.C, e.V, e.eps, e.scale, e.E, e.norm_w, self.cosT, self.sinT, tok,
                       pos, pmask, t, t_dev, int(embed_done), self.x, self.qkv, self.o, self.h, self.act, self.ss_ws, self.nss,
                       self.part_o,
                       self.part_ml, logits, current_stream(self.x))


def _set_proj(d, R, use_packed: bool):
    """projection weight pointers of one layer record: the fragment-packed copies when the streaming kernels can take them"""
    sfx = "_p" if use_packed else ""
    d.Wqkv, d.Wo, d.Wgu, d.Wd = (R["Wqkv" + sfx].data_ptr(), R["Wo" + sfx].data_ptr(), R["Wgu" + sfx].data_ptr(),
                                  R["Wd" + sfx].data_ptr())
    d.flags = (1 | (2 if R.get("folded") else 0)) if use_packed else 0
    if use_packed and R.get("fp8"):                   # e4m3 images + row scales (bra_dec_gemm2_fp8)
        d.flags |= 8
        d.sc_qkv, d.sc_o, d.sc_gu, d.sc_d = R["Wqkv_s"].data_ptr(), R["Wo_s"].data_ptr(), R["Wgu_s"].data_ptr(), R["Wd_s"].data_ptr()


def _packed_ok(B: int, rw) -> bool:
    """the streaming projections (bra_dec_gemm2) take up to 16 rows; above 8 they need the norm-folded packed weights"""
    if not all("Wqkv_p" in R for R in rw):
        return False
    return B <= 8 or (B <= 16 and all(R.get("folded") for R in rw))


def _norm_stat_ws(eng, dev, rows: int = 8):
    """zeroed workspace [2][8 | 16][nss] of RMSNorm partial sums of squares for the bra_dec_gemm2 projections (None, 0: the
    hidden size has more column workgroups than the 256 partials a consumer folds -> first-generation kernels)"""
    nblk = eng.H // 8 if (eng.H % 8 == 0 and (eng.H + 15) // 16 < 256 and eng.H // 8 <= 256) else (eng.H + 15) // 16
    nss = (nblk + 31) // 32 * 32
    if nss > 256 or os.environ.get("BRA_DEC_GEMM_V1") == "1":
        return None, 0
    return torch.zeros((2, 16 if rows > 8 else 8, nss), dtype=torch.float32, device=dev), nss


def _uniform_groups(prompt_alias):
    """-> (R, copies) if the aliases describe R contiguous groups of equal size led by their first row, else None"""
    B = len(prompt_alias)
    reps = sorted(set(prompt_alias))
    R = len(reps)
    if R == 0 or B % R:
        return None
    c = B // R
    return (R, c) if all(prompt_alias[b] == (b // c) * c for b in range(B)) else None


@torch.no_grad()
def decode_step(model, tok: torch.Tensor, cache: KVCache, kmask: torch.Tensor, pos: torch.Tensor, cur_len: int):
    """(Python-orchestrated variant, kept for tests) tok int32 [B] -> final hidden [B, H]."""
    eng: QwenEngine = model.engine
    on = model._lora_enabled
    B = tok.shape[0]
    dev = tok.device
    cosT, sinT = eng.rope(cache.Smax + 1)
    x = torch.empty((B, eng.H), dtype=BF16, device=dev)
    ops.embed_scatter_fwd(tok, None, eng.E, None, x)
    q = torch.empty((B, 1, eng.Hq, eng.hd), dtype=BF16, device=dev)
    for li, L in enumerate(eng.layers):
        xn = ops.rmsnorm_fwd(x, L.ln1, eng.eps)
        qkv, _ = eng._lora_fwd(xn, L.Wqkv, L.lora["qkv"], on)
        kc, vc = cache.k[li], cache.v[li]
        ops.qk_norm_rope_fwd(qkv, L.qn, L.kn, cosT, sinT, pos, 1, eng.Hq, eng.Hkv, eng.hd, eng.eps, 1.0, q,
                             kc.permute(0, 2, 1, 3), vc.permute(0, 2, 1, 3), cur_len)
        o = ops.attn_decode(q.view(B, eng.Hq, eng.hd), kc, vc, kmask, cur_len + 1, eng.scale)
        h, _ = eng._lora_fwd(o, L.Wo, L.lora["o"], on, res=x)
        hn = ops.rmsnorm_fwd(h, L.ln2, eng.eps)
        gu, _ = eng._lora_fwd(hn, L.Wgu, L.lora["gu"], on)
        act = ops.swiglu_fwd(gu)
        x, _ = eng._lora_fwd(act, L.Wd, L.lora["d"], on, res=h)
    return ops.rmsnorm_fwd(x, eng.norm_w, eng.eps)


MAX_DECODE_ROWS = 16            # sequences per launch of the streaming decode kernels (bra_dec_gemm2: one 16-row MFMA tile)


def _row_chunks(B: int, prompt_alias):
    """[lo, hi) row ranges of at most MAX_DECODE_ROWS rows: consecutive whole prompt groups (rows with the same alias) where they
    fit, a larger group in pieces"""
    if prompt_alias is None:
        runs = [(i, i + 1) for i in range(B)]
    else:
        al = [int(a) for a in prompt_alias]
        runs, lo = [], 0
        for i in range(1, B + 1):
            if i == B or al[i] != al[lo]:
                runs.append((lo, i))
                lo = i
    chunks, cur = [], None
    for lo, hi in runs:
        while hi - lo > MAX_DECODE_ROWS:                       # a group of more than 16 copies: pieces of 16
            if cur is not None:
                chunks.append(cur)
                cur = None
            chunks.append((lo, lo + MAX_DECODE_ROWS))
            lo += MAX_DECODE_ROWS
        if cur is not None and hi - cur[0] <= MAX_DECODE_ROWS:
            cur = (cur[0], hi)
        else:
            if cur is not None:
                chunks.append(cur)
            cur = (lo, hi)
    if cur is not None:
        chunks.append(cur)
    return chunks


def _generate_in_row_chunks(model, inputs_embeds, attention_mask, prompt_alias, force_tokens, eos_schedule, seed, kw):
    """`kw`: EVERY other keyword of generate() (built from its locals, so a keyword added later is forwarded too)"""
    B = inputs_embeds.shape[0]
    trace = kw.pop("trace_logits", None)
    traces = []
    eos_ = kw["eos_token_id"]
    eos0 = (list(eos_)[0] if len(eos_) else None) if isinstance(eos_, (list, tuple)) else eos_
    pad = int(kw["pad_token_id"]) if kw["pad_token_id"] is not None else (int(eos0) if eos0 is not None else 0)
    outs = []
    for ci, (lo, hi) in enumerate(_row_chunks(B, prompt_alias)):
        alias = None
        if prompt_alias is not None:
            al = [int(a) for a in prompt_alias[lo:hi]]
            first = {}
            alias = [first.setdefault(a, i) for i, a in enumerate(al)]          # re-based: index of the group's first row in the chunk
        outs.append(generate(model, inputs_embeds[lo:hi], attention_mask[lo:hi], prompt_alias=alias,
                             force_tokens=None if force_tokens is None else force_tokens[lo:hi],
                             eos_schedule=None if eos_schedule is None else eos_schedule[lo:hi].contiguous(),
                             seed=seed + 7919 * ci,                               # rows restart at 0 in every chunk: distinct draw streams
                             trace_logits=None if trace is None else traces.append([]) or traces[-1],
                             **kw))
    if trace is not None:                                                          # per step: the chunks' logits stacked back into [B, V]
        for t_ in range(max(len(tr) for tr in traces)):
            trace.append(torch.cat([tr[min(t_, len(tr) - 1)] for tr in traces], 0))
    T = max(o.shape[1] for o in outs)
    full = torch.full((B, T), pad, dtype=torch.long, device=inputs_embeds.device)
    r = 0
    for o in outs:
        full[r:r + o.shape[0], :o.shape[1]] = o
        r += o.shape[0]
    return full


@torch.no_grad()
def generate(model, inputs_embeds: torch.Tensor, attention_mask: torch.Tensor, max_new_tokens: int = 20,
             do_sample: bool = False, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0,
             eos_token_id: Optional[int] = None, pad_token_id: Optional[int] = None, seed: int = 0,
             check_every: int = 16, return_full_length: bool = False,
             force_tokens: Optional[torch.Tensor] = None, native_step: bool = True,
             decode_impl: str = "fused", prompt_alias=None, use_graph: Optional[bool] = None,
             shared_prefix_decode: bool = True, profile: Optional[dict] = None, loop_events: Optional[list] = None,
             eos_schedule: Optional[torch.Tensor] = None, trace_logits: Optional[list] = None) -> torch.Tensor:
    """`force_tokens` [B, max_new_tokens] (optional): teacher forcing — the model's own choice is still recorded
    in the output, but the given token is fed back (used to compare decodes position by position).
    `use_graph`: True = sample + decode step + counter update are captured once in a hipGraph and replayed per token
    (a step is ~180 dependent launches of a few microseconds each; a slow host issuing them one by one makes the
    rollout launch-bound); False = eager issue; None (default) = measured once per engine: eager unless the host
    cannot stay ahead of the device.
    `eos_token_id` may be an int or a list of up to two ids (HF stops a row on any listed id; Qwen3's generation_config
    lists two); the first one is the pad default.  `eos_schedule` int32 [B] (benchmarks / tests with random-init weights):
    row b is made to draw the first EOS id at step eos_schedule[b].
    `trace_logits` (tests): a list that receives a copy of the fp32 logits [B, V] after every decode step (eager issue only)."""
    _call_kw = dict(locals())                      # every argument of this call, by name (forwarded whole by the row-chunk path)
    eng = model.ensure_packed()
    B, P, H = inputs_embeds.shape
    dev = inputs_embeds.device
    if B > MAX_DECODE_ROWS and native_step and decode_impl == "fused":
        # the streaming decode kernels take up to 16 sequences per launch (one MFMA tile of rows); the reference's
        # per_device_train_batch_size is free (grpo_config.py), so larger batches run as consecutive row chunks — whole prompt
        # groups where they fit — each with its own prefill, cache and token loop
        own = ("model", "inputs_embeds", "attention_mask", "prompt_alias", "force_tokens", "eos_schedule", "seed", "eng", "B", "P", "H", "dev")
        return _generate_in_row_chunks(model, inputs_embeds, attention_mask, prompt_alias, force_tokens, eos_schedule, seed,
                                       {k: v for k, v in _call_kw.items() if k not in own})
    import time as _time
    _t0 = [_time.perf_counter()]

    def _tick(name):
        # `profile` (diagnostics only): host-synchronised wall time of the phases of this call, in ms
        if profile is not None:
            if dev.type == "cuda":
                torch.cuda.synchronize()
            now = _time.perf_counter()
            profile[name] = profile.get(name, 0.0) + (now - _t0[0]) * 1e3
            _t0[0] = now

    eos2 = -1
    if isinstance(eos_token_id, (list, tuple)):
        ids_ = list(dict.fromkeys(int(e) for e in eos_token_id))
        if len(ids_) > 2:
            raise NotImplementedError("at most two eos_token_id values are supported by the device-side sampler")
        eos2 = ids_[1] if len(ids_) > 1 else -1
        eos_token_id = ids_[0] if ids_ else None
    eos = -1 if eos_token_id is None else int(eos_token_id)
    if eos_schedule is not None:
        assert eos >= 0 and eos_schedule.dtype == torch.int32 and eos_schedule.numel() == inputs_embeds.shape[0]
    pad = int(pad_token_id) if pad_token_id is not None else (eos if eos >= 0 else 0)
    Smax = P + max_new_tokens
    am = attention_mask.to(torch.long)
    pos_prompt = (am.cumsum(-1) - 1).masked_fill(am == 0, 0).to(torch.int32)      # TF:generation/utils.py:763-765
    kmask = torch.ones((B, Smax), dtype=torch.uint8, device=dev)
    kmask[:, :P] = am.to(torch.uint8)
    shared = None
    grp = _uniform_groups(prompt_alias) if prompt_alias is not None else None
    # (more than 16 query rows per (prompt, kv-head) — Qwen3-4B: 8 rollouts x G = 4 — run as virtual prompts of <= 16 rows each in
    #  bra_dec_attn_one; the first-generation attention keeps the 16-row limit)
    G_ = eng.Hq // eng.Hkv
    # (SharedDecodeState falls back from "one" to the first-generation attention beyond 256 partial slots — P + C above ~16 K — and
    #  that one keeps the 16-row limit: such a batch takes the per-copy decode instead of failing in the kernel)
    one_ok = os.environ.get("BRA_DEC_ATTN", "one") == "one" and decode_slots(P, max_new_tokens) <= 256
    rows_ok = grp is not None and (grp[1] * G_ <= 16 or (one_ok and G_ <= 16
                                                          and any(grp[1] % s_ == 0 and (grp[1] // s_) * G_ <= 16 for s_ in range(1, grp[1] + 1))))
    use_shared = (grp is not None and shared_prefix_decode and native_step and decode_impl == "fused" and eng.hd >= 64 and rows_ok)
    if prompt_alias is None:
        cache = KVCache(eng, B, Smax, dev)
        hid = prefill(model, inputs_embeds, attention_mask, cache, pos_prompt, decode_rows=B)
    else:
        # identical prompts (GRPO: the G copies of a prompt, grpo_trainer.py:107-116) are prefetched once: the rows of a
        # batched forward are independent, so the K/V rows and the last hidden state of a copy equal its representative's
        reps = sorted(set(prompt_alias))
        where = {r: i for i, r in enumerate(reps)}
        sel = torch.tensor(reps, device=dev)
        gmap = torch.tensor([where[a] for a in prompt_alias], device=dev)
        if use_shared:
            # decode reads ONE copy of the prompt K / V^T per prompt (bra_dec_attn_shared); no per-copy replication
            cache_r = KVCache(eng, len(reps), P, dev)
            vtp = []
            hid_r = prefill(model, inputs_embeds[sel], attention_mask[sel], cache_r, pos_prompt[sel], vt_sink=vtp, decode_rows=B)
            shared = SharedDecodeState(model, cache_r, vtp, grp[0], grp[1], P, max_new_tokens)
            pmask = am[sel].to(torch.uint8).contiguous()
            cache = None
        else:
            cache_r = KVCache(eng, len(reps), Smax, dev)
            hid_r = prefill(model, inputs_embeds[sel], attention_mask[sel], cache_r, pos_prompt[sel], decode_rows=B)
            cache = KVCache.__new__(KVCache)
            cache.Smax = Smax
            cache.k = [t.index_select(0, gmap) for t in cache_r.k]
            cache.v = [t.index_select(0, gmap) for t in cache_r.v]
            del cache_r
        hid = hid_r.index_select(0, gmap).contiguous()
    next_pos = (pos_prompt[:, -1] + 1).contiguous()                               # TF:generation/utils.py:979-984
    _tick("prefill")

    tokens = torch.full((B, max_new_tokens), pad, dtype=torch.int32, device=dev)
    finished = torch.zeros((B,), dtype=torch.uint8, device=dev)
    cur = torch.empty((B,), dtype=torch.int32, device=dev)
    step_t = torch.zeros((1,), dtype=torch.int32, device=dev)
    logits = torch.empty((B, eng.V), dtype=torch.float32, device=dev)
    n_done = max_new_tokens
    kk = (min(top_k, 64) if top_k > 0 else 64) if do_sample else 1
    sample_ws = torch.empty((2 * B * 64 * kk,), dtype=torch.float32, device=dev) if eng.V >= 4096 else None
    fused = native_step and decode_impl == "fused"
    state = None
    if shared is None and native_step:
        state = FusedDecodeState(model, cache, B) if fused else DecodeState(model, cache, B)
    ops.gemm_nt(hid, eng.E, out=logits, out_f32=True)
    len_t = torch.full((1,), P, dtype=torch.int32, device=dev)          # device-side cur_len of the fused step
    # graph_mode: True / False as requested, None = decide by measurement (see below); graph_ok: replay is possible at all
    graph_ok = dev.type == "cuda" and fused and force_tokens is None and max_new_tokens > 2 and trace_logits is None
    graph_mode = use_graph if use_graph is None else bool(use_graph)
    dstate = shared if shared is not None else (state if fused else None)
    # the drawing wave of the sampler also gathers x = E[token] and its RMSNorm statistic for the fused step (not under
    # teacher forcing, where the token fed back is not the sampled one)
    # sampler over tile maxima (ops.sample_tiles): the lm_head epilogue of the fused step leaves the maximum of every 16-column
    # tile of the logits in `tmax`; the top-k stage then scans V / 16 maxima + 16 k logits instead of V logits (same tokens)
    ntile = (eng.V + 15) // 16
    use_tiles = (dstate is not None and os.environ.get("BRA_SAMPLE_TILES", "1") == "1" and B <= 64 and ntile <= 8 * 4096
                 and (not do_sample or 1 <= top_k <= 64))
    tmax = None
    if use_tiles:
        tmax = torch.empty((B, ntile), dtype=torch.float32, device=dev)
        dstate.arr[0].head_tmax = tmax.data_ptr()
        ops.tile_max(logits, tmax)
        sample_ws = torch.empty((2 * B * 8 * kk,), dtype=torch.float32, device=dev)
    fuse_embed = (dstate is not None and dstate.ss_ws is not None and force_tokens is None and sample_ws is not None
                  and B <= 16 and (not do_sample or 1 <= top_k <= 64))     # (top_k = 0 / > 64: the general sampler, no fused gather)
    rope_args = None
    if shared is not None and getattr(shared, "rope_rows", None) is not None:
        rope_args = (shared.cosT, shared.sinT, eng.hd, shared.rope_rows)
        ops.rope_rows(shared.cosT, shared.sinT, next_pos, eng.hd, shared.rope_rows)
    # issued launch by launch, the shared-prefix loop needs no device-side counters at all: the step index is a launch argument of
    # the sampler too, and its drawing wave leaves the (cos | sin) rows of the positions the step that follows rotates with —
    # the bra_advance_counters launch per token is gone (a replayed graph keeps the counters: its launch arguments are frozen)
    fuse_adv = use_tiles and rope_args is not None
    pos0 = next_pos.clone() if fuse_adv else None
    counters_at = [0]                       # the step index the device-side counters (step_t, len_t, next_pos) hold

    def sample_(t_host: Optional[int] = None):
        """the draw of step t.  `t_host` given: launch-by-launch issue (the index travels as an argument where the kernels take it)"""
        host_step = t_host is not None and fuse_adv
        fin = finished if eos >= 0 else None
        emb = (eng.E, dstate.x, dstate.ss_ws[0]) if fuse_embed else None
        if use_tiles:
            st = t_host if host_step else step_t
            if eos_schedule is not None:
                ops.force_token_tiles(logits, eos, st, eos_schedule, tmax)
            ops.sample_tiles(logits, tmax, temperature, top_k, top_p, do_sample, seed, st, fin, pad, cur, None, eos_id=eos,
                             eos_id2=eos2, tokens_out=tokens, ws=sample_ws, embed=emb,
                             advance=(pos0, next_pos, rope_args[0], rope_args[1], rope_args[2], rope_args[3]) if host_step else None)
            return
        if eos_schedule is not None:
            ops.force_token(logits, eos, step_t, eos_schedule)
        ops.sample(logits, temperature, top_k, top_p, do_sample, seed, step_t, fin, pad,
                   cur, None, eos_id=eos, eos_id2=eos2, tokens_out=tokens, ws=sample_ws, embed=emb)

    def advance_(t_grid: int, exact_t: bool = False):
        """one fused decode step.  Under graph replay the kernels take the step index from step_t / len_t and `t_grid` only sizes
        grids; issued launch by launch (`exact_t`: t_grid IS the step index) the shared-prefix step gets it as a launch argument —
        the attention kernels then start without the dependent scalar load of the device-side counter"""
        if shared is not None:
            shared.step(cur, next_pos, pmask, t_grid, logits, t_dev=None if exact_t else step_t, embed_done=fuse_embed)
        else:
            state.step(cur, next_pos, kmask, P + t_grid, logits, len_dev=len_t, embed_done=fuse_embed)
        if exact_t and fuse_adv:
            return                          # (the next draw moves the positions on)
        ops.advance_counters(next_pos, step_t, len_t, rope=rope_args)
        counters_at[0] += 1

    def sync_counters(tt: int):
        """device-side counters <- step tt (before a graph capture that follows launch-by-launch steps)"""
        if fuse_adv and counters_at[0] != tt:
            step_t.fill_(tt)
            len_t.fill_(P + tt)
            next_pos.copy_(pos0 + tt)
            ops.rope_rows(rope_args[0], rope_args[1], next_pos, eng.hd, rope_args[3])
            counters_at[0] = tt
    _tick("decode_setup")
    # `loop_events` (measurement): a HIP-event pair on the launch stream around the token loop of THIS call is appended — bench.py
    # times the loop inside its timed steps with it (no host synchronisation, unlike `profile`)
    ev_loop = None
    if loop_events is not None and dev.type == "cuda":
        ev_loop = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev_loop[0].record()
    last = max_new_tokens - 1                 # index of the final draw (no decode step follows it)
    t = 0
    alive = True                              # False once every row has emitted EOS

    def eos_stop(tt: int) -> bool:
        return eos >= 0 and force_tokens is None and (tt + 1) % check_every == 0 and bool(finished.all().item())

    def eager_steps(n: int) -> bool:
        """up to n (draw, decode step) pairs issued launch by launch; False if the EOS check ended the rollout"""
        nonlocal t, n_done, hid
        for _ in range(n):
            if t >= last:
                return True
            sample_(t)
            if eos_stop(t):
                n_done = t + 1
                return False
            if force_tokens is not None:
                cur.copy_(force_tokens[:, t].to(torch.int32))
            if fused:
                advance_(t, exact_t=True)
                if trace_logits is not None:
                    trace_logits.append(logits.clone())
            else:
                if state is not None:
                    hid = state.step(cur, next_pos, kmask, P + t, model._lora_enabled)
                else:
                    hid = decode_step(model, cur, cache, kmask, next_pos, P + t)
                ops.gemm_nt(hid, eng.E, out=logits, out_f32=True)
                next_pos.add_(1)
                step_t.add_(1)
            t += 1
        return True

    graph = None
    want_graph = graph_mode
    if want_graph is None and graph_ok:
        # auto: replay a hipGraph only if the host cannot issue the ~180 launches of a step faster than the device runs
        # them (measured once per engine over 8 eager steps).  On a fast host eager issue is the quicker of the two:
        # graph nodes execute with larger gaps than back-to-back stream launches.
        want_graph = getattr(eng, "_rollout_use_graph", None)
        if want_graph is None and last >= 24:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            alive = eager_steps(2)                       # warm: one-time kernel attribute set-up, allocator
            torch.cuda.synchronize()
            h0 = _time.perf_counter()
            e0.record()
            if alive:
                alive = eager_steps(8)
            host_s = _time.perf_counter() - h0
            e1.record()
            e1.synchronize()
            vote = host_s > 0.8 * e0.elapsed_time(e1) * 1e-3
            # one slow probe must not lock a whole run into replay (a capture per rollout costs ~90 ms at Qwen3-1.7B; a transient
            # on the host — first-use set-up, a page-cache miss — was seen to inflate a single probe 4x): eager is settled by one
            # vote, replay needs two consecutive ones; an undecided rollout continues launch by launch
            if not vote:
                eng._rollout_use_graph = False
            else:
                eng._rollout_graph_votes = getattr(eng, "_rollout_graph_votes", 0) + 1
                if eng._rollout_graph_votes >= 2:
                    eng._rollout_use_graph = True
            want_graph = bool(getattr(eng, "_rollout_use_graph", False))
            eng._rollout_probe_ms = (host_s * 1e3 / 8, e0.elapsed_time(e1) / 8)      # host issue vs device time per step
            if profile is not None:
                profile["auto_host_ms_per_step"] = host_s * 1e3 / 8
                profile["auto_gpu_ms_per_step"] = e0.elapsed_time(e1) / 8
    if alive and want_graph and graph_ok and t < last:
        if t == 0:
            alive = eager_steps(1)                       # eager first step: one-time kernel attribute set-up happens here
        if alive and t < last:
            sync_counters(t)
            graph = torch.cuda.CUDAGraph()
            # thread_local: other threads of the process (the RCCL watchdog of a data-parallel run) keep making HIP calls
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                sample_()
                advance_(max_new_tokens - 1)
            _tick("graph_capture")
            while t < last:
                graph.replay()
                if eos >= 0 and (t + 1) % check_every == 0 and bool(finished.all().item()):
                    n_done = t + 1
                    alive = False
                    break
                t += 1
    elif alive:
        alive = eager_steps(last - t)
    if alive:
        sample_(t if graph is None else None)
    if ev_loop is not None:
        ev_loop[1].record()
        loop_events.append((ev_loop[0], ev_loop[1], (max_new_tokens if alive else n_done) - 1))
    _tick("decode_loop")
    if shared is not None and shared.persist is not None and shared.persist["ok"] and shared.persist_timed_out():
        raise RuntimeError("bioreason_amd: a grid barrier of the persistent decode step timed out (GridSync::err set); the rollout is "
                           "invalid. BRA_DEC_PERSIST=0 selects the launched kernels.")
    if graph is not None:
        graph.reset()            # release the executable graph here, not whenever the cyclic GC finds the closures
        del graph
        _tick("graph_reset")
    out = tokens[:, :n_done]
    if eos >= 0 and not return_full_length and force_tokens is None:
        # HF stops right after the step in which the last row finished: trim the all-pad tail we may have produced
        is_eos = (out == eos) | (out == eos2) if eos2 >= 0 else out == eos
        has = is_eos.any(dim=1)
        first = torch.where(has, is_eos.int().argmax(dim=1), torch.full_like(has, out.shape[1] - 1, dtype=torch.long))
        out = out[:, : int(first.max().item()) + 1]
    return out.to(torch.long)
