"""Thin tensor-level wrappers over the C-ABI kernels (one function per entry point).

PyTorch supplies device memory and the stream; every arithmetic step is a HIP
kernel from ``libbioreason_hip.so``.  No function here falls back to torch math.
"""
from __future__ import annotations

import math
import os
from typing import Optional, Tuple

import torch

from ._lib import current_stream, get_lib

BF16 = torch.bfloat16


def _ld(t: torch.Tensor) -> int:
    """leading dimension (elements) of a 2-D-viewable tensor with contiguous last dim"""
    assert t.stride(-1) == 1, "last dim must be contiguous"
    return t.stride(-2) if t.dim() >= 2 else t.shape[-1]


def _rows(t: torch.Tensor) -> int:
    return t.numel() // t.shape[-1]


def _as2d(t: torch.Tensor) -> torch.Tensor:
    return t.reshape(-1, t.shape[-1]) if t.dim() != 2 else t


# --------------------------------------------------------------------------- GEMM
def gemm_nt(
    a: torch.Tensor,
    b: torch.Tensor,
    *,
    a2: Optional[torch.Tensor] = None,
    b2: Optional[torch.Tensor] = None,
    bias: Optional[torch.Tensor] = None,
    res: Optional[torch.Tensor] = None,
    out: Optional[torch.Tensor] = None,
    alpha: float = 1.0,
    out_f32: bool = False,
    accumulate: bool = False,
) -> torch.Tensor:
    """out[M,N] = alpha * (a[M,K] @ b[N,K]^T + a2[M,K2] @ b2[N,K2]^T) (+bias) (+res)"""
    a = _as2d(a)
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and a.dtype == BF16 and b.dtype == BF16
    K2 = 0
    if a2 is not None:
        a2 = _as2d(a2)
        K2 = a2.shape[1]
        assert b2 is not None and b2.shape == (N, K2) and a2.shape[0] == M
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32 if out_f32 else BF16, device=a.device)
    else:
        assert out.shape == (M, N) and out.dtype == (torch.float32 if out_f32 else BF16)
    if res is not None:
        res = _as2d(res)
        assert res.shape == (M, N) and res.dtype == BF16
    prof = GEMM_PROFILE
    timed = prof is not None and a.is_cuda and prof.wants(M, N, K, K2)
    if timed:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    get_lib().call(
        "bra_gemm_bf16_nt", a, _ld(a), b, _ld(b), a2, _ld(a2) if a2 is not None else 0, b2,
        _ld(b2) if b2 is not None else 0, K2, out, _ld(out), M, N, K, alpha, bias, res,
        _ld(res) if res is not None else 0, int(out_f32), int(accumulate), current_stream(a),
    )
    if timed:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        # flops and algorithmic bytes (every operand once: A, B, LoRA pair, residual, output)
        nbytes = 2.0 * (M * K + N * K + (M + N) * K2 + (M * N if res is not None else 0)) + M * N * (4.0 if out_f32 else 2.0)
        prof.records.append((2.0 * M * N * (K + K2), e0, e1, nbytes, (M, N, K, K2, res is not None)))
    return out


class GemmProfile:
    """bench.py hook: HIP events (on the launch stream) around every large-M launch of the dominant kernel."""

    def __init__(self, min_m: int = 1024, dominant_only: bool = True):
        self.min_m = min_m
        self.dominant_only = dominant_only
        self.records = []

    def wants(self, M: int, N: int, K: int, K2: int) -> bool:
        """dominant_only: exactly the launches k_gemm.hip's pick_variant() sends to the LDS-DMA MFMA kernels (gemm_ring_kernel
        256x256 / gemm_glds_kernel 256 | 192 | 128 x 128 / — late round 4, the same calls — gemm_w4_kernel, MFMA-bound); otherwise every launch with M >= min_m (that also counts the N = 32
        LoRA projections, which are HBM-bound reads of the activations)"""
        if not self.dominant_only:
            return M >= self.min_m
        if K % 64 or K2 % 64:
            return False
        t = ((M + 255) // 256) * ((N + 255) // 256)               # k_gemm.hip pick_variant(): 256 x 256 ring kernel ...
        rounds = (t + 255) // 256
        if t >= 140 and (t <= 256 or 100 * t >= 75 * rounds * 256):
            return True
        if ((M + 255) // 256) * ((N + 127) // 128) >= 128:        # ... else the LDS-DMA kernel (256 / 192 / 128-row tiles) ...
            return True
        return ((M + 127) // 128) * ((N + 127) // 128) >= 128     # ... also where only its 128-row tiles are numerous enough (round 4)

    def summary(self):
        torch.cuda.synchronize()
        fl = sum(r[0] for r in self.records)
        ms = sum(r[1].elapsed_time(r[2]) for r in self.records)
        n = len(self.records)
        by = sum(r[3] for r in self.records)
        return {"launches": n, "flops": fl, "ms": ms, "avg_launch_ms": ms / max(n, 1), "tflops": fl / max(ms, 1e-9) / 1e9,
                "flops_per_launch": fl / max(n, 1), "bytes_per_launch": by / max(n, 1)}


def gemm_profile_by_shape(prof: "GemmProfile"):
    """(tools) per (M, N, K, K2, residual): launches, total ms, TFLOP/s — which shapes carry the GEMM time of a step"""
    torch.cuda.synchronize()
    agg = {}
    for r in prof.records:
        a = agg.setdefault(r[4], [0, 0.0, 0.0])
        a[0] += 1
        a[1] += r[1].elapsed_time(r[2])
        a[2] += r[0]
    return sorted(((k, n, ms, fl / max(ms, 1e-9) / 1e9) for k, (n, ms, fl) in agg.items()), key=lambda t: -t[2])


GEMM_PROFILE: Optional[GemmProfile] = None


def gemm_nt_splitk(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, alpha: float = 1.0, split_k: int = 0) -> torch.Tensor:
    """out[M,N] (f32) += alpha * a[M,K] @ b[N,K]^T, K sliced over workgroups (atomics)."""
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and out.shape == (M, N) and out.dtype == torch.float32
    if split_k <= 0:
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        split_k = max(1, min(K // 256 if K >= 256 else 1, (512 + tiles - 1) // tiles))
    get_lib().call("bra_gemm_bf16_nt_splitk", a, _ld(a), b, _ld(b), out, _ld(out), M, N, K, alpha, split_k, current_stream(a))
    return out


def wgrad_tn(y: torch.Tensor, t: torch.Tensor, out: torch.Tensor, transposed_out: bool = False, alpha: float = 1.0,
             m_chunk: int = 0, drop=None) -> torch.Tensor:
    """out += alpha * y^T t from row-major y [M, N], t [M, R]: out [N, R], or [R, N] with transposed_out.
    drop = (p, seeds): y is used as dropout_rb(y), one mask stream per 32 columns of t."""
    M, N = y.shape
    R = t.shape[1]
    assert t.shape[0] == M and out.dtype == torch.float32 and out.shape == ((R, N) if transposed_out else (N, R))
    c_sn, c_sr = (1, _ld(out)) if transposed_out else (_ld(out), 1)
    if drop is not None:
        get_lib().call("bra_wgrad_tn_drop", y, _ld(y), t, _ld(t), out, c_sn, c_sr, M, N, R, alpha, m_chunk, drop[0],
                       *_seeds4(drop[1]), min(len(drop[1]), 4), current_stream(y))
        return out
    get_lib().call("bra_wgrad_tn", y, _ld(y), t, _ld(t), out, c_sn, c_sr, M, N, R, alpha, m_chunk, current_stream(y))
    return out


def _seeds4(seeds):
    s = [int(x) & 0xFFFFFFFF for x in seeds] + [0, 0, 0, 0]
    return s[:4]


def dropout_mask(M: int, K: int, p: float, seed: int, device) -> torch.Tensor:
    """keep mask (uint8 [M, K]) of one dropout stream — the mask the LoRA kernels regenerate on the fly"""
    out = torch.empty((M, K), dtype=torch.uint8, device=device)
    get_lib().call("bra_dropout_mask", out, M, K, p, int(seed) & 0xFFFFFFFF, current_stream(out))
    return out


LORA_DOWN_SPLITK = os.environ.get("BRA_LORA_SPLITK", "1") != "0"


def lora_down_drop(x: torch.Tensor, A: torch.Tensor, alpha: float, p: float, seeds) -> torch.Tensor:
    """t [M, R] = alpha * dropout_j(x) A^T with one mask stream per 32 rows of A (PEFT: per target module)"""
    M, K = x.shape
    R = A.shape[0]
    t = torch.empty((M, R), dtype=BF16, device=x.device)
    lib = get_lib()
    ks = int(lib._dll.bra_lora_down_splitk_plan(int(M), int(K))) if LORA_DOWN_SPLITK else 1
    if ks > 1:
        part = torch.empty((ks, M, R), dtype=torch.float32, device=x.device)
        lib.call("bra_lora_down_drop_splitk", x, _ld(x), A, _ld(A), t, _ld(t), M, K, R, alpha, p, *_seeds4(seeds), min(len(seeds), 4),
                 part, ks, current_stream(x))
        return t
    lib.call("bra_lora_down_drop", x, _ld(x), A, _ld(A), t, _ld(t), M, K, R, alpha, p, *_seeds4(seeds), min(len(seeds), 4),
             current_stream(x))
    return t


def lora_up_drop(dts: torch.Tensor, AT: torch.Tensor, p: float, seeds) -> torch.Tensor:
    """[M, K] = sum_j dropout_j'(dts[:, block j] A[block j, :]): input gradient of the LoRA branch (AT = A^T [K, R])"""
    M, R = dts.shape
    K = AT.shape[0]
    out = torch.empty((M, K), dtype=BF16, device=dts.device)
    get_lib().call("bra_lora_up_drop", dts, _ld(dts), AT, _ld(AT), out, _ld(out), M, K, R, p, *_seeds4(seeds), min(len(seeds), 4),
                   current_stream(dts))
    return out


def lmhead_logprob(h: torch.Tensor, emb: torch.Tensor, tgt: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (logp[M], lse[M]) of target tokens under softmax(h @ emb^T) without materialising logits."""
    M, K = h.shape
    V = emb.shape[0]
    nchunk = (V + 63) // 64
    dev = h.device
    pm = torch.empty((M, nchunk), dtype=torch.float32, device=dev)
    ps = torch.empty((M, nchunk), dtype=torch.float32, device=dev)
    tl = torch.zeros((M,), dtype=torch.float32, device=dev)
    lse = torch.empty((M,), dtype=torch.float32, device=dev)
    logp = torch.empty((M,), dtype=torch.float32, device=dev)
    st = current_stream(h)
    lib = get_lib()
    lib.call("bra_lmhead_lse_partials", h, _ld(h), emb, _ld(emb), M, V, K, tgt, pm, ps, tl, st)
    lib.call("bra_lse_merge", pm, ps, tl, lse, logp, M, nchunk, st)
    return logp, lse


def lmhead_dlogits(h: torch.Tensor, emb: torch.Tensor, tgt: torch.Tensor, lse: torch.Tensor, coef: torch.Tensor) -> torch.Tensor:
    M, K = h.shape
    V = emb.shape[0]
    out = torch.empty((M, V), dtype=BF16, device=h.device)
    get_lib().call("bra_lmhead_dlogits", h, _ld(h), emb, _ld(emb), M, V, K, tgt, lse, coef, out, _ld(out), current_stream(h))
    return out


# --------------------------------------------------------------------------- norms / activations
def rmsnorm_fwd(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    x2 = _as2d(x)
    y = torch.empty_like(x2)
    get_lib().call("bra_rmsnorm_fwd", x2, _ld(x2), w, y, _ld(y), None, x2.shape[0], x2.shape[1], eps, current_stream(x))
    return y.view(x.shape)


def rmsnorm_bwd(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, eps: float, dres: Optional[torch.Tensor] = None) -> torch.Tensor:
    x2, dy2 = _as2d(x), _as2d(dy)
    dx = torch.empty_like(x2)
    d2 = _as2d(dres) if dres is not None else None
    get_lib().call("bra_rmsnorm_bwd", dy2, _ld(dy2), x2, _ld(x2), w, d2, _ld(d2) if d2 is not None else 0, dx, _ld(dx),
                   x2.shape[0], x2.shape[1], eps, current_stream(x))
    return dx.view(x.shape)


def layernorm_fwd(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    x2 = _as2d(x)
    y = torch.empty_like(x2)
    get_lib().call("bra_layernorm_fwd", x2, _ld(x2), w, b, y, _ld(y), x2.shape[0], x2.shape[1], eps, current_stream(x))
    return y.view(x.shape)


def gemm_swiglu(x: torch.Tensor, W: torch.Tensor, a2: Optional[torch.Tensor] = None, b2: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """act [M, F] = swiglu_fwd(gemm_nt(x, W [2 F, K], a2=a2, b2=b2)) in ONE launch (bra_gemm_swiglu_bf16_nt; same values), or None where
    the fused kernel does not apply (the caller runs the two launches)"""
    M, K = x.shape
    F = W.shape[0] // 2
    K2 = a2.shape[1] if a2 is not None else 0
    if M <= 16 or K % 64 or K2 % 64 or F % 128 or W.shape[0] != 2 * F:
        return None
    act = torch.empty((M, F), dtype=BF16, device=x.device)
    rc = get_lib().call_rc("bra_gemm_swiglu_bf16_nt", x, _ld(x), W, _ld(W), a2, _ld(a2) if a2 is not None else 0, b2,
                           _ld(b2) if b2 is not None else 0, K2, act, _ld(act), M, F, K, 1.0, current_stream(x))
    return None if rc != 0 else act        # (call_rc raises on every status but BRA_ERR_UNSUPPORTED)


def swiglu_fwd(gu: torch.Tensor) -> torch.Tensor:
    gu2 = _as2d(gu)
    F = gu2.shape[1] // 2
    act = torch.empty((gu2.shape[0], F), dtype=BF16, device=gu.device)
    get_lib().call("bra_swiglu_fwd", gu2, _ld(gu2), act, _ld(act), gu2.shape[0], F, current_stream(gu))
    return act


def swiglu_bwd(gu: torch.Tensor, dact: torch.Tensor) -> torch.Tensor:
    gu2, d2 = _as2d(gu), _as2d(dact)
    F = gu2.shape[1] // 2
    dgu = torch.empty_like(gu2)
    get_lib().call("bra_swiglu_bwd", gu2, _ld(gu2), d2, _ld(d2), dgu, _ld(dgu), gu2.shape[0], F, current_stream(gu))
    return dgu


def _bsh_strides(t: torch.Tensor) -> Tuple[int, int, int]:
    """t is [B, S, H, hd] (any permutation underneath) -> element strides (b, s, h)"""
    assert t.dim() == 4 and t.stride(3) == 1
    return t.stride(0), t.stride(1), t.stride(2)


def qk_norm_rope_fwd(qkv, qw, kw, cosT, sinT, pos, S, Hq, Hkv, hd, eps, qscale, q, k, v, s_off=0):
    """qkv [T, (Hq+2Hkv)*hd]; q/k/v: 4-D views indexed [B, S, H, hd] (arbitrary strides, hd contiguous)."""
    T = qkv.shape[0]
    get_lib().call("bra_qk_norm_rope_fwd", qkv, _ld(qkv), qw, kw, cosT, sinT, pos, T, S, Hq, Hkv, hd, eps, qscale,
                   q, *_bsh_strides(q), k, *_bsh_strides(k), v, *_bsh_strides(v), s_off, current_stream(qkv))


def qk_norm_rope_bwd(qkv, qw, kw, cosT, sinT, pos, S, Hq, Hkv, hd, eps, qscale, dq, dk, dv):
    T = qkv.shape[0]
    dqkv = torch.empty_like(qkv)
    get_lib().call("bra_qk_norm_rope_bwd", qkv, _ld(qkv), qw, kw, cosT, sinT, pos, T, S, Hq, Hkv, hd, eps, qscale,
                   dq, *_bsh_strides(dq), dk, *_bsh_strides(dk), dv, *_bsh_strides(dv), dqkv, _ld(dqkv), current_stream(qkv))
    return dqkv


# --------------------------------------------------------------------------- attention
def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def head_transpose(x: torch.Tensor) -> torch.Tensor:
    """x: [B, S, H, hd] view -> [B, H, hd, pad64(S)] (zero padded)"""
    B, S, H, hd = x.shape
    out = torch.empty((B, H, hd, pad64(S)), dtype=BF16, device=x.device)
    sb, ss, sh = _bsh_strides(x)
    get_lib().call("bra_head_transpose", x, sb, ss, sh, out, out.stride(0), out.stride(1), out.stride(2), B, S, H, hd, current_stream(x))
    return out


def attn_fwd(q, k, vt, kmask, causal: bool, scale: float, q_off: Optional[int] = None, need_lse: bool = True,
             out: Optional[torch.Tensor] = None, nsplit: Optional[int] = None):
    """q: [B,Sq,Hq,hd] view, k: [B,Sk,Hkv,hd] view, vt: [B,Hkv,hd,pitch]; -> (o [B,Sq,Hq,hd] contiguous, lse [B,Hq,Sq])"""
    B, Sq, Hq, hd = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    if q_off is None:
        q_off = Sk - Sq
    o = out if out is not None else torch.empty((B, Sq, Hq, hd), dtype=BF16, device=q.device)
    lse = torch.empty((B, Hq, Sq), dtype=torch.float32, device=q.device) if need_lse else None
    ns = attn_fwd_split_parts(B, Hq, Sq, Sk, hd, causal) if nsplit is None else int(nsplit)
    if ns > 1:
        # grids that cannot fill the chip (one prompt: 144 workgroups; a 256-query completion segment: 128): the key range of every
        # query block in `ns` parts + one merge launch (bra_attn_fwd_split)
        part_o = torch.empty((B, Hq, ns, Sq, hd), dtype=torch.float32, device=q.device)
        part_ml = torch.empty((B, Hq, ns, Sq, 2), dtype=torch.float32, device=q.device)
        get_lib().call("bra_attn_fwd_split", q, *_bsh_strides(q), k, *_bsh_strides(k), vt, vt.stride(0), vt.stride(1), vt.stride(2),
                       o, *_bsh_strides(o), lse, kmask, B, Hq, Hkv, Sq, Sk, hd, int(causal), q_off, scale, ns, part_o, part_ml,
                       current_stream(q))
        return o, lse
    get_lib().call("bra_attn_fwd", q, *_bsh_strides(q), k, *_bsh_strides(k), vt, vt.stride(0), vt.stride(1), vt.stride(2),
                   o, *_bsh_strides(o), lse, kmask, B, Hq, Hkv, Sq, Sk, hd, int(causal), q_off, scale, current_stream(q))
    return o, lse


ATTN_SPLIT = os.environ.get("BRA_ATTN_SPLIT", "1") == "1"


def attn_fwd_split_parts(B: int, Hq: int, Sq: int, Sk: int, hd: int, causal: bool) -> int:
    """key parts per query block of the forward: 1 unless the grid of 256-query workgroups is far from filling 256 CUs"""
    if not ATTN_SPLIT or hd < 64 or Sq <= 128 or Sk < 1024:
        return 1
    wgs = ((Sq + 255) // 256) * Hq * B
    if wgs >= 200:
        return 1
    return max(1, min(4, (300 + wgs // 2) // wgs))


def attn_bwd_split_parts(B: int, Hq: int, Hkv: int, Sq: int, Sk: int, hd: int, causal: bool):
    """-> (key parts of the dQ kernel, loop parts of the one-launch dK + dV kernel); (1, 1) for grids that fill the chip"""
    ns_dq = attn_fwd_split_parts(B, Hq, Sq, Sk, hd, causal)              # same grid as the forward: 256-query workgroups
    ns_kv = 1
    if ATTN_SPLIT and hd >= 64 and Sq >= 1024 and Sk > 128:
        wgs = ((Sk + 255) // 256) * Hkv * B                              # 256-key workgroups of the pipelined kernels (k_attn4b.hip)
        if wgs < 200:
            ns_kv = max(1, min(4, (300 + wgs // 2) // wgs))
    return ns_dq, ns_kv


def attn_bwd(q, k, v, o, dout, lse, kmask, causal: bool, scale: float, q_off: Optional[int] = None, nsplit=None):
    """all of q,k,v,o,dout are [B,S,H,hd] views; returns dq, dk, dv (contiguous [B,S,H,hd]).  `nsplit` = (dQ parts, dK/dV parts) or
    None: chosen by the grid size (attn_bwd_split_parts)"""
    B, Sq, Hq, hd = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    if q_off is None:
        q_off = Sk - Sq
    lib = get_lib()
    st = current_stream(q)
    delta = torch.empty((B, Hq, Sq), dtype=torch.float32, device=q.device)
    lib.call("bra_attn_delta", dout, *_bsh_strides(dout), o, *_bsh_strides(o), delta, B, Sq, Hq, hd, st)
    kt = head_transpose(k)
    qt = head_transpose(q)
    dot = head_transpose(dout)
    dq = torch.empty((B, Sq, Hq, hd), dtype=BF16, device=q.device)
    dk = torch.empty((B, Sk, Hkv, hd), dtype=BF16, device=q.device)
    dv = torch.empty((B, Sk, Hkv, hd), dtype=BF16, device=q.device)
    ns_dq, ns_kv = attn_bwd_split_parts(B, Hq, Hkv, Sq, Sk, hd, causal) if nsplit is None else nsplit
    if ns_dq > 1 or ns_kv > 1:
        part_dq = torch.empty((B, Hq, ns_dq, Sq, hd), dtype=torch.float32, device=q.device) if ns_dq > 1 else None
        part_dk = torch.empty((B, Hkv, ns_kv, Sk, hd), dtype=torch.float32, device=q.device) if ns_kv > 1 else None
        part_dv = torch.empty((B, Hkv, ns_kv, Sk, hd), dtype=torch.float32, device=q.device) if ns_kv > 1 else None
        lib.call("bra_attn_bwd_split", q, *_bsh_strides(q), k, *_bsh_strides(k), v, *_bsh_strides(v), dout, *_bsh_strides(dout),
                 kt, kt.stride(0), kt.stride(1), kt.stride(2), qt, qt.stride(0), qt.stride(1), qt.stride(2),
                 dot, dot.stride(0), dot.stride(1), dot.stride(2), lse, delta, kmask,
                 dq, *_bsh_strides(dq), dk, *_bsh_strides(dk), dv, *_bsh_strides(dv),
                 B, Hq, Hkv, Sq, Sk, hd, int(causal), q_off, scale, ns_dq, part_dq, ns_kv, part_dk, part_dv, st)
        return dq, dk, dv
    lib.call("bra_attn_bwd", q, *_bsh_strides(q), k, *_bsh_strides(k), v, *_bsh_strides(v), dout, *_bsh_strides(dout),
             kt, kt.stride(0), kt.stride(1), kt.stride(2), qt, qt.stride(0), qt.stride(1), qt.stride(2),
             dot, dot.stride(0), dot.stride(1), dot.stride(2), lse, delta, kmask,
             dq, *_bsh_strides(dq), dk, *_bsh_strides(dk), dv, *_bsh_strides(dv),
             B, Hq, Hkv, Sq, Sk, hd, int(causal), q_off, scale, st)
    return dq, dk, dv


def attn_decode(q, kc, vc, kmask, length: int, scale: float, ws=None):
    """q [B,Hq,hd]; kc/vc [B,Hkv,Smax,hd] contiguous; -> o [B, Hq*hd]"""
    B, Hq, hd = q.shape
    Hkv, Smax = kc.shape[1], kc.shape[2]
    nchunk = (length + 127) // 128
    if ws is None:
        po = torch.empty((B, Hq, nchunk, hd), dtype=torch.float32, device=q.device)
        pml = torch.empty((B, Hq, nchunk, 2), dtype=torch.float32, device=q.device)
        o = torch.empty((B, Hq * hd), dtype=BF16, device=q.device)
    else:
        po, pml, o = ws
    get_lib().call("bra_attn_decode", q, kc, vc, kmask, po, pml, o, B, Hq, Hkv, hd, Smax, length, scale, current_stream(q))
    return o


# --------------------------------------------------------------------------- data movement
def dna_scatter_plan(ids32, dna_id, dna_mask_u8, seq_order, tok_src, counts):
    nseq, Sd = (dna_mask_u8.shape if dna_mask_u8 is not None else (0, 0))
    get_lib().call("bra_dna_scatter_plan", ids32, ids32.numel(), dna_id, dna_mask_u8, nseq, Sd, seq_order, tok_src, counts,
                   current_stream(ids32))


def embed_scatter_fwd(ids32, tok_src, emb, dna_rows, out):
    H = emb.shape[1]
    get_lib().call("bra_embed_scatter_fwd", ids32, tok_src, emb, _ld(emb), dna_rows, _ld(dna_rows) if dna_rows is not None else 0,
                   out, _ld(out), ids32.numel(), H, current_stream(emb))
    return out


def embed_scatter_bwd(tok_src, dout, ddna):
    H = dout.shape[-1]
    d2 = _as2d(dout)
    get_lib().call("bra_embed_scatter_bwd", tok_src, d2, _ld(d2), ddna, _ld(ddna), d2.shape[0], H, current_stream(dout))
    return ddna


def gather_rows(rows32, x):
    out = torch.empty((rows32.numel(), x.shape[1]), dtype=BF16, device=x.device)
    get_lib().call("bra_gather_rows", rows32, x, _ld(x), out, _ld(out), rows32.numel(), x.shape[1], current_stream(x))
    return out


def scatter_rows(rows32, x, nrows_out):
    out = torch.zeros((nrows_out, x.shape[1]), dtype=BF16, device=x.device)
    get_lib().call("bra_scatter_rows", rows32, x, _ld(x), out, _ld(out), rows32.numel(), x.shape[1], current_stream(x))
    return out


def group_sum(src: torch.Tensor, copies: int, add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """src [R * copies, ...] (bf16; everything behind dim 0 contiguous, dim 0 may be strided) -> [R, ...]: sum over the members of
    each group (+ add [R, ...]), fp32 accumulation.  The upstream gradient of rows shared by the copies of a GRPO group."""
    n = 1
    for d in src.shape[1:]:
        n *= d
    assert src.dtype == BF16 and src[0].is_contiguous() and src.shape[0] % copies == 0
    R = src.shape[0] // copies
    out = torch.empty((R,) + tuple(src.shape[1:]), dtype=BF16, device=src.device)
    if add is not None:
        assert add.shape == out.shape and add.dtype == BF16 and add[0].is_contiguous()
    get_lib().call("bra_group_sum", src, src.stride(0), copies, add, add.stride(0) if add is not None else 0, out, out.stride(0),
                   R, n, current_stream(src))
    return out


def group_broadcast(src: torch.Tensor, out: torch.Tensor, copies: int) -> torch.Tensor:
    """src [R, I, n...] (contiguous) -> out[(r copies + c), i, :n...] for every copy c; out is [R * copies, I, N...] with the same
    trailing dims except dim 2 (N >= n rows; the first n are written): a shared prompt's K / V in front of each rollout's own."""
    R, I = src.shape[0], src.shape[1]
    assert src.dtype == BF16 and out.dtype == BF16 and src.is_contiguous() and out.is_contiguous()
    assert out.shape[0] == R * copies and out.shape[1] == I and out.shape[3:] == src.shape[3:] and out.shape[2] >= src.shape[2]
    n = src[0, 0].numel()
    get_lib().call("bra_group_broadcast", src, src.stride(1), out, out.stride(1), R, copies, I, n, current_stream(src))
    return out


def transpose2d(x: torch.Tensor, pad_to: int = 8) -> torch.Tensor:
    """x [R, C] -> [C, Rp] view [:, :R] with row pitch Rp = R rounded up to `pad_to` (zero padded)"""
    R, C = x.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    buf = torch.zeros((C, Rp), dtype=BF16, device=x.device) if Rp != R else torch.empty((C, Rp), dtype=BF16, device=x.device)
    get_lib().call("bra_transpose2d", x, _ld(x), buf, _ld(buf), R, C, current_stream(x))
    return buf


def colsum(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    get_lib().call("bra_colsum", x, _ld(x), out, x.shape[0], x.shape[1], current_stream(x))
    return out


# --------------------------------------------------------------------------- GRPO / optimiser
def sample(logits, temperature, top_k, top_p, do_sample, seed, step_t, finished, pad_id, out_ids, out_logp=None,
           eos_id=-1, tokens_out=None, ws=None, embed=None, eos_id2=-1):
    """`embed` = (E [V, H], x [B, H], ss [8, nss] or None): the drawing wave also writes x[b] = E[token] and the RMSNorm
    statistic of that row (two-stage path only: V >= 4096)."""
    B, V = logits.shape
    if do_sample and not (1 <= top_k <= 64):
        # HF's top_k = 0 ("disabled") or more than 64 survivors: the general kernel (exact thresholds by bisection; slow — see
        # bra_sample_full).  The fused embedding gather of the fast paths is not part of it: the caller's step embeds the token.
        if embed is not None:
            raise NotImplementedError("the fused embedding gather needs 1 <= top_k <= 64 (callers fall back to the unfused step)")
        return sample_full(logits, temperature, top_k, top_p, seed, step_t, finished, pad_id, out_ids, out_logp, eos_id=eos_id,
                           eos_id2=eos_id2, tokens_out=tokens_out)
    if ws is None and V >= 4096:
        k = min(top_k, 64) if top_k > 0 else 64
        ws = torch.empty((2 * B * 64 * k,), dtype=torch.float32, device=logits.device)
    ldt = tokens_out.stride(0) if tokens_out is not None else 0
    if embed is not None:
        E, x, ss = embed
        get_lib().call("bra_sample_embed", logits, _ld(logits), B, V, temperature, top_k, top_p, int(do_sample),
                       seed & 0xFFFFFFFF, step_t, finished, pad_id, eos_id, eos_id2, out_ids, out_logp, tokens_out, ldt, ws, E, _ld(E),
                       E.shape[1], x, _ld(x), ss, ss.shape[-1] if ss is not None else 0, current_stream(logits))
        return out_ids
    get_lib().call("bra_sample", logits, _ld(logits), B, V, temperature, top_k, top_p, int(do_sample), seed & 0xFFFFFFFF,
                   step_t, finished, pad_id, eos_id, eos_id2, out_ids, out_logp, tokens_out, ldt, ws, current_stream(logits))
    return out_ids


def sample_full(logits, temperature, top_k, top_p, seed, step, finished, pad_id, out_ids, out_logp=None, eos_id=-1, eos_id2=-1,
                tokens_out=None):
    """temperature -> top-k -> top-p -> multinomial for ANY top_k >= 0 (0: HF's 'disabled'); `step`: device int32 [1], python int or None"""
    B, V = logits.shape
    ldt = tokens_out.stride(0) if tokens_out is not None else 0
    step_ptr, step_i = (step, 0) if isinstance(step, torch.Tensor) else (None, int(step or 0))
    get_lib().call("bra_sample_full", logits, _ld(logits), B, V, temperature, max(int(top_k), 0), top_p, seed & 0xFFFFFFFF, step_ptr, step_i,
                   finished, pad_id, eos_id, eos_id2, out_ids, out_logp, tokens_out, ldt, current_stream(logits))
    return out_ids


def tile_max(logits: torch.Tensor, tmax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """maxima of the 16-column tiles of fp32 logits [B, V] -> [B, ceil(V / 16)] (the second input of `sample_tiles`)"""
    B, V = logits.shape
    if tmax is None:
        tmax = torch.empty((B, (V + 15) // 16), dtype=torch.float32, device=logits.device)
    get_lib().call("bra_tile_max", logits, _ld(logits), B, V, tmax, _ld(tmax), current_stream(logits))
    return tmax


def sample_tiles(logits, tmax, temperature, top_k, top_p, do_sample, seed, step, finished, pad_id, out_ids, out_logp=None, eos_id=-1,
                 tokens_out=None, ws=None, embed=None, eos_id2=-1, advance=None):
    """`sample` over (logits, tile maxima): same tokens, two light launches.  `step`: a device int32 [1] (replayed loops) or a
    python int (the loop issued launch by launch).  `advance` = (pos0 int32 [B], pos_out int32 [B] or None, cosT, sinT, hd,
    rope_rows fp32 [B, hd]): the drawing wave also leaves pos0 + step and the (cos | sin) row of that position."""
    B, V = logits.shape
    if do_sample and not (1 <= top_k <= 64):
        raise NotImplementedError("sampling needs 1 <= top_k <= 64 on the HIP path (see ops.sample)")
    k = top_k if do_sample else 1
    if ws is None:
        ws = torch.empty((2 * B * 8 * k,), dtype=torch.float32, device=logits.device)
    ldt = tokens_out.stride(0) if tokens_out is not None else 0
    E, x, ss = embed if embed is not None else (None, None, None)
    pos0, pos_out, cosT, sinT, hd, rows = advance if advance is not None else (None, None, None, None, 0, None)
    step_ptr, step_i = (step, 0) if isinstance(step, torch.Tensor) else (None, int(step))
    get_lib().call("bra_sample_tiles", logits, _ld(logits), tmax, _ld(tmax), B, V, temperature, top_k, top_p, int(do_sample),
                   seed & 0xFFFFFFFF, step_ptr, step_i, finished, pad_id, eos_id, eos_id2, out_ids, out_logp, tokens_out, ldt, ws,
                   E, _ld(E) if E is not None else 0, E.shape[1] if E is not None else 0, x, _ld(x) if x is not None else 0, ss,
                   ss.shape[-1] if ss is not None else 0, pos0, pos_out, cosT, sinT, hd, rows, current_stream(logits))
    return out_ids


def force_token_tiles(logits: torch.Tensor, token: int, step, at: torch.Tensor, tmax: Optional[torch.Tensor]):
    """`force_token` that keeps the tile maxima consistent; `step`: device int32 [1] or python int"""
    B, V = logits.shape
    step_ptr, step_i = (step, 0) if isinstance(step, torch.Tensor) else (None, int(step))
    get_lib().call("bra_force_token_tiles", logits, _ld(logits), B, V, int(token), step_ptr, step_i, at, tmax,
                   _ld(tmax) if tmax is not None else 0, current_stream(logits))


def force_token(logits: torch.Tensor, token: int, step_t: torch.Tensor, at: torch.Tensor):
    """logits[b, token] = +big where at[b] == step_t[0] (synthetic EOS schedule of the straggler bench / tests)"""
    B, V = logits.shape
    get_lib().call("bra_force_token", logits, _ld(logits), B, V, int(token), step_t, at, current_stream(logits))


def advance_counters(pos: torch.Tensor, a: Optional[torch.Tensor] = None, b: Optional[torch.Tensor] = None, rope=None):
    """rope = (cosT, sinT, hd, rows [n, hd] fp32): also refresh the (cos | sin) rows of the advanced positions"""
    cosT, sinT, hd, rows = rope if rope is not None else (None, None, 0, None)
    get_lib().call("bra_advance_counters", pos, pos.numel(), a, b, cosT, sinT, hd, rows, current_stream(pos))


def rope_rows(cosT: torch.Tensor, sinT: torch.Tensor, pos: torch.Tensor, hd: int, rows: torch.Tensor):
    get_lib().call("bra_rope_rows", cosT, sinT, pos, pos.numel(), hd, rows, current_stream(pos))


def row_sumsq(x: torch.Tensor, nss: int = 32) -> torch.Tensor:
    """partial sums of squares of the rows of x [M<=16, K] in the layout bra_dec_gemm2 consumes: fp32 [8, nss] ([16, nss] for M > 8)"""
    ss = torch.zeros((16 if x.shape[0] > 8 else 8, nss), dtype=torch.float32, device=x.device)
    get_lib().call("bra_row_sumsq", x, _ld(x), x.shape[0], x.shape[1], ss, nss, current_stream(x))
    return ss


def dec_gemm2(x, W, ss_in=None, norm_w=None, eps=1e-6, res=None, act=False, out_f32=False, want_ss=False, packed=False,
              tile_max=None):
    """decode-time projection (bra_dec_gemm2): y = rmsnorm(x) W^T (+res | SwiGLU | fp32); returns (y, ss_out or None).
    packed: W is the fragment-ordered copy made by dec_pack_weights (same logical shape); 3 = with the norm weight folded.
    tile_max (out_f32 only): fp32 [M, >= ceil(N / 16)] that receives the maximum of every 16-column tile of the logits"""
    M, K = x.shape
    N = W.shape[0]
    out = torch.empty((M, N // 2 if act else N), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    nss_out = (N // 8 + 32) // 32 * 32
    ss_out = torch.zeros((16 if M > 8 else 8, nss_out), dtype=torch.float32, device=x.device) if want_ss else None
    if tile_max is not None:
        assert out_f32 and not want_ss
        ss_out, nss_out, want_ss = tile_max, _ld(tile_max), True
    get_lib().call("bra_dec_gemm2_packed", x, _ld(x), ss_in, ss_in.shape[1] if ss_in is not None else 0, norm_w, eps, W, _ld(W),
                   res, _ld(res) if res is not None else 0, out, _ld(out), ss_out, nss_out if want_ss else 0, M, N, K,
                   int(act), int(out_f32), int(packed), current_stream(x))
    return out, (None if tile_max is not None else ss_out)


def dec_pack_weights(W: torch.Tensor, act: bool = False, out_f32: bool = False, norm_w: Optional[torch.Tensor] = None,
                     rows: int = 8) -> Optional[torch.Tensor]:
    """W [N, K] -> fragment-ordered copy for bra_dec_gemm2(packed=1); None when the shape is not a multiple of the tile.
    norm_w [K]: the input's RMSNorm weight is folded in (stream the copy with packed=3).  rows: batch rows the copy is streamed
    against (above 8 every projection uses the 16-column tile order)"""
    N, K = W.shape
    out = torch.empty((N, K), dtype=BF16, device=W.device)
    rc = get_lib().call_rc("bra_dec_pack_weights_rows", W, _ld(W), N, K, int(act), int(out_f32), norm_w, int(rows), out, current_stream(W))
    if rc == -2:
        return None
    if rc != 0:
        raise RuntimeError(f"bra_dec_pack_weights failed with status {rc}")
    return out


def dec_pack_weights_fp8(W: torch.Tensor, act: bool = False, out_f32: bool = False, norm_w: Optional[torch.Tensor] = None):
    """W [N, K] bf16 -> (q uint8 [N, K] in bra_dec_gemm2_fp8's fragment order, scale fp32 [N]) — e4m3 with one scale per output row
    (bra_dec_pack_weights_fp8); None when the shape is not a whole number of tiles / k-step pairs"""
    N, K = W.shape
    q = torch.empty((N, K), dtype=torch.uint8, device=W.device)
    scale = torch.empty((N,), dtype=torch.float32, device=W.device)
    rc = get_lib().call_rc("bra_dec_pack_weights_fp8", W, _ld(W), N, K, int(act), int(out_f32), norm_w, q, scale, current_stream(W))
    if rc == -2:
        return None
    return q, scale


def quant_rows_fp8(x: torch.Tensor, colw: Optional[torch.Tensor] = None, rms_eps: Optional[float] = None):
    """x [M, K] bf16 -> (q uint8 [M, K] e4m3, scale fp32 [M]) (bra_quant_rows_fp8).  colw: bf16 [K] multiplied in first (a weight with
    its input's RMSNorm weight folded); rms_eps given: scale = rstd * absmax / 448 (rows feeding a folded-norm projection)."""
    M, K = x.shape
    q = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    scale = torch.empty((M,), dtype=torch.float32, device=x.device)
    get_lib().call("bra_quant_rows_fp8", x, _ld(x), M, K, colw, q, K, scale, int(rms_eps is not None),
                   float(rms_eps if rms_eps is not None else 0.0), current_stream(x))
    return q, scale


def swiglu_quant_fp8(gu: torch.Tensor):
    """[gate | up] rows [M, 2 F] bf16 -> (q uint8 [M, F], scale [M]) of act = bf16(bf16(silu(gate)) up) (bra_swiglu_quant_fp8)"""
    M, F2 = gu.shape
    F = F2 // 2
    q = torch.empty((M, F), dtype=torch.uint8, device=gu.device)
    scale = torch.empty((M,), dtype=torch.float32, device=gu.device)
    get_lib().call("bra_swiglu_quant_fp8", gu, _ld(gu), M, F, q, F, scale, current_stream(gu))
    return q, scale


def gemm_fp8_nt(a8: torch.Tensor, sa: torch.Tensor, b8: torch.Tensor, sb: torch.Tensor, res: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None, out_f32: bool = False) -> torch.Tensor:
    """C [M, N] = sa[m] sb[n] (a8 b8^T) (+ res) on the fp8 MFMA path (bra_gemm_fp8_nt); a8 [M, K], b8 [N, K] uint8 (e4m3), K % 128 == 0"""
    M, K = a8.shape
    N = b8.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32 if out_f32 else BF16, device=a8.device)
    get_lib().call("bra_gemm_fp8_nt", a8, a8.stride(0), sa, b8, b8.stride(0), sb, out, _ld(out), M, N, K, res,
                   _ld(res) if res is not None else 0, int(out_f32), current_stream(a8))
    return out


def dec_gemm2_fp8(x, Wq, scale, ss_in=None, eps=1e-6, res=None, act=False, out_f32=False, want_ss=False, tile_max=None):
    """decode-time projection over fp8 weights (bra_dec_gemm2_fp8): y = [rstd] scale[n] (x q^T) (+res | SwiGLU | fp32);
    `ss_in` given = the input's RMSNorm weight was folded into the weights; returns (y, ss_out or None); None when unsupported"""
    M, K = x.shape
    N = Wq.shape[0]
    out = torch.empty((M, N // 2 if act else N), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    nss_out = (N // 8 + 32) // 32 * 32
    ss_out = torch.zeros((8, nss_out), dtype=torch.float32, device=x.device) if want_ss else None
    if tile_max is not None:
        assert out_f32 and not want_ss
        ss_out, nss_out, want_ss = tile_max, _ld(tile_max), True
    rc = get_lib().call_rc("bra_dec_gemm2_fp8", x, _ld(x), ss_in, ss_in.shape[1] if ss_in is not None else 0, eps, Wq, scale,
                           res, _ld(res) if res is not None else 0, out, _ld(out), ss_out, nss_out if want_ss else 0, M, N, K,
                           int(act), int(out_f32), int(ss_in is not None), current_stream(x))
    if rc == -2:
        return None
    return out, (None if tile_max is not None else ss_out)


def eos_mask(ids32: torch.Tensor, eos_id: int):
    B, C = ids32.shape
    mask = torch.empty((B, C), dtype=torch.int32, device=ids32.device)
    lengths = torch.empty((B,), dtype=torch.int32, device=ids32.device)
    get_lib().call("bra_eos_mask", ids32, B, C, eos_id, mask, lengths, current_stream(ids32))
    return mask, lengths


def group_advantage(rewards: torch.Tensor, G: int):
    N, F = rewards.shape
    adv = torch.empty((N,), dtype=torch.float32, device=rewards.device)
    gm = torch.empty((N // G,), dtype=torch.float32, device=rewards.device)
    gs = torch.empty((N // G,), dtype=torch.float32, device=rewards.device)
    get_lib().call("bra_group_advantage", rewards, N, F, G, adv, gm, gs, current_stream(rewards))
    return adv, gm, gs


def grpo_loss(logp, old_logp, ref_logp, adv, mask, eps_lo, eps_hi, beta, need_grad=True):
    B, C = logp.shape
    out3 = torch.empty((3,), dtype=torch.float32, device=logp.device)
    dlogp = torch.empty_like(logp) if need_grad else None
    get_lib().call("bra_grpo_loss", logp, old_logp, ref_logp, adv, mask, B, C, eps_lo, eps_hi, beta, out3, dlogp, current_stream(logp))
    return out3, dlogp


def cast_grad(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """flat fp32 -> bf16 (or bf16 -> fp32) copy of a gradient range: the bf16 transport of the data-parallel all-reduce"""
    assert src.numel() == dst.numel() and src.is_contiguous() and dst.is_contiguous()
    to_f32 = dst.dtype == torch.float32
    assert (src.dtype, dst.dtype) in ((torch.float32, BF16), (BF16, torch.float32))
    get_lib().call("bra_cast_grad", src, dst, src.numel(), int(to_f32), current_stream(src))
    return dst


def vec_sum(x: torch.Tensor, scale: float) -> torch.Tensor:
    out = torch.empty((1,), dtype=torch.float32, device=x.device)
    get_lib().call("bra_vec_sum", x, x.numel(), scale, out, current_stream(x))
    return out


def sumsq(g: torch.Tensor, out: torch.Tensor, mask=None, ws: Optional[torch.Tensor] = None):
    """out[0] = sum(g^2) (overwritten), summed in a fixed order"""
    if ws is None:
        ws = torch.empty((1024,), dtype=torch.float32, device=g.device)
    get_lib().call("bra_sumsq", g, mask, g.numel(), out, ws, current_stream(g))


def adamw(p, g, m, v, lr, b1, b2, eps, wd, step, sumsq_t=None, max_norm=0.0, grad_scale=1.0, mask=None):
    get_lib().call("bra_adamw", p, g, m, v, mask, p.numel(), lr, b1, b2, eps, wd, step, sumsq_t, max_norm, grad_scale, current_stream(p))


def pack_params(desc_table: torch.Tensor, ndesc: int, max_elems: int):
    get_lib().call("bra_pack_params", desc_table, ndesc, max_elems, current_stream(desc_table))
