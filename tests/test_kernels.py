"""Kernel-level parity: every C-ABI entry point against a plain PyTorch fp32 statement of the same op.

Each test runs on the CPU kernel-source emulator (`backend == cpu`, part of `-m "not gpu"`) and on the real
gfx950 library (`-m gpu`).  Inputs are bf16; tolerances are those of one bf16 rounding of the output
(rel Frobenius <= 4e-3) unless the op is exact.
"""
import math

import pytest
import torch

from bioreason_amd import ops, _lib

BF = torch.bfloat16


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def rnd(*shape, dev, scale=1.0, seed=None):
    g = torch.Generator().manual_seed(seed if seed is not None else sum(shape) + len(shape))
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(dev)


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K,K2", [(128, 128, 64, 0), (200, 96, 128, 64), (130, 260, 32, 32), (1, 8, 64, 0), (257, 136, 192, 0), (8, 144, 2560, 64), (16, 40, 96, 0)])
def test_gemm_nt(backend, M, N, K, K2):
    a, b = rnd(M, K, dev=backend), rnd(N, K, dev=backend)
    a2 = rnd(M, K2, dev=backend) if K2 else None
    b2 = rnd(N, K2, dev=backend) if K2 else None
    bias, res = rnd(N, dev=backend), rnd(M, N, dev=backend)
    ref = a.float() @ b.float().T + (a2.float() @ b2.float().T if K2 else 0)
    c = ops.gemm_nt(a, b, a2=a2, b2=b2, bias=bias, res=res, alpha=0.5)
    want = (0.5 * ref + bias.float()).to(BF).float() + res.float()
    assert rel(c, want) < 4e-3
    c32 = ops.gemm_nt(a, b, a2=a2, b2=b2, out_f32=True)
    assert rel(c32, ref) < 1e-5
    acc = torch.ones(M, N, device=backend)
    ops.gemm_nt(a, b, a2=a2, b2=b2, out=acc, out_f32=True, accumulate=True)
    assert rel(acc, ref + 1) < 1e-5
    sk = torch.zeros(M, N, device=backend)
    ops.gemm_nt_splitk(a, b, sk, alpha=2.0, split_k=3)
    assert rel(sk, 2 * (a.float() @ b.float().T)) < 1e-5


def test_gemm_transpose_detect(backend):
    """A = I-like with asymmetric B: catches a C written transposed (guide §3)."""
    M = N = 128
    K = 128
    a = torch.eye(M, K).to(BF).to(backend)
    b = (torch.arange(N * K).reshape(N, K) % 37).float().to(BF).to(backend)
    c = ops.gemm_nt(a, b, out_f32=True)
    assert torch.equal(c.cpu(), b.float().T.cpu()[:M, :N].contiguous())


def test_gemm_strided_slices(backend):
    M, K, N = 96, 64, 64
    big_a = rnd(M, 3 * K, dev=backend)
    big_c = torch.zeros(M, 4 * N, dtype=BF, device=backend)
    b = rnd(N, K, dev=backend)
    a_slice = big_a[:, K:2 * K]
    c_slice = big_c[:, 2 * N:3 * N]
    ops.gemm_nt(a_slice, b, out=c_slice)
    assert rel(c_slice, a_slice.float() @ b.float().T) < 4e-3
    assert big_c[:, :2 * N].abs().sum() == 0 and big_c[:, 3 * N:].abs().sum() == 0


@pytest.mark.parametrize("M,V,K", [(70, 300, 64), (130, 1000, 128)])
def test_lmhead_logprob_and_dlogits(backend, M, V, K):
    h, e = rnd(M, K, dev=backend, scale=0.5), rnd(V, K, dev=backend, scale=0.5)
    tgt = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(1)).to(torch.int32).to(backend)
    logp, lse = ops.lmhead_logprob(h, e, tgt)
    logits = (h.float() @ e.float().T).to(BF).float()          # the reference's lm_head output is bf16
    ref_lse = torch.logsumexp(logits, -1)
    ref_lp = logits.gather(1, tgt.long()[:, None])[:, 0] - ref_lse
    assert (lse.cpu() - ref_lse.cpu()).abs().max() < 2e-3
    assert (logp.cpu() - ref_lp.cpu()).abs().max() < 2e-3
    coef = torch.randn(M, generator=torch.Generator().manual_seed(2)).to(backend)
    dl = ops.lmhead_dlogits(h, e, tgt, lse, coef)
    onehot = torch.zeros(M, V, device=backend).scatter_(1, tgt.long()[:, None], 1.0)
    want = coef[:, None] * (onehot - torch.softmax(logits, -1))
    assert rel(dl, want) < 6e-3


# ----------------------------------------------------------------------------- norms
@pytest.mark.parametrize("rows,cols", [(5, 64), (9, 2048), (4, 520)])
def test_rmsnorm_fwd_bwd(backend, rows, cols):
    x, w = rnd(rows, cols, dev=backend), (1 + 0.1 * rnd(cols, dev=backend).float()).to(BF)
    eps = 1e-6
    y = ops.rmsnorm_fwd(x, w, eps)
    xf = x.float().requires_grad_(True)
    normed = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    ref = w.float() * normed
    assert rel(y, w.float() * normed.detach().to(BF).float()) < 4e-3
    dy, dres = rnd(rows, cols, dev=backend, seed=5), rnd(rows, cols, dev=backend, seed=6)
    ref.backward(dy.float())
    dx = ops.rmsnorm_bwd(dy, x, w, eps, dres=dres)
    assert rel(dx, xf.grad + dres.float()) < 4e-3
    dx2 = ops.rmsnorm_bwd(dy, x, w, eps)
    assert rel(dx2, xf.grad) < 4e-3


def test_layernorm_fwd(backend):
    x, w, b = rnd(7, 1024, dev=backend), rnd(1024, dev=backend), rnd(1024, dev=backend)
    y = ops.layernorm_fwd(x, w, b, 1e-12)
    ref = torch.nn.functional.layer_norm(x.float(), (1024,), w.float(), b.float(), 1e-12)
    assert rel(y, ref) < 4e-3


def test_swiglu_fwd_bwd(backend):
    rows, F = 9, 136
    gu = rnd(rows, 2 * F, dev=backend)
    act = ops.swiglu_fwd(gu)
    guf = gu.float().requires_grad_(True)
    ref = torch.nn.functional.silu(guf[:, :F]) * guf[:, F:]
    assert rel(act, ref) < 5e-3
    dact = rnd(rows, F, dev=backend, seed=3)
    ref.backward(dact.float())
    dgu = ops.swiglu_bwd(gu, dact)
    assert rel(dgu, guf.grad) < 5e-3


def _rope_tables(npos, hd, theta, dev):
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.arange(npos, dtype=torch.float32)[:, None] * inv[None, :]
    return fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev)


def _rope_ref(x, cos, sin, pos):
    # x [B,S,H,hd] float; rotate-half
    half = x.shape[-1] // 2
    c = torch.cat([cos[pos], cos[pos]], -1)[:, :, None, :]
    s = torch.cat([sin[pos], sin[pos]], -1)[:, :, None, :]
    rot = torch.cat([-x[..., half:], x[..., :half]], -1)
    return x * c + rot * s


@pytest.mark.parametrize("hd,Hq,Hkv,norm,qscale", [(128, 4, 2, True, 1.0), (64, 2, 2, False, 64 ** -0.5), (32, 4, 1, True, 1.0)])
def test_qk_norm_rope_fwd_bwd(backend, hd, Hq, Hkv, norm, qscale):
    B, S = 2, 9
    T = B * S
    eps = 1e-6
    qkv = rnd(T, (Hq + 2 * Hkv) * hd, dev=backend)
    qw = (1 + 0.1 * rnd(hd, dev=backend).float()).to(BF) if norm else None
    kw = (1 + 0.1 * rnd(hd, dev=backend, seed=9).float()).to(BF) if norm else None
    cos, sin = _rope_tables(64, hd, 10000.0, backend)
    pos = (torch.arange(S).repeat(B) + 3).to(torch.int32).to(backend)
    q = torch.empty(B, S, Hq, hd, dtype=BF, device=backend)
    kc = torch.zeros(B, Hkv, S + 5, hd, dtype=BF, device=backend)     # cache layout, append at offset 5
    vc = torch.zeros(B, Hkv, S + 5, hd, dtype=BF, device=backend)
    ops.qk_norm_rope_fwd(qkv, qw, kw, cos, sin, pos, S, Hq, Hkv, hd, eps, qscale, q, kc.permute(0, 2, 1, 3), vc.permute(0, 2, 1, 3), s_off=5)

    def ref_fn(x):
        x = x.view(B, S, Hq + 2 * Hkv, hd)
        qq, kk, vv = x[:, :, :Hq], x[:, :, Hq:Hq + Hkv], x[:, :, Hq + Hkv:]
        if norm:
            qq = qw.float() * (qq * torch.rsqrt(qq.pow(2).mean(-1, keepdim=True) + eps))
            kk = kw.float() * (kk * torch.rsqrt(kk.pow(2).mean(-1, keepdim=True) + eps))
        qq = qq * qscale
        p = pos.long().view(B, S)
        return _rope_ref(qq, cos, sin, p), _rope_ref(kk, cos, sin, p), vv

    xf = qkv.float().requires_grad_(True)
    rq, rk, rv = ref_fn(xf)
    assert rel(q, rq) < 6e-3
    assert rel(kc[:, :, 5:].permute(0, 2, 1, 3), rk) < 6e-3
    assert torch.equal(vc[:, :, 5:].permute(0, 2, 1, 3).float().cpu(), rv.detach().cpu())
    assert kc[:, :, :5].abs().sum() == 0
    dq, dk, dv = rnd(B, S, Hq, hd, dev=backend, seed=1), rnd(B, S, Hkv, hd, dev=backend, seed=2), rnd(B, S, Hkv, hd, dev=backend, seed=3)
    (rq * dq.float()).sum().backward(retain_graph=True)
    (rk * dk.float()).sum().backward(retain_graph=True)
    (rv * dv.float()).sum().backward()
    dqkv = ops.qk_norm_rope_bwd(qkv, qw, kw, cos, sin, pos, S, Hq, Hkv, hd, eps, qscale, dq, dk, dv)
    assert rel(dqkv, xf.grad) < 6e-3


# ----------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, kmask, causal, scale, q_off):
    # q [B,Sq,Hq,hd], k/v [B,Sk,Hkv,hd] float -> o [B,Sq,Hq,hd], lse [B,Hq,Sq]
    B, Sq, Hq, hd = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    g = Hq // Hkv
    kk = k.repeat_interleave(g, dim=2)
    vv = v.repeat_interleave(g, dim=2)
    s = torch.einsum("bqhd,bkhd->bhqk", q, kk) * scale
    ok = torch.ones(B, 1, Sq, Sk, dtype=torch.bool, device=q.device)
    if kmask is not None:
        ok = ok & (kmask.bool()[:, None, None, :])
    if causal:
        i = torch.arange(Sq, device=q.device)[:, None]
        j = torch.arange(Sk, device=q.device)[None, :]
        ok = ok & (j <= i + q_off)[None, None]
    s = s.masked_fill(~ok, float("-inf"))
    lse = torch.logsumexp(s, -1)
    p = torch.exp(s - lse[..., None])
    p = torch.nan_to_num(p, nan=0.0)
    o = torch.einsum("bhqk,bkhd->bqhd", p, vv)
    return o, lse


@pytest.mark.parametrize("hd,Hq,Hkv,Sq,Sk,causal,pad", [
    (128, 4, 2, 150, 150, True, "left"),
    (64, 2, 2, 100, 100, False, "right"),
    (32, 2, 1, 70, 70, True, None),
    (128, 2, 1, 40, 200, True, "left"),      # chunked prefill against a longer key range
    (128, 2, 2, 300, 300, False, None),      # 8-wave workgroups, several unmasked tiles (fast path) at hd 128
    (64, 2, 1, 260, 260, True, None),
])
def test_attn_fwd_bwd(backend, hd, Hq, Hkv, Sq, Sk, causal, pad):
    B = 2
    q, k, v = rnd(B, Sq, Hq, hd, dev=backend, seed=1), rnd(B, Sk, Hkv, hd, dev=backend, seed=2), rnd(B, Sk, Hkv, hd, dev=backend, seed=3)
    kmask = torch.ones(B, Sk, dtype=torch.uint8, device=backend)
    if pad == "left":
        kmask[0, :7] = 0
    elif pad == "right":
        kmask[1, Sk - 13:] = 0
    scale = hd ** -0.5
    q_off = Sk - Sq
    vt = ops.head_transpose(v)
    assert torch.equal(vt[..., :Sk].cpu(), v.permute(0, 2, 3, 1).cpu()) and vt[..., Sk:].abs().sum() == 0
    o, lse = ops.attn_fwd(q, k, vt, kmask if pad else None, causal, scale)
    qf, kf, vf = q.float().requires_grad_(True), k.float().requires_grad_(True), v.float().requires_grad_(True)
    ro, rlse = _attn_ref(qf, kf, vf, kmask if pad else None, causal, scale, q_off)
    # rows with no visible key are unspecified in the reference (pad queries); compare the others
    valid_q = torch.isfinite(rlse)                        # [B,Hq,Sq]
    vq = valid_q.permute(0, 2, 1)[..., None].cpu()
    assert rel(o.cpu() * vq, ro.detach().cpu() * vq) < 6e-3
    assert ((lse.cpu() - rlse.detach().cpu()).abs() * valid_q.cpu()).nan_to_num(0).max() < 2e-2
    dout = rnd(B, Sq, Hq, hd, dev=backend, seed=4) * vq.to(backend).to(BF)
    (ro.nan_to_num(0) * dout.float()).sum().backward()
    dq, dk, dv = ops.attn_bwd(q, k, v, o, dout, lse, kmask if pad else None, causal, scale)
    assert rel(dq.cpu() * vq, qf.grad.cpu() * vq) < 1.5e-2
    assert rel(dk, kf.grad) < 1.5e-2
    assert rel(dv, vf.grad) < 1.5e-2



@pytest.mark.parametrize("hd,Hq,Hkv,Sq,Sk,causal,pad,spike", [
    (128, 2, 1, 700, 700, True, "left", True),        # eleven key tiles: the interleaved loop in both slot parities + its remainder trip
    (128, 2, 2, 256, 700, True, "holes", False),      # a prefix (Sk > Sq) with masked keys in the middle of tiles
    (128, 2, 2, 513, 513, True, None, True),          # a last workgroup with one live query row
    (64, 2, 2, 400, 400, False, "right", True),
    (64, 2, 1, 1024, 1024, False, None, False),       # the NT-v2 encoder's sequence length, every step unmasked
    (128, 2, 1, 129, 300, True, "right", False),
])
def test_attn_fwd_pipelined_kernel(debug_backend, hd, Hq, Hkv, Sq, Sk, causal, pad, spike):
    """k_attn4.hip (4 waves x 64 queries, software-pipelined 32-key steps, deferred rescaling) against the fp32 statement
    (TF:qwen3:185-207, TF:esm:292-317) and against the 8-wave kernel of rounds 1-5 on the same inputs.  `spike`: a late key that
    dominates some rows and an early one that dominates others — the running maximum grows long after the first step, so the rare
    path (redo the step against the true maximum, rescale O and l) runs in the middle of the loop; rows without any visible key stay 0."""
    backend = debug_backend
    lib = _lib.get_lib()
    B = 2
    q, k, v = rnd(B, Sq, Hq, hd, dev=backend, seed=1), rnd(B, Sk, Hkv, hd, dev=backend, seed=2), rnd(B, Sk, Hkv, hd, dev=backend, seed=3)
    if spike:
        k[:, Sk - 40] = q[:, Sq - 3, :Hkv] * 6
        k[:, 70] = q[:, Sq // 2, :Hkv] * 5
    kmask = torch.ones(B, Sk, dtype=torch.uint8, device=backend)
    if pad == "left":
        kmask[0, :75] = 0
    elif pad == "right":
        kmask[1, Sk - 45:] = 0
    elif pad == "holes":
        kmask[0, 33:40] = 0
        kmask[1, 200:290] = 0
    km = kmask if pad else None
    scale = hd ** -0.5
    vt = ops.head_transpose(v)
    res = {}
    try:
        for on in (1, 0):
            lib.call("bra_attn_set_fwd4", on)
            res[on] = ops.attn_fwd(q, k, vt, km, causal, scale, nsplit=1)
    finally:
        lib.call("bra_attn_set_fwd4", 1)
    ro, rlse = _attn_ref(q.float(), k.float(), v.float(), km, causal, scale, Sk - Sq)
    valid_q = torch.isfinite(rlse).cpu()
    vq = valid_q.permute(0, 2, 1)[..., None]
    (o4, l4), (o8, l8) = [(o.cpu(), l.cpu()) for o, l in (res[1], res[0])]
    assert rel(o4 * vq, ro.cpu() * vq) < 6e-3
    assert ((l4 - rlse.cpu()).abs() * valid_q).nan_to_num(0).max() < 2e-2
    assert rel(o4 * vq, o8 * vq) < 5e-3                                  # two roundings of P apart at most
    if (~vq).any():
        assert o4[~vq.expand_as(o4)].float().abs().max() == 0 and torch.equal(l4[~valid_q], l8[~valid_q])


@pytest.mark.parametrize("hd,Hq,Hkv,Sq,Sk,causal,pad", [
    (128, 2, 1, 700, 700, True, "left"),              # eleven tiles per (head, block): the interleaved loops in both slot parities + remainder
    (128, 4, 2, 256, 700, True, "holes"),             # a prefix (Sk > Sq), masked keys inside tiles, a GQA group of two q-heads per kv-head
    (128, 2, 2, 513, 513, True, None),                # last workgroups with one live row
    (64, 2, 2, 400, 400, False, "right"),
    (128, 2, 1, 129, 300, True, "right"),
])
def test_attn_bwd_pipelined_kernels(debug_backend, hd, Hq, Hkv, Sq, Sk, causal, pad):
    """k_attn4b.hip (dQ: unit = key step x query block; dK / dV: unit = query half x key block; Q / dO resp. K / V fragments and the
    accumulators resident in AGPRs) against the fp32 statement (TF:qwen3:185-207 backward) and against the kernels of rounds 1-5 on the
    same inputs — the two differ only in the association of fp32 sums and in where `scale` enters dS"""
    backend = debug_backend
    lib = _lib.get_lib()
    B = 2
    q, k, v = rnd(B, Sq, Hq, hd, dev=backend, seed=1), rnd(B, Sk, Hkv, hd, dev=backend, seed=2), rnd(B, Sk, Hkv, hd, dev=backend, seed=3)
    kmask = torch.ones(B, Sk, dtype=torch.uint8, device=backend)
    if pad == "left":
        kmask[0, :75] = 0
    elif pad == "right":
        kmask[1, Sk - 45:] = 0
    elif pad == "holes":
        kmask[0, 33:40] = 0
        kmask[1, 200:290] = 0
    km = kmask if pad else None
    scale = hd ** -0.5
    vt = ops.head_transpose(v)
    o, lse = ops.attn_fwd(q, k, vt, km, causal, scale, nsplit=1)
    qf, kf, vf = q.float().requires_grad_(True), k.float().requires_grad_(True), v.float().requires_grad_(True)
    ro, rlse = _attn_ref(qf, kf, vf, km, causal, scale, Sk - Sq)
    valid_q = torch.isfinite(rlse)
    vq = valid_q.permute(0, 2, 1)[..., None]
    dout = rnd(B, Sq, Hq, hd, dev=backend, seed=4) * vq.to(BF)
    (ro.nan_to_num(0) * dout.float()).sum().backward()
    res = {}
    try:
        for mask in (3, 0):
            lib.call("bra_attn_set_bwd4", mask)
            res[mask] = [t.float().cpu() for t in ops.attn_bwd(q, k, v, o, dout, lse, km, causal, scale, nsplit=(1, 1))]
    finally:
        lib.call("bra_attn_set_bwd4", 3)
    vq = vq.cpu()
    for i, (nm, want) in enumerate((("dq", qf.grad), ("dk", kf.grad), ("dv", vf.grad))):
        w = want.cpu() * vq if nm == "dq" else want.cpu()
        new, old = (res[3][i] * vq, res[0][i] * vq) if nm == "dq" else (res[3][i], res[0][i])
        assert rel(new, w) < 1.5e-2, nm
        assert rel(new, old) < 3e-4, nm
        assert torch.isfinite(res[3][i]).all(), nm


@pytest.mark.parametrize("hd,Hq,Hkv,Sq,Sk,causal,pad,ns", [(128, 4, 2, 300, 300, True, "left", (2, 2)), (128, 2, 1, 256, 700, True, None, (3, 1)),
                                                         (64, 2, 2, 200, 200, False, "right", (2, 3)), (128, 4, 1, 520, 520, True, None, (4, 4)),
                                                         (128, 2, 2, 600, 600, True, None, (1, 2))])
def test_attn_bwd_split_equals_unsplit(backend, hd, Hq, Hkv, Sq, Sk, causal, pad, ns):
    """the backward with the dQ kernel's key range and the one-launch dK + dV kernel's (q-head, query tile) loop cut into parts + the
    sum launches (bra_attn_bwd_split) against the one-part backward: the parts stay fp32 and are added in order, so the results differ
    from the unsplit kernels only by fp32 re-association before the one bf16 rounding (TF:qwen3:185-207 backward)"""
    B = 2
    q, k, v = rnd(B, Sq, Hq, hd, dev=backend, seed=1), rnd(B, Sk, Hkv, hd, dev=backend, seed=2), rnd(B, Sk, Hkv, hd, dev=backend, seed=3)
    do = rnd(B, Sq, Hq, hd, dev=backend, seed=4)
    kmask = torch.ones(B, Sk, dtype=torch.uint8, device=backend)
    if pad == "left":
        kmask[0, :70] = 0
    elif pad == "right":
        kmask[1, Sk - 13:] = 0
    scale = hd ** -0.5
    vt = ops.head_transpose(v)
    o, lse = ops.attn_fwd(q, k, vt, kmask if pad else None, causal, scale, nsplit=1)
    want = ops.attn_bwd(q, k, v, o, do, lse, kmask if pad else None, causal, scale, nsplit=(1, 1))
    got = ops.attn_bwd(q, k, v, o, do, lse, kmask if pad else None, causal, scale, nsplit=ns)
    for g_, w_, nm in zip(got, want, ("dq", "dk", "dv")):
        assert rel(g_, w_) < 3e-3, nm
        assert torch.isfinite(g_.float()).all()
    assert ops.attn_bwd_split_parts(8, 16, 8, 2436, 2436, 128, True) == (1, 1)
    assert ops.attn_bwd_split_parts(1, 16, 8, 2180, 2180, 128, True) == (2, 4)
    assert ops.attn_bwd_split_parts(8, 16, 8, 256, 2436, 128, True) == (2, 1)


@pytest.mark.parametrize("hd,Hq,Hkv,Sq,Sk,causal,pad,ns", [(128, 4, 2, 300, 300, True, "left", 2), (128, 2, 1, 256, 700, True, None, 3),
                                                         (64, 2, 2, 200, 200, False, "right", 2), (128, 4, 2, 520, 520, True, None, 4)])
def test_attn_fwd_split_equals_unsplit(backend, hd, Hq, Hkv, Sq, Sk, causal, pad, ns):
    """the forward with every query block's key range cut into `ns` parts + the merge launch (bra_attn_fwd_split: grids that cannot fill
    the chip) against the one-part forward: same softmax(QK^T)V up to the fp32 re-association of the merge (one extra bf16 rounding
    never: the parts stay fp32), same LSE; causal with a prefix (Sk > Sq), padding on either side, a part that sees no key of a query
    (TF:qwen3:185-207)"""
    B = 2
    q, k, v = rnd(B, Sq, Hq, hd, dev=backend, seed=1), rnd(B, Sk, Hkv, hd, dev=backend, seed=2), rnd(B, Sk, Hkv, hd, dev=backend, seed=3)
    kmask = torch.ones(B, Sk, dtype=torch.uint8, device=backend)
    if pad == "left":
        kmask[0, :70] = 0                      # (more than one key tile: the first part of row 0's early queries sees nothing)
    elif pad == "right":
        kmask[1, Sk - 13:] = 0
    scale = hd ** -0.5
    vt = ops.head_transpose(v)
    o1, l1 = ops.attn_fwd(q, k, vt, kmask if pad else None, causal, scale, nsplit=1)
    o2, l2 = ops.attn_fwd(q, k, vt, kmask if pad else None, causal, scale, nsplit=ns)
    live = torch.isfinite(l1.cpu()) & (l1.cpu() > -1e29)
    assert rel(o2, o1) < 3e-3
    assert (l2.cpu()[live] - l1.cpu()[live]).abs().max() < 2e-4
    assert torch.equal(l2.cpu()[~live], l1.cpu()[~live])
    # the automatic choice: one part for grids that fill the chip, more for one prompt / a short completion segment
    assert ops.attn_fwd_split_parts(8, 16, 2436, 2436, 128, True) == 1
    assert ops.attn_fwd_split_parts(1, 16, 2180, 2180, 128, True) == 2
    assert ops.attn_fwd_split_parts(8, 16, 256, 2436, 128, True) == 2
    assert ops.attn_fwd_split_parts(1, 16, 1090, 1090, 128, True) == 4


@pytest.mark.parametrize("hd,Hq,Hkv,L", [(128, 4, 2, 300), (64, 2, 2, 129), (32, 4, 1, 64)])
def test_attn_decode(backend, hd, Hq, Hkv, L):
    B, Smax = 3, 320
    q = rnd(B, Hq, hd, dev=backend, seed=1)
    kc, vc = rnd(B, Hkv, Smax, hd, dev=backend, seed=2), rnd(B, Hkv, Smax, hd, dev=backend, seed=3)
    kmask = torch.ones(B, Smax, dtype=torch.uint8, device=backend)
    kmask[1, :11] = 0
    o = ops.attn_decode(q, kc, vc, kmask, L, hd ** -0.5)
    ro, _ = _attn_ref(q.float()[:, None], kc[:, :, :L].permute(0, 2, 1, 3).float(), vc[:, :, :L].permute(0, 2, 1, 3).float(),
                      kmask[:, :L], False, hd ** -0.5, 0)
    assert rel(o.view(B, Hq, hd), ro[:, 0]) < 6e-3


# ----------------------------------------------------------------------------- scatter / movement
def test_dna_scatter_plan_and_embed(backend):
    B, P, H, V = 3, 50, 64, 97
    nseq, Sd = 5, 12
    dna_id = 90
    g = torch.Generator().manual_seed(0)
    batch_idx_map = [0, 0, 1, 2, 2]
    dmask = torch.ones(nseq, Sd, dtype=torch.uint8)
    dmask[1, 7:] = 0
    dmask[3, 10:] = 0
    valid = dmask.sum(1).tolist()
    per_sample = [sum(valid[i] for i in range(nseq) if batch_idx_map[i] == b) for b in range(B)]
    ids = torch.randint(0, 89, (B, P), generator=g)
    for b in range(B):
        start = 3 + b
        ids[b, start:start + per_sample[b]] = dna_id
    order = sorted(range(nseq), key=lambda i: batch_idx_map[i])
    ids32 = ids.to(torch.int32).to(backend)
    tok_src = torch.empty(B * P, dtype=torch.int32, device=backend)
    counts = torch.zeros(2, dtype=torch.int32, device=backend)
    ops.dna_scatter_plan(ids32.view(-1), dna_id, dmask.to(backend), torch.tensor(order, dtype=torch.int32, device=backend), tok_src, counts)
    assert counts.tolist() == [sum(per_sample), sum(valid)]
    emb = rnd(V, H, dev=backend, seed=1)
    proj = rnd(nseq * Sd, H, dev=backend, seed=2)
    out = torch.empty(B * P, H, dtype=BF, device=backend)
    ops.embed_scatter_fwd(ids32.view(-1), tok_src, emb, proj, out)
    # reference: dna_llm.py:163-177 + :216-229
    ref = emb.float()[ids.view(-1).to(backend).long()].clone()
    result = [[] for _ in range(B)]
    pv = proj.float().view(nseq, Sd, H)
    for si, bi in enumerate(batch_idx_map):
        result[bi].append(pv[si, :valid[si]])
    flat = torch.cat([torch.cat(r, 0) for r in result], 0)
    ref[(ids.view(-1) == dna_id).to(backend)] = flat
    assert torch.equal(out.float().cpu(), ref.cpu())
    dout = rnd(B * P, H, dev=backend, seed=3)
    ddna = torch.zeros(nseq * Sd, H, dtype=BF, device=backend)
    ops.embed_scatter_bwd(tok_src, dout, ddna)
    want = torch.zeros(nseq * Sd, H)
    src = tok_src.cpu().long()
    want[src[src >= 0]] = dout.float().cpu()[src >= 0]
    assert torch.equal(ddna.float().cpu(), want)


def test_dna_scatter_mismatch_counts(backend):
    ids32 = torch.tensor([5, 9, 9, 9, 1], dtype=torch.int32, device=backend)
    dmask = torch.ones(1, 2, dtype=torch.uint8, device=backend)
    tok_src = torch.empty(5, dtype=torch.int32, device=backend)
    counts = torch.zeros(2, dtype=torch.int32, device=backend)
    ops.dna_scatter_plan(ids32, 9, dmask, torch.zeros(1, dtype=torch.int32, device=backend), tok_src, counts)
    assert counts.tolist() == [3, 2]      # caller raises ValueError as dna_llm.py:222-225 does


def test_transpose_gather_colsum(backend):
    x = rnd(70, 104, dev=backend)[:, :100]       # 100 used columns, row pitch 104
    xt = ops.transpose2d(x)
    assert xt.shape == (100, 72)
    assert torch.equal(xt[:, :70].cpu(), x.T.cpu()) and xt[:, 70:].abs().sum() == 0
    rows = torch.tensor([3, 0, 69, 5], dtype=torch.int32, device=backend)
    x2 = rnd(70, 64, dev=backend)
    assert torch.equal(ops.gather_rows(rows, x2).cpu(), x2[rows.long()].cpu())
    sc = ops.scatter_rows(rows, x2[:4].contiguous(), 70)
    assert torch.equal(sc[rows.long()].cpu(), x2[:4].cpu()) and sc.float().abs().sum() == x2[:4].float().abs().sum()
    cs = torch.ones(100, device=backend)
    ops.colsum(x, cs)
    assert rel(cs, x.float().sum(0) + 1) < 1e-5


# ----------------------------------------------------------------------------- GRPO math / optimiser
def test_eos_mask_and_advantage(backend):
    ids = torch.tensor([[4, 5, 2, 7, 2], [1, 1, 1, 1, 1], [2, 0, 0, 0, 0]], dtype=torch.int32, device=backend)
    mask, lengths = ops.eos_mask(ids, 2)
    assert mask.tolist() == [[1, 1, 1, 0, 0], [1, 1, 1, 1, 1], [1, 0, 0, 0, 0]]
    assert lengths.tolist() == [3, 5, 1]
    r = torch.tensor([[1.0, 0.5], [0.0, 0.0], [2.0, 0.5], [0.5, 0.5], [1.0, 1.0], [1.0, 1.0], [1.0, 1.0], [1.0, 1.0]], device=backend)
    adv, gm, gs = ops.group_advantage(r, 4)
    tot = r.sum(1).view(-1, 4)
    want = ((tot - tot.mean(1, keepdim=True)) / (tot.std(1, keepdim=True) + 1e-4)).view(-1)
    assert torch.allclose(adv.cpu(), want.cpu(), atol=1e-5)


@pytest.mark.parametrize("use_old,beta", [(False, 0.04), (True, 0.04), (True, 0.0)])
def test_grpo_loss(backend, use_old, beta):
    B, C = 4, 37
    g = torch.Generator().manual_seed(0)
    lp = (-torch.rand(B, C, generator=g) * 3).to(backend)
    old = (lp + 0.3 * torch.randn(B, C, generator=g).to(backend)) if use_old else None
    ref = lp + 0.2 * torch.randn(B, C, generator=g).to(backend)
    adv = torch.randn(B, generator=g).to(backend)
    mask = (torch.rand(B, C, generator=g) > 0.3).to(torch.int32).to(backend)
    mask[:, 0] = 1
    out3, dlogp = ops.grpo_loss(lp, old, ref if beta else None, adv, mask, 0.2, 0.2, beta)
    # grpo_trainer.py:786-814 restated
    lpt = lp.clone().requires_grad_(True)
    o = old if use_old else lpt.detach()
    c1 = torch.exp(lpt - o)
    c2 = torch.clamp(c1, 0.8, 1.2)
    l1, l2 = c1 * adv[:, None], c2 * adv[:, None]
    ptl = -torch.min(l1, l2)
    if beta:
        kl = torch.exp(ref - lpt) - (ref - lpt) - 1
        ptl = ptl + beta * kl
    m = mask.float()
    loss = ((ptl * m).sum(1) / m.sum(1)).mean()
    loss.backward()
    assert abs(out3[0].item() - loss.item()) < 1e-5
    assert torch.allclose(dlogp.cpu(), lpt.grad.cpu(), atol=1e-6)
    if beta:
        assert abs(out3[1].item() - ((kl * m).sum(1) / m.sum(1)).mean().item()) < 1e-5
    assert abs(out3[2].item() - (((l1 < l2).float() * m).sum() / m.sum()).item()) < 1e-6


def test_adamw_with_clip(backend):
    n = 1000
    g0 = torch.Generator().manual_seed(0)
    p = torch.randn(n, generator=g0).to(backend)
    grads = [torch.randn(n, generator=g0).to(backend) * 3 for _ in range(3)]
    ref_p = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    m, v = torch.zeros(n, device=backend), torch.zeros(n, device=backend)
    for step, g in enumerate(grads, 1):
        ref_p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        ss = torch.zeros(1, device=backend)
        ops.sumsq(g, ss)
        ops.adamw(p, g, m, v, 1e-2, 0.9, 0.999, 1e-8, 0.01, step, sumsq_t=ss, max_norm=1.0)
    assert torch.allclose(p.cpu(), ref_p.detach().cpu(), atol=2e-6)


def test_sampler_greedy_and_topk(backend):
    B, V = 3, 5000
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(B, V, generator=g).to(backend)
    logits[1, 77] = logits[1].max() + 1      # unique max
    logits[2, 10] = logits[2, 4000] = logits[2].max() + 2   # tie -> first index (torch.argmax)
    out = torch.empty(B, dtype=torch.int32, device=backend)
    ops.sample(logits, 1.0, 0, 1.0, False, 0, None, None, 0, out)
    assert out.tolist() == logits.argmax(-1).tolist()
    assert out[2].item() == 10
    # sampling: draws must come from the top-k / top-p support with plausible frequencies
    step = torch.zeros(1, dtype=torch.int32, device=backend)
    T, k, p = 0.6, 20, 0.95
    draws = []
    on_gpu = backend.type == "cuda"                  # (the emulator run checks the support with a few draws; frequencies on the GPU)
    for s in range(200 if on_gpu else 12):
        step.fill_(s)
        ops.sample(logits, T, k, p, True, 1234, step, None, 0, out)
        draws.append(out.clone().cpu())
    draws = torch.stack(draws)                      # [200, B]
    sc = logits.cpu() / T
    topv, topi = sc.topk(k, -1)
    pr = torch.softmax(topv, -1)
    # HF top-p on the ascending list
    asc = torch.flip(pr, [-1])
    remove = torch.flip(asc.cumsum(-1) <= (1 - p), [-1])
    remove[:, 0] = False
    pr = pr.masked_fill(remove, 0)
    pr = pr / pr.sum(-1, keepdim=True)
    for b in range(B):
        support = set(topi[b][pr[b] > 0].tolist())
        assert set(draws[:, b].tolist()) <= support
        top_tok = topi[b, 0].item()
        freq = (draws[:, b] == top_tok).float().mean().item()
        assert not on_gpu or abs(freq - pr[b, 0].item()) < 0.15
    fin = torch.tensor([0, 1, 0], dtype=torch.uint8, device=backend)
    ops.sample(logits, T, k, p, True, 1, step, fin, 42, out)
    assert out[1].item() == 42
    # fused tail: the drawing wave gathers x = E[token] and the row's RMSNorm statistic; same draw as the plain call
    E = rnd(V, 64, dev=backend)
    x = torch.zeros(B, 64, dtype=BF, device=backend)
    ss = torch.full((8, 32), 7.0, device=backend)
    out2 = torch.empty_like(out)
    step.fill_(5)
    ops.sample(logits, T, k, p, True, 1234, step, None, 0, out)
    ops.sample(logits, T, k, p, True, 1234, step, None, 0, out2, embed=(E, x, ss))
    assert out.tolist() == out2.tolist()
    assert torch.equal(x.cpu(), E[out2.long()].cpu())
    assert rel(ss[:B, 0], (x.float() ** 2).sum(1)) < 1e-5 and float(ss[:B, 1:].abs().max()) == 0 and float(ss[B:].min()) == 7.0
    pos, a_, b_ = torch.arange(5, dtype=torch.int32, device=backend), torch.zeros(1, dtype=torch.int32, device=backend), torch.full((1,), 9, dtype=torch.int32, device=backend)
    ops.advance_counters(pos, a_, b_)
    assert pos.tolist() == [1, 2, 3, 4, 5] and a_.item() == 1 and b_.item() == 10
    # with the rotary rows of the advanced positions (token loop: one launch bumps the counters and refreshes cos | sin rows)
    for n, hd in ((3, 32), (8, 128), (16, 128)):
        cosT, sinT = torch.rand(40, hd // 2, device=backend), torch.rand(40, hd // 2, device=backend)
        pos = (torch.arange(n, dtype=torch.int32, device=backend) * 2) % 7 + 3
        want_pos = pos + 1
        rows = torch.zeros(n, hd, device=backend)
        ops.advance_counters(pos, a_, b_, rope=(cosT, sinT, hd, rows))
        assert torch.equal(pos.cpu(), want_pos.cpu())
        assert torch.equal(rows.cpu(), torch.cat([cosT[want_pos.long()], sinT[want_pos.long()]], -1).cpu())


def test_sampler_distribution_matches_oracle_warpers(backend):
    """full-support check of temperature -> top-k -> top-p -> multinomial against the oracle's restatement of HF's warpers
    (oracle.grpo_math.warp_probs, TF:generation/logits_process.py:238,473,542): support equality + a chi-square test of the
    draw frequencies over every token of the warped distribution"""
    from oracle.grpo_math import warp_probs
    B, V = 4, 6000
    g = torch.Generator().manual_seed(3)
    logits = (torch.randn(B, V, generator=g) * 2.0)
    logits[1] *= 0.2                                  # flat row: top-p keeps (almost) all k
    logits[2, 123] += 6.0                             # peaked row: top-p cuts deep
    dl = logits.to(backend)
    out = torch.empty(B, dtype=torch.int32, device=backend)
    step = torch.zeros(1, dtype=torch.int32, device=backend)
    n = 4000 if backend.type == "cuda" else 10      # the emulator run checks the support; the statistics run on the GPU
    T, k, p = 0.6, 20, 0.95
    want = warp_probs(logits, T, k, p)                # [B, V], zeros outside the support
    counts = torch.zeros(B, V)
    for s_ in range(n):
        step.fill_(s_)
        ops.sample(dl, T, k, p, True, 99, step, None, 0, out)
        counts[torch.arange(B), out.cpu().long()] += 1
    for b in range(B):
        sup = want[b] > 0
        assert counts[b][~sup].sum() == 0, "draw outside the oracle's support"
        e = want[b][sup] * n
        chi2 = (((counts[b][sup] - e) ** 2) / e).sum().item()
        dof = int(sup.sum()) - 1
        # chi-square upper tail: mean dof, sd sqrt(2 dof); 5 sd + small-expectation slack
        assert chi2 <= dof + 5.0 * (2.0 * max(dof, 1)) ** 0.5 + 5.0, (b, chi2, dof)


@pytest.mark.parametrize("top_k,top_p", [(0, 0.9), (0, 1.0), (100, 0.95), (300, 0.5)])
def test_sampler_general_top_k(backend, top_k, top_p):
    """HF's top_k = 0 ("disabled": top-p over the whole vocabulary) and top_k > 64 go to the general kernel (bra_sample_full): its
    support equals the support of the installed warpers' distribution (oracle.grpo_math.warp_probs == Temperature -> TopK -> TopP,
    tests/test_oracle_pinned.py) up to tokens within fp32 rounding of the top-p boundary, and its draw frequencies pass a chi-square
    test against it (GPU run; the emulator run checks the support with a few draws)"""
    from oracle.grpo_math import warp_probs
    B, V = 3, (6000 if backend.type == "cuda" else 400)
    g = torch.Generator().manual_seed(17 + top_k)
    logits = torch.randn(B, V, generator=g) * 2.0
    logits[1] *= 0.3
    logits[2, 7] += 5.0
    dl = logits.to(backend)
    T = 0.7
    want = warp_probs(logits, T, top_k, top_p).double()
    # tokens whose membership hangs on rounding: within 1e-5 of the top-p boundary in cumulative mass
    sc = logits.double() / T
    if top_k > 0:
        kth = torch.topk(sc, min(top_k, V))[0][..., -1, None]
        sc = sc.masked_fill(sc < kth, float("-inf"))
    pr = torch.softmax(sc, -1)
    srt, idx = torch.sort(pr, descending=True)
    above = srt.cumsum(-1) - srt                                           # mass strictly before each token in descending order
    fuzzy = torch.zeros(B, V, dtype=torch.bool).scatter(1, idx, (above - top_p).abs() < 1e-5)
    out = torch.empty(B, dtype=torch.int32, device=backend)
    lp = torch.empty(B, device=backend)
    n = 3000 if backend.type == "cuda" else 30
    counts = torch.zeros(B, V, dtype=torch.double)
    for s_ in range(n):
        ops.sample(dl, T, top_k, top_p, True, 5, s_ if s_ % 2 else torch.tensor([s_], dtype=torch.int32, device=backend), None, 0, out, out_logp=lp)
        counts[torch.arange(B), out.cpu().long()] += 1
        b0 = 0
        assert abs(float(lp[b0]) - float(torch.log(want[b0, int(out[b0])] + 1e-300))) < 2e-3 or bool(fuzzy[b0].any())
    for b in range(B):
        sup = want[b] > 0
        assert counts[b][~sup & ~fuzzy[b]].sum() == 0, "draw outside the warpers' support"
        if backend.type == "cuda":
            big = sup & (want[b] * n >= 5)                                 # chi-square over the tokens with an expectation >= 5; the rest pooled
            e = torch.cat([want[b][big] * n, (want[b][sup & ~big].sum() * n).reshape(1)])
            o = torch.cat([counts[b][big], counts[b][sup & ~big].sum().reshape(1)])
            keep = e > 0
            chi2 = (((o[keep] - e[keep]) ** 2) / e[keep]).sum().item()
            dof = int(keep.sum()) - 1
            assert chi2 <= dof + 5.0 * (2.0 * max(dof, 1)) ** 0.5 + 5.0, (b, chi2, dof)
    # finished rows emit pad, EOS finishes a row
    fin = torch.tensor([0, 1, 0], dtype=torch.uint8, device=backend)
    ops.sample(dl, T, top_k, top_p, True, 5, 3, fin, 77, out, eos_id=int(out[0]))
    assert int(out[1]) == 77


@pytest.mark.parametrize("M,N,K,act,f32", [(8, 2048, 2048, 0, 0), (8, 4096, 2048, 0, 0), (5, 512, 256, 1, 0), (8, 4112, 128, 0, 1),
                                            (8, 256, 384, 0, 0), (3, 64, 64, 0, 0),
                                            # K == waves x chunks x k-step exactly: the "fast" instantiations of the decode step
                                            (8, 64, 2048, 1, 0), (6, 80, 2048, 0, 1), (8, 64, 6144, 0, 0), (8, 256, 1024, 0, 0),
                                            (7, 48, 2048, 0, 0), (8, 8208, 512, 0, 1), (8, 12288, 2048, 1, 0)])
def test_dec_gemm2_packed_weights(backend, M, N, K, act, f32):
    """fragment-packed weight stream (bra_dec_pack_weights) == the row-major stream: same products, the K-reduction only
    visits the k-steps in a different wave order"""
    if backend.type == "cpu" and N * K > 9000000:
        pytest.skip("emulator: small shapes only")
    x, W = rnd(M, K, dev=backend), rnd(N, K, dev=backend, scale=0.1)
    nw = (1.0 + 0.1 * torch.randn(K)).to(BF).to(backend)
    ss = ops.row_sumsq(x, 256)
    Wp = ops.dec_pack_weights(W, act=bool(act), out_f32=bool(f32))
    assert Wp is not None and Wp.shape == W.shape
    # the packed copy is a permutation of the 16-byte chunks of W
    assert torch.equal(Wp.view(-1, 8).float().sum(1).sort().values.cpu(), W.reshape(-1, 8).float().sum(1).sort().values.cpu())
    y0, _ = ops.dec_gemm2(x, W, ss_in=ss, norm_w=nw, act=bool(act), out_f32=bool(f32))
    y1, _ = ops.dec_gemm2(x, Wp, ss_in=ss, norm_w=nw, act=bool(act), out_f32=bool(f32), packed=True)
    assert rel(y1, y0) < (1e-5 if f32 else 4e-3)
    if not act and not f32:
        r = rnd(M, N, dev=backend)
        yr0, s0 = ops.dec_gemm2(x, W, res=r, want_ss=True)
        yr1, s1 = ops.dec_gemm2(x, Wp, res=r, want_ss=True, packed=True)
        assert rel(yr1, yr0) < 4e-3 and rel(s1.sum(1), s0.sum(1)) < 1e-2
    assert ops.dec_pack_weights(rnd(40, 48, dev=backend)) is None          # not a tile multiple -> caller keeps row-major
    # norm weight folded into the packed copy, rstd applied to the reduced products: same result up to bf16 rounding placement
    Wf = ops.dec_pack_weights(W, act=bool(act), out_f32=bool(f32), norm_w=nw)
    y2, _ = ops.dec_gemm2(x, Wf, ss_in=ss, norm_w=nw, act=bool(act), out_f32=bool(f32), packed=3)
    xf = x.float()
    xn = xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + 1e-6) * nw.float()
    yr = xn @ W.float().T
    if act:
        Fh = N // 2
        yr = yr.view(M, Fh // 8, 2, 8)                                      # gate / up rows interleaved in blocks of 8
        yr = (torch.nn.functional.silu(yr[:, :, 0]) * yr[:, :, 1]).reshape(M, Fh)
    assert rel(y2, yr) < 1.5e-2 and rel(y2, y0) < 1.5e-2


@pytest.mark.parametrize("M,N,K,f32", [(8, 64, 9728, 0), (6, 64, 4096, 1), (8, 2560, 4096, 0), (8, 2560, 9728, 0), (12, 64, 4096, 1), (12, 64, 9728, 0),
                                       (16, 2560, 9728, 0)])
def test_dec_gemm2_packed_weights_beyond_one_register_round(backend, M, N, K, f32):
    """Qwen3-4B widths: o_proj (K = 4096) and down_proj (K = 9728) need more k-steps than one register round of the streaming
    projection holds (waves x chunks); the multi-round path must walk the fragment-packed image too (it used to address it as
    row-major: silently wrong products)"""
    if backend.type == "cpu" and N * K > 9000000:
        pytest.skip("emulator: small shapes only")
    x, W = rnd(M, K, dev=backend), rnd(N, K, dev=backend, scale=0.05)
    r = None if f32 else rnd(M, N, dev=backend)
    rows = 16 if M > 8 else 8
    Wp = ops.dec_pack_weights(W, out_f32=bool(f32), rows=rows)
    assert Wp is not None
    y0, _ = ops.dec_gemm2(x, W, res=r, out_f32=bool(f32))
    y1, s1 = ops.dec_gemm2(x, Wp, res=r, out_f32=bool(f32), packed=True, want_ss=not f32)
    ref = x.float() @ W.float().T
    if r is not None:
        ref = ref.to(BF).float() + r.float()
    assert rel(y1, ref) < 5e-3 and rel(y1, y0) < 5e-3
    if not f32:
        assert rel(s1[:M].sum(1), (y1.float() ** 2).sum(1)) < 1e-2


def test_sampler_two_eos_ids_and_forced_token(backend):
    """a row finishes on EITHER listed EOS id (HF accepts a list; Qwen3's generation_config has two); bra_force_token raises
    one logit at the scheduled step only"""
    B, V = 3, 5000
    logits = torch.zeros(B, V, device=backend)
    logits[0, 11] = logits[1, 22] = logits[2, 33] = 9.0
    out = torch.empty(B, dtype=torch.int32, device=backend)
    fin = torch.zeros(B, dtype=torch.uint8, device=backend)
    step = torch.zeros(1, dtype=torch.int32, device=backend)
    ops.sample(logits, 1.0, 0, 1.0, False, 0, step, fin, 0, out, eos_id=11, eos_id2=22)
    assert out.tolist() == [11, 22, 33] and fin.tolist() == [1, 1, 0]
    at = torch.tensor([5, 0, 7], dtype=torch.int32, device=backend)
    ops.force_token(logits, 44, step, at)             # step 0: only row 1
    ops.sample(logits, 1.0, 0, 1.0, False, 0, step, None, 0, out)
    assert out.tolist() == [11, 44, 33]


@pytest.mark.parametrize("V,k", [(151936, 20), (5003, 20), (9000, 1), (6000, 64), (300, 7), (200000, 20)])
def test_one_launch_sampler_equals_two_launches(debug_backend, V, k):
    """sample_tiles_one_kernel (round 6: the tile-maxima sampler as one 1024-thread workgroup per row — measured neutral, kept as an
    opt-in of the debug library) draws the tokens, log-probs, embedding rows, RMSNorm statistics and rotary rows of the two-launch
    form bit for bit: the k best tiles under a strict total order do not depend on how the maxima are dealt to the lists."""
    from bioreason_amd._lib import get_lib
    backend = debug_backend
    B, H, hd = 3, 256, 64
    g = torch.Generator().manual_seed(V * 3 + k)
    logits = torch.randn(B, V, generator=g)
    logits[1] = torch.round(logits[1] * 2) / 2
    logits[2, :] = -1.0
    logits[2, V - 1] = logits[2, 17] = logits[2, 16] = 3.0
    dl = logits.to(backend)
    tmax = ops.tile_max(dl)
    E = (torch.randn(V, H, generator=g) * 0.05).to(torch.bfloat16).to(backend)
    cosT = torch.randn(64, hd // 2, generator=g).to(backend)
    sinT = torch.randn(64, hd // 2, generator=g).to(backend)
    pos0 = torch.tensor([3, 9, 0], dtype=torch.int32, device=backend)
    do_sample = k > 1
    res = []
    try:
        for one in (0, 1):
            get_lib().call("bra_sample_set_one_launch", one)
            outs = []
            for s_ in range(2 if backend.type == "cpu" else 24):
                out = torch.empty(B, dtype=torch.int32, device=backend)
                lp = torch.empty(B, device=backend)
                x = torch.zeros(B, H, dtype=torch.bfloat16, device=backend)
                ss = torch.zeros(B, 4, device=backend)
                pos = torch.zeros(B, dtype=torch.int32, device=backend)
                rows = torch.zeros(B, hd, device=backend)
                ops.sample_tiles(dl, tmax, 0.7, k if do_sample else 0, 0.9, do_sample, 31, s_, None, 0, out, out_logp=lp,
                                 embed=(E, x, ss), advance=(pos0, pos, cosT, sinT, hd, rows))
                outs.append([t.cpu().clone() for t in (out, lp, x.float(), ss, pos, rows)])
            res.append(outs)
    finally:
        get_lib().call("bra_sample_set_one_launch", 0)
    for a, b in zip(res[0], res[1]):
        for ta, tb in zip(a, b):
            assert torch.equal(ta, tb)


@pytest.mark.parametrize("V,k", [(5000, 20), (5003, 20), (9000, 1), (6000, 64), (300, 7)])
def test_sampler_over_tile_maxima_equals_full_scan(backend, V, k):
    """bra_sample_tiles (top-k over the maxima of the logits' 16-column tiles, then over the 16 k logits of the k best tiles) draws
    exactly the tokens bra_sample draws from the same logits: same (value desc, index asc) order, same ties, ragged last tile, a
    row whose best values are all equal, fewer distinct tiles than k.  HF: TF:generation/logits_process.py:238,473,542."""
    B = 4
    g = torch.Generator().manual_seed(V + k)
    logits = torch.randn(B, V, generator=g)
    logits[1] = torch.round(logits[1] * 2) / 2                   # heavy ties across and inside tiles
    logits[2, :] = -1.0
    logits[2, V - 1] = logits[2, 17] = logits[2, 16] = 3.0      # ties at the ragged end and inside one tile
    logits[3, 5:5 + 40] += 8.0                                   # the k best packed into three neighbouring tiles
    dl = logits.to(backend)
    tmax = ops.tile_max(dl)
    want_tm = torch.nn.functional.pad(logits, (0, (-V) % 16), value=-3e38).view(B, -1, 16).amax(-1)
    assert torch.equal(tmax.cpu(), want_tm)
    do_sample = k > 1
    out0 = torch.empty(B, dtype=torch.int32, device=backend)
    out1 = torch.empty_like(out0)
    step = torch.zeros(1, dtype=torch.int32, device=backend)
    lp0 = torch.empty(B, device=backend)
    lp1 = torch.empty(B, device=backend)
    for s_ in range((2 if k > 20 else 4) if backend.type == "cpu" else 60):       # (emulator: a few draws exercise every path; the GPU run does the statistics)
        step.fill_(s_)
        if V >= 4096:
            ops.sample(dl, 0.7, k if do_sample else 0, 0.9, do_sample, 31, step, None, 0, out0, out_logp=lp0)
        else:                                       # (the one-workgroup-per-row kernel: the only full-scan form below 4096)
            ops.sample(dl, 0.7, k if do_sample else 0, 0.9, do_sample, 31, step, None, 0, out0, out_logp=lp0, ws=None)
        ops.sample_tiles(dl, tmax, 0.7, k if do_sample else 0, 0.9, do_sample, 31, step, None, 0, out1, out_logp=lp1)
        assert out0.tolist() == out1.tolist(), (s_, out0.tolist(), out1.tolist())
        assert torch.equal(lp0.cpu(), lp1.cpu())
        # the step index as a launch argument == the device-side counter
        out2 = torch.empty_like(out0)
        ops.sample_tiles(dl, tmax, 0.7, k if do_sample else 0, 0.9, do_sample, 31, s_, None, 0, out2)
        assert out2.tolist() == out1.tolist()
    if not do_sample:
        assert out1.tolist() == logits.argmax(-1).tolist()


@pytest.mark.gpu
def test_sampler_over_tile_maxima_at_qwen3_vocabulary(hip_device):
    """the bench's sampler call — B = 8 rows, V = 151 936 (9 496 tiles in 8 slices), T = 0.6 / top-k 20 / top-p 0.95 — through the lm_head
    epilogue's tile maxima against the full-scan sampler on the same logits: identical tokens and log-probs over 300 steps, also with
    a forced EOS logit (bra_force_token_tiles) and bf16-grid logits (many exact ties across tiles)"""
    dev = hip_device
    B, V, K = 8, 151936, 256
    x, W = rnd(B, K, dev=dev), rnd(V, K, dev=dev, scale=0.3)
    nw = (1.0 + 0.1 * torch.randn(K)).to(BF).to(dev)
    ss = ops.row_sumsq(x, 256)
    Wf = ops.dec_pack_weights(W, out_f32=True, norm_w=nw)
    tm = torch.empty(B, V // 16, device=dev)
    logits, _ = ops.dec_gemm2(x, Wf, ss_in=ss, norm_w=nw, out_f32=True, packed=3, tile_max=tm)
    assert torch.equal(tm, logits.view(B, V // 16, 16).amax(-1))
    lg2 = logits.to(BF).float()                     # values on the bf16 grid: exact ties are common
    for lg in (logits, lg2):
        tmx = ops.tile_max(lg)
        o0, o1 = torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)
        l0, l1 = torch.empty(B, device=dev), torch.empty(B, device=dev)
        step = torch.zeros(1, dtype=torch.int32, device=dev)
        at = torch.tensor([3, 50, 7, 299, 0, 120, 9, 11], dtype=torch.int32, device=dev)
        lg_a, lg_b = lg.clone(), lg.clone()
        for s_ in range(300):
            step.fill_(s_)
            ops.force_token(lg_a, 151645, step, at)
            ops.force_token_tiles(lg_b, 151645, s_, at, tmx)
            ops.sample(lg_a, 0.6, 20, 0.95, True, 42, step, None, 0, o0, out_logp=l0)
            ops.sample_tiles(lg_b, tmx, 0.6, 20, 0.95, True, 42, s_, None, 0, o1, out_logp=l1)
            assert torch.equal(o0, o1) and torch.equal(l0, l1), s_
        assert torch.equal(lg_a, lg_b) and torch.equal(tmx, ops.tile_max(lg_b))


def test_sampler_over_tile_maxima_tail_work(backend):
    """the drawing wave of bra_sample_tiles also gathers x = E[token] + its RMSNorm statistic (as bra_sample_embed) and leaves
    pos0 + step and the (cos | sin) row of that position (what bra_advance_counters left for the next decode step); finished rows
    emit pad, two EOS ids, bra_force_token_tiles keeps the tile maxima consistent"""
    B, V, H, hd = 5, 4800, 64, 128
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(B, V, generator=g).to(backend)
    tmax = ops.tile_max(logits)
    E = rnd(V, H, dev=backend)
    x = torch.zeros(B, H, dtype=BF, device=backend)
    ss = torch.full((8, 32), 7.0, device=backend)
    out, out_ref = torch.empty(B, dtype=torch.int32, device=backend), torch.empty(B, dtype=torch.int32, device=backend)
    step = torch.full((1,), 3, dtype=torch.int32, device=backend)
    cosT, sinT = torch.rand(64, hd // 2, device=backend), torch.rand(64, hd // 2, device=backend)
    pos0 = torch.tensor([4, 9, 0, 31, 2], dtype=torch.int32, device=backend)
    pos_out = torch.zeros_like(pos0)
    rows = torch.zeros(B, hd, device=backend)
    toks = torch.zeros(B, 8, dtype=torch.int32, device=backend)
    ops.sample(logits, 0.6, 20, 0.95, True, 77, step, None, 0, out_ref)
    ops.sample_tiles(logits, tmax, 0.6, 20, 0.95, True, 77, 3, None, 0, out, tokens_out=toks, embed=(E, x, ss),
                     advance=(pos0, pos_out, cosT, sinT, hd, rows))
    assert out.tolist() == out_ref.tolist()
    assert toks[:, 3].tolist() == out.tolist() and int(toks.abs().sum()) == int(out.abs().sum())
    assert torch.equal(x.cpu(), E[out.long()].cpu())
    assert rel(ss[:B, 0], (x.float() ** 2).sum(1)) < 1e-5 and float(ss[:B, 1:].abs().max()) == 0 and float(ss[B:].min()) == 7.0
    want_pos = pos0 + 3
    assert torch.equal(pos_out.cpu(), want_pos.cpu())
    assert torch.equal(rows.cpu(), torch.cat([cosT[want_pos.long()], sinT[want_pos.long()]], -1).cpu())
    # finished rows emit pad; either EOS id finishes a row
    lz = torch.zeros(3, V, device=backend)
    lz[0, 11] = lz[1, 22] = lz[2, 33] = 9.0
    tz = ops.tile_max(lz)
    o3 = torch.empty(3, dtype=torch.int32, device=backend)
    fin = torch.zeros(3, dtype=torch.uint8, device=backend)
    ops.sample_tiles(lz, tz, 1.0, 0, 1.0, False, 0, 0, fin, 5, o3, eos_id=11, eos_id2=22)
    assert o3.tolist() == [11, 22, 33] and fin.tolist() == [1, 1, 0]
    ops.sample_tiles(lz, tz, 1.0, 0, 1.0, False, 0, 1, fin, 5, o3, eos_id=11, eos_id2=22)
    assert o3.tolist() == [5, 5, 33]
    # forced token at the scheduled step only (device counter and launch argument forms)
    at = torch.tensor([5, 0, 7], dtype=torch.int32, device=backend)
    st0 = torch.zeros(1, dtype=torch.int32, device=backend)
    ops.force_token_tiles(lz, 4444, st0, at, tz)
    ops.sample_tiles(lz, tz, 1.0, 0, 1.0, False, 0, st0, None, 0, o3)
    assert o3.tolist() == [11, 4444, 33]
    ops.force_token_tiles(lz, 4321, 7, at, tz)
    ops.sample_tiles(lz, tz, 1.0, 0, 1.0, False, 0, 7, None, 0, o3)
    assert o3.tolist() == [11, 4444, 4321]
    assert torch.equal(tz.cpu(), ops.tile_max(lz).cpu())


@pytest.mark.parametrize("M,N,K", [(8, 8208, 512), (6, 4112, 256), (12, 4096, 256)])
def test_lm_head_epilogue_leaves_tile_maxima(backend, M, N, K):
    """the fp32-logits projection of the decode step (packed, norm folded: the lm_head) leaves the maximum of every 16-column tile
    next to the logits — exactly the maxima of the logits it wrote"""
    x, W = rnd(M, K, dev=backend), rnd(N, K, dev=backend, scale=0.1)
    nw = (1.0 + 0.1 * torch.randn(K)).to(BF).to(backend)
    ss = ops.row_sumsq(x, 256)
    Wf = ops.dec_pack_weights(W, out_f32=True, norm_w=nw, rows=16 if M > 8 else 8)
    tm = torch.full((M, N // 16), -7.0, device=backend)
    y, _ = ops.dec_gemm2(x, Wf, ss_in=ss, norm_w=nw, out_f32=True, packed=3, tile_max=tm)
    assert torch.equal(tm.cpu(), y.float().view(M, N // 16, 16).amax(-1).cpu())
    y0, _ = ops.dec_gemm2(x, Wf, ss_in=ss, norm_w=nw, out_f32=True, packed=3)
    assert torch.equal(y.cpu(), y0.cpu())


def test_process_dna_embeddings_public_method(backend):
    """DNALLMModel.process_dna_embeddings (dna_llm.py:103-179): per batch item, the first `attention_mask.sum()` projected
    rows of each of its sequences, concatenated — against the golden fixture's oracle weights"""
    import os as _os
    from test_model_parity import GOLD, build, to_dev
    from oracle import dna_llm_oracle as O
    fix = torch.load(_os.path.join(GOLD, "tiny_a.pt"), weights_only=False)
    m = build(fix, backend, False)
    b = to_dev(fix["batch"], backend)
    got = m.process_dna_embeddings(b["dna_tokenized"], b["batch_idx_map"], b["input_ids"].shape[0])
    cfg = fix["config"]
    dna = O.make_nt_v2(cfg["dna"], "eager")
    dna.load_state_dict({k: v.float() for k, v in fix["state"]["dna"].items() if "inv_freq" not in k}, strict=False)
    text = O.make_qwen3(cfg["text"], "eager")
    ora = O.OracleDNALLM(text, dna, cfg["dna_token_id"]).eval()
    ora.dna_projection.load_state_dict({k: v.float() for k, v in fix["state"]["proj"].items()})
    want = ora.process_dna_embeddings(fix["batch"]["dna_tokenized"], fix["batch"]["batch_idx_map"], b["input_ids"].shape[0])
    assert len(got) == len(want)
    for g_, w_ in zip(got, want):
        assert tuple(g_.shape) == tuple(w_.shape)
        assert rel(g_, w_.detach()) < 2e-2


@pytest.mark.gpu
def test_gemm_glds_at_bench_shape(hip_device):
    """the dominant kernel at the shapes the timed step launches it with: M = 8 x 2436 rows, N = 12288 (gate/up), K = 2048 with
    the LoRA pair K2 = 64 riding in the accumulators, and the nt = 96 K-step down projection (K = 6144); vs fp32 torch"""
    dev = hip_device
    for (M, N, K, K2) in [(19488, 12288, 2048, 64), (19488, 2048, 6144, 64), (2180, 4096, 2048, 128)]:
        a, b = rnd(M, K, dev=dev), rnd(N, K, dev=dev)
        a2, b2 = rnd(M, K2, dev=dev), rnd(N, K2, dev=dev)
        got = ops.gemm_nt(a, b, a2=a2, b2=b2)
        # spot-check 64 row blocks against fp32 (the full fp32 product is 0.9 GB x 3)
        g = torch.Generator().manual_seed(M + N)
        rows = torch.randint(0, M, (512,), generator=g).to(dev)
        rows[:4] = torch.tensor([0, 255, 256, M - 1], device=dev)
        ref = a[rows].float() @ b.float().T + a2[rows].float() @ b2.float().T
        err = rel(got[rows], ref)
        assert err < 4e-3, (M, N, K, K2, err)        # one bf16 output rounding


@pytest.mark.gpu
def test_lmhead_lse_at_full_vocab(hip_device):
    """fused lm_head + log-sum-exp + target gather at V = 151936 (1187 column tiles + the ragged tail), M = 8 x 256 rows"""
    dev = hip_device
    M, V, K = 2048, 151936, 2048
    h = rnd(M, K, dev=dev)
    E = rnd(V, K, dev=dev, scale=0.05)
    g = torch.Generator().manual_seed(1)
    tgt = torch.randint(0, V, (M,), generator=g).to(torch.int32).to(dev)
    tgt[:3] = torch.tensor([0, V - 1, V - 64], dtype=torch.int32, device=dev)
    logp, lse = ops.lmhead_logprob(h, E, tgt)
    ref_logits = h.float() @ E.float().T
    ref_lse = torch.logsumexp(ref_logits, -1)
    ref_lp = ref_logits.gather(1, tgt.long()[:, None]).squeeze(1) - ref_lse
    assert (lse - ref_lse).abs().max().item() < 2e-3 * ref_lse.abs().max().item() + 1e-3
    assert (logp - ref_lp).abs().max().item() < 3e-2
    assert rel(logp, ref_lp) < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("M,F,K,K2", [(2180, 6144, 2048, 64), (16384, 4096, 1024, 0), (2181, 9728, 2560, 64)])
def test_gemm_swiglu_at_model_shapes(hip_device, M, F, K, K2):
    """the fused gate/up + SwiGLU launch at the shapes it runs in the step (one prompt with the LoRA rank part, the NT-v2 FFN, Qwen3-4B
    widths with a ragged row count): bit-identical to bra_gemm_bf16_nt + bra_swiglu_fwd"""
    dev = hip_device
    a, w = rnd(M, K, dev=dev, seed=31), rnd(2 * F, K, dev=dev, seed=32, scale=0.05)
    a2 = rnd(M, K2, dev=dev, seed=33) if K2 else None
    b2 = rnd(2 * F, K2, dev=dev, seed=34, scale=0.05) if K2 else None
    got = ops.gemm_swiglu(a, w, a2=a2, b2=b2)
    assert got is not None and torch.equal(got, ops.swiglu_fwd(ops.gemm_nt(a, w, a2=a2, b2=b2)))


@pytest.mark.parametrize("M,F,K,K2", [(300, 128, 128, 0), (256, 256, 64, 64), (530, 384, 192, 128), (17, 128, 64, 0)])
def test_gemm_swiglu_equals_gemm_then_swiglu(backend, M, F, K, K2):
    """bra_gemm_swiglu_bf16_nt (round 6: SwiGLU in the epilogue of the ring kernel, gate / up rows of the same features interleaved on
    the DMA source side) against bra_gemm_bf16_nt + bra_swiglu_fwd on the same operands: the same roundings, bit for bit — ragged row
    counts (rows past M are computed and never stored), with and without the LoRA rank part.  TF:qwen3:81-83."""
    a, w = rnd(M, K, dev=backend, seed=21), rnd(2 * F, K, dev=backend, seed=22)
    a2 = rnd(M, K2, dev=backend, seed=23) if K2 else None
    b2 = rnd(2 * F, K2, dev=backend, seed=24) if K2 else None
    want = ops.swiglu_fwd(ops.gemm_nt(a, w, a2=a2, b2=b2))
    got = ops.gemm_swiglu(a, w, a2=a2, b2=b2)
    assert got is not None
    assert torch.equal(got.cpu(), want.cpu()), float((got.float() - want.float()).abs().max())
    assert ops.gemm_swiglu(a, rnd(2 * 96, K, dev=backend, seed=25)) is None          # F % 128 != 0: the caller takes the two launches


@pytest.mark.parametrize("variant", [0, 5, 7, 9, 10, 11, 12, 13, 14])
def test_gemm_bf16_epilogue_interior_and_edge_waves(debug_backend, variant):
    backend = debug_backend
    """k_gemm.hip epi_bf16_interior: a wave whose fragments all lie inside the matrix takes the batched epilogue (one base
    pointer per operand, every residual word requested up front), the others the generic one: every bias / residual combination
    on a matrix made of interior tiles only, on one with ragged edges, and with a residual whose row pitch is not a multiple of
    four elements (generic path everywhere).  The two paths must agree bit for bit on the rows they share."""
    from bioreason_amd._lib import get_lib
    get_lib().call("bra_gemm_set_variant", variant)
    try:
        K = 128
        for (M, N) in [(512, 256), (530, 300)]:
            a, b = rnd(M, K, dev=backend, seed=11), rnd(N, K, dev=backend, seed=12)
            bias, res = rnd(N, dev=backend, seed=13), rnd(M, N, dev=backend, seed=14)
            acc = a.float() @ b.float().T
            for use_bias in (False, True):
                for use_res in (False, True):
                    c = ops.gemm_nt(a, b, bias=bias if use_bias else None, res=res if use_res else None, alpha=0.25)
                    want = 0.25 * acc + (bias.float() if use_bias else 0)
                    want = want.to(BF).float() + res.float() if use_res else want
                    assert rel(c, want.to(BF)) < 4e-3, (M, N, use_bias, use_res)
                    if use_res:
                        wide = torch.zeros(M, N + 2, dtype=BF, device=backend)       # row pitch N + 2: not 8-byte aligned rows
                        wide[:, :N] = res
                        c2 = ops.gemm_nt(a, b, bias=bias if use_bias else None, res=wide[:, :N], alpha=0.25)
                        assert torch.equal(c, c2), (M, N, use_bias)
    finally:
        get_lib().call("bra_gemm_set_variant", -1)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14])
def test_gemm_tile_variants(debug_backend, variant):
    backend = debug_backend
    """every tile variant (128/256-row tiles x register prefetch depth 1/2; 5 / 9 / 10: the LDS-DMA kernel at 256 / 192 / 128-row tiles;
    11 - 14: four waves with 80 x 128 / 64 x 128 / 80 x 64 / 64 x 64 per-wave tiles (opt-in);
    6 / 7: the 256 x 256 ring) against the fp32 statement"""
    from bioreason_amd._lib import get_lib
    get_lib().call("bra_gemm_set_variant", variant)
    try:
        for (M, N, K, K2) in [(300, 200, 256, 64), (520, 136, 64, 0), (100, 128, 192, 0)]:
            a, b = rnd(M, K, dev=backend), rnd(N, K, dev=backend)
            a2 = rnd(M, K2, dev=backend) if K2 else None
            b2 = rnd(N, K2, dev=backend) if K2 else None
            ref = a.float() @ b.float().T + (a2.float() @ b2.float().T if K2 else 0)
            c = ops.gemm_nt(a, b, a2=a2, b2=b2, out_f32=True)
            assert rel(c, ref) < 1e-5
            sk = torch.zeros(M, N, device=backend)
            ops.gemm_nt_splitk(a, b, sk, split_k=3)
            assert rel(sk, a.float() @ b.float().T) < 1e-5
            h, e = rnd(M, K, dev=backend, scale=0.3), rnd(N, K, dev=backend, scale=0.3)
            tgt = (torch.arange(M) % N).to(torch.int32).to(backend)
            logp, lse = ops.lmhead_logprob(h, e, tgt)
            lg = (h.float() @ e.float().T).to(BF).float()
            assert (lse.cpu() - torch.logsumexp(lg, -1).cpu()).abs().max() < 2e-3
        if variant >= 6:
            # the ring kernel's own corner cases: tiles straddling M and N, a single K-tile (prologue == whole loop), K-tiles
            # from both operand pairs, every epilogue (bf16 + bias + residual, dlogits)
            for (M, N, K, K2) in [(600, 300, 320, 64), (257, 260, 64, 0), (256, 512, 128, 128)]:
                a, b = rnd(M, K, dev=backend), rnd(N, K, dev=backend)
                a2 = rnd(M, K2, dev=backend) if K2 else None
                b2 = rnd(N, K2, dev=backend) if K2 else None
                bias, res = rnd(N, dev=backend), rnd(M, N, dev=backend)
                ref = a.float() @ b.float().T + (a2.float() @ b2.float().T if K2 else 0)
                c = ops.gemm_nt(a, b, a2=a2, b2=b2, bias=bias, res=res, alpha=0.5)
                want = ((0.5 * ref + bias.float()).to(BF).float() + res.float())
                assert rel(c, want) < 4e-3
            M, N, K = 300, 520, 128
            h, e = rnd(M, K, dev=backend, scale=0.3), rnd(N, K, dev=backend, scale=0.3)
            tgt = (torch.arange(M) * 7 % N).to(torch.int32).to(backend)
            logp, lse = ops.lmhead_logprob(h, e, tgt)
            lg = (h.float() @ e.float().T).to(BF).float()
            assert (lse.cpu() - torch.logsumexp(lg, -1).cpu()).abs().max() < 2e-3
            assert (logp.cpu() - (lg.gather(1, tgt.long()[:, None]).squeeze(1) - torch.logsumexp(lg, -1)).cpu()).abs().max() < 2e-3
            coef = torch.randn(M).to(backend)
            dl = ops.lmhead_dlogits(h, e, tgt, lse, coef)
            p_ = torch.softmax(lg, -1)
            onehot = torch.zeros(M, N, device=lg.device)
            onehot[torch.arange(M), tgt.long().to(lg.device)] = 1
            assert rel(dl, coef.float().to(lg.device)[:, None] * (onehot - p_)) < 1e-2
    finally:
        get_lib().call("bra_gemm_set_variant", -1)


# ----------------------------------------------------------------------------- decode-time streaming projections
@pytest.mark.parametrize("M,N,K", [(8, 64, 64), (5, 96, 256), (8, 2048, 2048), (8, 2048, 6144), (8, 4096, 2048), (3, 4104, 96), (1, 40, 128),
                                   (8, 9616, 64), (4, 48, 1664), (2, 16, 3328)])
@pytest.mark.parametrize("norm", [False, True])
def test_dec_gemm2(backend, M, N, K, norm):
    """bra_dec_gemm2 against fp32 torch with the reference's rounding points (TF:qwen3:59-64 RMSNorm in fp32 -> bf16 ->
    times weight; residual added to the bf16-rounded projection TF:qwen3:309,315; SwiGLU TF:qwen3:81-83)."""
    if backend.type == "cpu" and N * K > 3e6:
        pytest.skip("emulator: large shape covered on the GPU")
    x, W = rnd(M, K, dev=backend), rnd(N, K, dev=backend, scale=K ** -0.5)
    res, nw = rnd(M, N, dev=backend), (1 + 0.1 * rnd(K, dev=backend).float()).to(BF)
    eps = 1e-6
    xf = x.float()
    if norm:
        ss = ops.row_sumsq(x, 32)
        assert rel(ss[:M, 0], (xf * xf).sum(1)) < 1e-5 and float(ss[:, 1:].abs().max()) == 0
        xn = (nw.float() * (xf * torch.rsqrt((xf * xf).mean(1, keepdim=True) + eps)).to(BF).float()).to(BF).float()
    else:
        ss, xn = None, xf
    ref = xn @ W.float().T
    kw = dict(ss_in=ss, norm_w=nw if norm else None, eps=eps)
    y, ss_out = ops.dec_gemm2(x, W, res=res, want_ss=True, **kw)
    want = (ref.to(BF).float() + res.float()).to(BF)
    assert rel(y, want) < 4e-3
    # the epilogue's partial sums of squares describe exactly the bf16 rows it stored
    assert rel(ss_out[:M].sum(1), (y.float() ** 2).sum(1)) < 1e-5
    y32, _ = ops.dec_gemm2(x, W, out_f32=True, **kw)
    assert rel(y32, ref) < (2e-5 if not norm else 1e-4)
    if N % 16 == 0:
        # rows interleaved [8 gate | 8 up] per 16-row block, as rollout_weights lays out gate_proj / up_proj
        a, _ = ops.dec_gemm2(x, W, act=True, **kw)
        r3 = ref.to(BF).float().view(M, N // 16, 2, 8)
        g, u = r3[:, :, 0].reshape(M, -1), r3[:, :, 1].reshape(M, -1)
        assert rel(a, (torch.nn.functional.silu(g).to(BF).float() * u).to(BF)) < 6e-3


# ----------------------------------------------------------------------------- low-rank weight gradients
@pytest.mark.parametrize("M,N,R,chunk", [(100, 136, 32, 0), (300, 128, 64, 64), (257, 264, 128, 96), (33, 8, 64, 0)])
def test_wgrad_tn(backend, M, N, R, chunk):
    """bra_wgrad_tn against fp32 torch: out += alpha * y^T t, both output orientations, accumulating"""
    y, t = rnd(M, N, dev=backend), rnd(M, R, dev=backend)
    ref = y.float().T @ t.float()
    out = torch.ones(N, R, device=backend)
    ops.wgrad_tn(y, t, out, alpha=0.5, m_chunk=chunk)
    assert rel(out, 0.5 * ref + 1) < 1e-5
    outT = torch.zeros(R, N, device=backend)
    ops.wgrad_tn(y, t, outT, transposed_out=True, m_chunk=chunk)
    assert rel(outT, ref.T) < 1e-5
    # strided views (a column block of a wider activation, as the fused projections hand over)
    big = rnd(M, N + 16, dev=backend)
    out2 = torch.zeros(N, R, device=backend)
    ops.wgrad_tn(big[:, 8:8 + N], t, out2)
    assert rel(out2, big[:, 8:8 + N].float().T @ t.float()) < 1e-5


# ----------------------------------------------------------------------------- LoRA branch under dropout
@pytest.mark.parametrize("M,K,R", [(70, 64, 32), (130, 200, 64), (65, 136, 128)])
def test_lora_dropout_kernels(backend, M, K, R):
    """the three places the masked operand is needed regenerate the same masks: forward t = s * drop_j(x) A^T, input
    gradient sum_j drop_j'(dts_j A_j), weight gradient dA += dts^T drop_j(x) — against torch with the exported masks"""
    p, seeds = 0.25, [11, 22, 33, 44]
    nb = R // 32
    x, A, dts = rnd(M, K, dev=backend), rnd(R, K, dev=backend, scale=K ** -0.5), rnd(M, R, dev=backend)
    masks = [ops.dropout_mask(M, K, p, seeds[j], backend).float().cpu() for j in range(nb)]
    frac = torch.stack(masks).mean().item()
    assert abs(frac - (1 - p)) < 0.03 and not torch.equal(masks[0], masks[-1]) or nb == 1
    assert torch.equal(masks[0], ops.dropout_mask(M, K, p, seeds[0], backend).float().cpu())
    xd = [(x.float().cpu() * mk / (1 - p)).to(BF).float() for mk in masks]           # torch: scale in fp32, round once
    t = ops.lora_down_drop(x, A, 0.5, p, seeds)
    want_t = torch.cat([0.5 * xd[j] @ A.float().cpu()[32 * j:32 * j + 32].T for j in range(nb)], dim=1)
    assert rel(t, want_t.to(BF)) < 4e-3
    up = ops.lora_up_drop(dts, A.T.contiguous(), p, seeds)
    want_up = sum((dts.float().cpu()[:, 32 * j:32 * j + 32] @ A.float().cpu()[32 * j:32 * j + 32]) * masks[j] / (1 - p) for j in range(nb))
    assert rel(up, want_up.to(BF)) < 4e-3
    dA = torch.zeros(R, K, device=backend)
    ops.wgrad_tn(x, dts, dA, transposed_out=True, drop=(p, seeds))
    want_dA = torch.cat([dts.float().cpu()[:, 32 * j:32 * j + 32].T @ xd[j] for j in range(nb)], dim=0)
    assert rel(dA, want_dA) < 1e-5
    # p = 0 reproduces the plain kernels
    t0 = ops.lora_down_drop(x, A, 0.5, 0.0, seeds)
    assert rel(t0, (0.5 * x.float() @ A.float().T).to(BF)) < 4e-3


@pytest.mark.parametrize("M,K,R,nlive", [(100, 768, 64, 2), (70, 1024, 128, 3), (33, 896, 32, 1), (40, 6144, 64, 1)])
def test_lora_down_drop_split_k(backend, M, K, R, nlive):
    """small M: `ksplit` workgroups per 32-row block take K / ksplit each, a second launch sums the fp32 partial tiles in a fixed order
    (bra_lora_down_drop_splitk) — same masks (the hash is of the element index), same result up to the fp32 summation order,
    deterministic from call to call"""
    from bioreason_amd._lib import get_lib
    p, seeds = 0.25, [11, 22, 33, 44][:nlive]
    # (big M: no split for K = 2048 — the chip is full; four slices for the long chain of K = 6144, round 6)
    assert get_lib()._dll.bra_lora_down_splitk_plan(M, K) >= 2 and get_lib()._dll.bra_lora_down_splitk_plan(19488, K) == (4 if K >= 4096 else 1)
    x, A = rnd(M, K, dev=backend), rnd(R, K, dev=backend, scale=K ** -0.5)
    A[32 * nlive:] = 0
    masks = [ops.dropout_mask(M, K, p, seeds[j], backend).float().cpu() for j in range(nlive)]
    xd = [(x.float().cpu() * mk / (1 - p)).to(BF).float() for mk in masks]
    want = torch.zeros(M, R)
    for j in range(nlive):
        want[:, 32 * j:32 * j + 32] = 0.5 * xd[j] @ A.float().cpu()[32 * j:32 * j + 32].T
    old = ops.LORA_DOWN_SPLITK
    try:
        ops.LORA_DOWN_SPLITK = True
        t1 = ops.lora_down_drop(x, A, 0.5, p, seeds)
        t2 = ops.lora_down_drop(x, A, 0.5, p, seeds)
        ops.LORA_DOWN_SPLITK = False
        t0 = ops.lora_down_drop(x, A, 0.5, p, seeds)
    finally:
        ops.LORA_DOWN_SPLITK = old
    assert torch.equal(t1.cpu(), t2.cpu())
    assert rel(t1, want.to(BF)) < 4e-3 and rel(t1, t0) < 4e-3 and (t1[:, 32 * nlive:] == 0).all()


# ----------------------------------------------------------------------------- one-launch shared-prefix decode attention
def _rope_rows_ref(x, cos, sin):
    """rotate-half RoPE (TF:qwen3:109-133) of x [..., hd] with cos / sin [..., hd/2] rows"""
    h = x.shape[-1] // 2
    x1, x2 = x[..., :h], x[..., h:]
    return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], -1)


@pytest.mark.parametrize("hd,Hq,Hkv,R,copies,P,C,t,use_rows,use_mask", [
    (128, 4, 2, 1, 8, 200, 256, 0, True, True),        # first decode step: no completion keys yet
    (128, 4, 2, 1, 8, 200, 256, 1, True, True),
    (128, 4, 2, 2, 3, 150, 192, 64, False, True),      # two prompts, completion exactly one full chunk
    (128, 2, 2, 1, 5, 70, 192, 130, True, True),       # G = 1, three completion chunks (the last ragged)
    (64, 8, 2, 2, 4, 129, 128, 77, False, True),       # hd 64, G = 4
    (128, 8, 2, 3, 2, 333, 128, 65, True, False),      # no padding mask passed; 3 prompts x 2 copies, G = 4
    (128, 16, 8, 1, 8, 2180, 256, 200, True, True),    # the cfg-3 geometry: 35 prompt chunks x 8 kv-heads, 4 completion chunks
])
def test_dec_attn_items_and_merge(backend, hd, Hq, Hkv, R, copies, P, C, t, use_rows, use_mask):
    """bra_dec_attn_one (items kernel + merge kernel): q/k RMSNorm + RoPE, cache append, attention over the shared prompt K / V^T + each sequence's own
    completion keys + the new key, merged in the same launch — against plain fp32 torch (TF:qwen3:231-284 on one token)."""
    from bioreason_amd._lib import get_lib, current_stream
    dev = backend
    B, G = R * copies, Hq // Hkv
    Nq, Nkv = Hq * hd, Hkv * hd
    eps, scale = 1e-6, hd ** -0.5
    qkv = rnd(B, Nq + 2 * Nkv, dev=dev, seed=1)
    qw, kw = (1.0 + 0.1 * rnd(hd, dev=dev, seed=2).float()).to(BF), (1.0 + 0.1 * rnd(hd, dev=dev, seed=3).float()).to(BF)
    kp = rnd(R, Hkv, P, hd, dev=dev, seed=4)
    vp = rnd(R, Hkv, P, hd, dev=dev, seed=5)
    pitch = (P + 63) // 64 * 64
    vtp = torch.zeros(R, Hkv, hd, pitch, dtype=BF, device=dev)
    vtp[..., :P] = vp.transpose(2, 3)
    pmask = torch.ones(R, P, dtype=torch.uint8, device=dev)
    pmask[0, :13] = 0                                            # left padding of prompt 0
    kc = torch.zeros(B, Hkv, C, hd, dtype=BF, device=dev)
    cp = (C + 63) // 64 * 64
    vct = torch.zeros(B, Hkv, hd, cp, dtype=BF, device=dev)
    kc_old, vc_old = rnd(B, Hkv, max(t, 1), hd, dev=dev, seed=6), rnd(B, Hkv, max(t, 1), hd, dev=dev, seed=7)
    if t > 0:
        kc[:, :, :t] = kc_old[:, :, :t]
        vct[:, :, :, :t] = vc_old[:, :, :t].transpose(2, 3)
    npos = P + C + 1
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    ang = torch.arange(npos).float()[:, None] * inv[None]
    cosT, sinT = ang.cos().to(BF).float().to(dev).contiguous(), ang.sin().to(BF).float().to(dev).contiguous()
    pos = (torch.arange(B, dtype=torch.int32) % 5 + P - 13 + t).to(dev)
    rope_rows = torch.cat([cosT[pos.long()], sinT[pos.long()]], -1).contiguous() if use_rows else None
    nslot = (P + 63) // 64 + (C + 63) // 64 + 1
    part_o = torch.full((B * Hq, nslot, hd), float("nan"), dtype=torch.float32, device=dev)
    part_ml = torch.full((B * Hq, nslot, 2), float("nan"), dtype=torch.float32, device=dev)
    o = torch.zeros(B, Nq, dtype=BF, device=dev)
    if not use_mask:
        pmask[:] = 1
    get_lib().call("bra_dec_attn_one", qkv, Nq + 2 * Nkv, qw, kw, cosT, sinT, pos, rope_rows, kp, Hkv * P * hd, P * hd, hd, vtp,
                   Hkv * hd * pitch, hd * pitch, pitch, pmask if use_mask else None, kc, vct, cp, part_o, part_ml, nslot, o, Nq, R, copies,
                   Hq, Hkv, hd, P, C, t, eps, scale, None, current_stream(qkv))
    # ---- reference
    f = qkv.float().cpu()
    q = f[:, :Nq].view(B, Hq, hd)
    k = f[:, Nq:Nq + Nkv].view(B, Hkv, hd)
    v = f[:, Nq + Nkv:].view(B, Hkv, hd)

    def nrm(x, w):
        y = (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)).to(BF).float()
        return (w.float().cpu() * y).to(BF).float()
    c_, s_ = cosT.cpu()[pos.long().cpu()][:, None], sinT.cpu()[pos.long().cpu()][:, None]
    qn = _rope_rows_ref(nrm(q, qw), c_, s_).to(BF).float()
    kn = _rope_rows_ref(nrm(k, kw), c_, s_).to(BF).float()
    # caches after the append
    assert torch.equal(kc[:, :, t].float().cpu(), kn)
    assert torch.equal(vct[:, :, :, t].float().cpu(), v)
    if t > 0:
        assert torch.equal(kc[:, :, :t].cpu(), kc_old[:, :, :t].cpu())
    ref = torch.zeros(B, Hq, hd)
    for b in range(B):
        r = b // copies
        for hq in range(Hq):
            h = hq // G
            keys = torch.cat([kp[r, h].float().cpu(), kc_old[b, h, :t].float().cpu() if t > 0 else torch.zeros(0, hd), kn[b, h][None]], 0)
            vals = torch.cat([vp[r, h].float().cpu(), vc_old[b, h, :t].float().cpu() if t > 0 else torch.zeros(0, hd), v[b, h][None]], 0)
            valid = torch.cat([pmask[r].bool().cpu(), torch.ones(t + 1, dtype=torch.bool)])
            sc = (keys @ qn[b, hq]) * scale
            sc[~valid] = float("-inf")
            ref[b, hq] = torch.softmax(sc, 0) @ vals
    assert rel(o.view(B, Hq, hd), ref) < 6e-3


@pytest.mark.parametrize("r,n_targets", [(64, 1), (16, 3), (32, 3), (8, 2), (40, 3)])
def test_lora_branch_without_dropout_at_any_rank(backend, r, n_targets):
    """the no-dropout LoRA branch (engine._lora_fwd / _lora_bwd: x A^T, dy B through the row-block kernel below 8192 rows)
    must be right for EVERY adapter rank, not just r = 32 — `lora_r` / `lora_rank` are user options of the reference
    (train_dna_qwen.py:155-167, reason.py:376-388).  Compared with the plain GEMM statement of y = x W^T + s (x A^T) B^T."""
    from bioreason_amd.engine import QwenEngine
    M, K, Nj = 70, 96, 64
    r_pad = (n_targets * r + 63) // 64 * 64
    x, W = rnd(M, K, dev=backend), rnd(n_targets * Nj, K, dev=backend, scale=K ** -0.5)
    A = torch.zeros(r_pad, K, dtype=BF, device=backend)
    B_ = torch.zeros(n_targets * Nj, r_pad, dtype=BF, device=backend)
    for j in range(n_targets):
        A[j * r:(j + 1) * r] = rnd(r, K, dev=backend, scale=K ** -0.5, seed=j + 1)
        B_[j * Nj:(j + 1) * Nj, j * r:(j + 1) * r] = rnd(Nj, r, dev=backend, scale=0.3, seed=j + 9)

    class G:                                            # the fields of engine.LoraGroup the branch reads
        pass
    G.A, G.AT, G.B, G.BT = A, A.T.contiguous(), B_, B_.T.contiguous()
    G.scaling, G.n_sizes, G.r = 2.0, [Nj] * n_targets, r
    G.A_grad = torch.zeros(r_pad, K, device=backend)
    G.B_grad = torch.zeros(n_targets * Nj, r_pad, device=backend)
    y, t = QwenEngine._lora_fwd(x, W, G, True)
    t_ref = 2.0 * (x.float() @ A.float().T)
    assert rel(t, t_ref) < 4e-3, "x A^T: rank columns past the first 32 of a target (or past target count x 32) must be live"
    y_ref = x.float() @ W.float().T + t_ref.to(BF).float() @ B_.float().T
    assert rel(y, y_ref) < 6e-3
    dy = rnd(M, n_targets * Nj, dev=backend, seed=77)
    dx = QwenEngine._lora_bwd(dy, W.T.contiguous(), G, True, x, t)
    dts_ref = 2.0 * (dy.float() @ B_.float())
    dx_ref = dy.float() @ W.float() + dts_ref.to(BF).float() @ A.float()
    assert rel(dx, dx_ref) < 6e-3
    assert rel(G.A_grad, dts_ref.to(BF).float().T @ x.float()) < 6e-3
    assert rel(G.B_grad, dy.float().T @ t.float()) < 6e-3


@pytest.mark.parametrize("R,nlive", [(64, 1), (128, 3)])
def test_lora_dropout_padded_rank_blocks(backend, R, nlive):
    """a fused projection pads its rank to 64 / 128 columns (o, down: 1 target in 64; q/k/v: 3 in 128): the kernels are told
    how many rank blocks are live (= number of seeds) and must treat the padding exactly as the zero rows / columns it is"""
    M, K, p = 97, 136, 0.25
    seeds = [5, 6, 7, 8][:nlive]
    x, A, dts = rnd(M, K, dev=backend), rnd(R, K, dev=backend, scale=K ** -0.5), rnd(M, R, dev=backend)
    A[32 * nlive:] = 0
    dts[:, 32 * nlive:] = 0
    masks = [ops.dropout_mask(M, K, p, seeds[j], backend).float().cpu() for j in range(nlive)]
    xd = [(x.float().cpu() * mk / (1 - p)).to(BF).float() for mk in masks]
    t = ops.lora_down_drop(x, A, 0.5, p, seeds)
    want_t = torch.zeros(M, R)
    for j in range(nlive):
        want_t[:, 32 * j:32 * j + 32] = 0.5 * xd[j] @ A.float().cpu()[32 * j:32 * j + 32].T
    assert rel(t, want_t.to(BF)) < 4e-3 and (t[:, 32 * nlive:] == 0).all()
    up = ops.lora_up_drop(dts, A.T.contiguous(), p, seeds)
    want_up = sum((dts.float().cpu()[:, 32 * j:32 * j + 32] @ A.float().cpu()[32 * j:32 * j + 32]) * masks[j] / (1 - p) for j in range(nlive))
    assert rel(up, want_up.to(BF)) < 4e-3
    dA = torch.zeros(R, K, device=backend)
    ops.wgrad_tn(x, dts, dA, transposed_out=True, drop=(p, seeds))
    want_dA = torch.zeros(R, K)
    for j in range(nlive):
        want_dA[32 * j:32 * j + 32] = dts.float().cpu()[:, 32 * j:32 * j + 32].T @ xd[j]
    assert rel(dA, want_dA) < 1e-5 and (dA[32 * nlive:] == 0).all()


@pytest.mark.gpu
def test_gemm_ring_row_split_is_exact(hip_debug_device):
    hip_device = hip_debug_device
    """k_gemm.hip ring_split_rows: at N = 2048 the 19488-row projections are 616 ring tiles = 2.41 rounds, so the rows are split
    between the ring kernel (whole rounds) and the 256 x 128 kernel (the rest).  Same K order per output element in both
    kernels: the result must be bit-identical to the unsplit launch, residual included."""
    from bioreason_amd._lib import get_lib
    M, N, K, K2 = 8 * 2436, 2048, 2048, 64
    a, b = rnd(M, K, dev=hip_device, seed=1), rnd(N, K, dev=hip_device, seed=2)
    a2, b2 = rnd(M, K2, dev=hip_device, seed=3), rnd(N, K2, dev=hip_device, seed=4)
    res = rnd(M, N, dev=hip_device, seed=5)
    try:
        get_lib().call("bra_gemm_set_row_split", 0)
        c0 = ops.gemm_nt(a, b, a2=a2, b2=b2, res=res)
        get_lib().call("bra_gemm_set_row_split", 1)
        c1 = ops.gemm_nt(a, b, a2=a2, b2=b2, res=res)
    finally:
        get_lib().call("bra_gemm_set_row_split", 1)
    assert torch.equal(c0, c1)
    ref = (a[-300:].float() @ b.float().T + a2[-300:].float() @ b2.float().T + res[-300:].float())
    assert rel(c1[-300:], ref) < 4e-3


@pytest.mark.parametrize("M,N,K", [(16, 256, 256), (12, 96, 512), (9, 2048, 2048), (16, 2048, 6144), (16, 4096, 2048)])
def test_dec_gemm2_wide_rows(backend, M, N, K):
    """9 .. 16 batch rows (two prompts x 8 rollouts per GPU): 16-column tiles for every projection, weights packed for them
    (dec_pack_weights(rows=16)), RMSNorm folded into the packed copy with rstd applied in the epilogue, 16-row statistics"""
    if backend.type == "cpu" and N * K > 3e6:
        pytest.skip("emulator: large shape covered on the GPU")
    x, W = rnd(M, K, dev=backend), rnd(N, K, dev=backend, scale=K ** -0.5)
    res, nw = rnd(M, N, dev=backend), (1 + 0.1 * rnd(K, dev=backend).float()).to(BF)
    eps = 1e-6
    xf = x.float()
    ss = ops.row_sumsq(x, 32)
    assert ss.shape[0] == 16 and rel(ss[:M, 0], (xf * xf).sum(1)) < 1e-5
    xn = xf * torch.rsqrt((xf * xf).mean(1, keepdim=True) + eps) * nw.float()
    # plain projection + residual + output statistics (o_proj / down_proj)
    Wp = ops.dec_pack_weights(W, rows=16)
    y, ss_out = ops.dec_gemm2(x, Wp, res=res, want_ss=True, packed=True)
    want = ((xf @ W.float().T).to(BF).float() + res.float()).to(BF)
    assert rel(y, want) < 4e-3
    assert ss_out.shape[0] == 16 and rel(ss_out[:M].sum(1), (y.float() ** 2).sum(1)) < 1e-5
    # the same against row-major weights (no packing)
    y_rm, _ = ops.dec_gemm2(x, W, res=res)
    assert rel(y_rm, want) < 4e-3
    # folded norm (qkv): y = rstd * (x (W . nw)^T)
    Wf = ops.dec_pack_weights(W, norm_w=nw, rows=16)
    yq, _ = ops.dec_gemm2(x, Wf, ss_in=ss, norm_w=nw, packed=3)
    assert rel(yq, xn @ W.float().T) < 1.5e-2
    # fp32 logits (lm_head)
    Wh = ops.dec_pack_weights(W, out_f32=True, norm_w=nw, rows=16)
    y32, _ = ops.dec_gemm2(x, Wh, ss_in=ss, norm_w=nw, out_f32=True, packed=3)
    assert rel(y32, xn @ W.float().T) < 1.5e-2
    # gate / up + SwiGLU
    if N % 16 == 0:
        Wa = ops.dec_pack_weights(W, act=True, norm_w=nw, rows=16)
        a, _ = ops.dec_gemm2(x, Wa, ss_in=ss, norm_w=nw, act=True, packed=3)
        r3 = (xn @ W.float().T).view(M, N // 16, 2, 8)
        g, u = r3[:, :, 0].reshape(M, -1), r3[:, :, 1].reshape(M, -1)
        assert rel(a, torch.nn.functional.silu(g) * u) < 2e-2
    # unfolded statistics are not available above 8 rows
    with pytest.raises(RuntimeError):
        ops.dec_gemm2(x, W, ss_in=ss, norm_w=nw)


def test_sampler_fused_embed_with_twelve_rows(backend):
    """the sampler's drawing wave gathers x = E[token] and the row statistic for every sequence of a 9 .. 16-row decode
    (16-row statistics array, as bra_dec_gemm2's 16-row form folds it)"""
    B, V, H = 12, 4608, 64
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(B, V, generator=g).to(backend)
    E = rnd(V, H, dev=backend)
    x = torch.zeros(B, H, dtype=BF, device=backend)
    ss = torch.full((16, 32), 7.0, device=backend)
    step = torch.zeros(1, dtype=torch.int32, device=backend)
    out, out2 = torch.empty(B, dtype=torch.int32, device=backend), torch.empty(B, dtype=torch.int32, device=backend)
    ops.sample(logits, 1.0, 0, 1.0, False, 0, step, None, 0, out)
    assert out.tolist() == logits.argmax(-1).tolist()
    ops.sample(logits, 1.0, 0, 1.0, False, 0, step, None, 0, out2, embed=(E, x, ss))
    assert out2.tolist() == out.tolist()
    assert torch.equal(x.cpu(), E[out2.long()].cpu())
    assert rel(ss[:B, 0], (x.float() ** 2).sum(1)) < 1e-5 and float(ss[:B, 1:].abs().max()) == 0 and float(ss[B:].min()) == 7.0


@pytest.mark.gpu
@pytest.mark.parametrize("K,R,nlive,what", [(2048, 128, 3, "q/k/v"), (2048, 64, 1, "o"), (2048, 64, 2, "gate/up"), (6144, 64, 1, "down")])
def test_lora_dropout_kernels_at_bench_shapes(hip_device, K, R, nlive, what):
    """the dropout kernels of the policy pass at the shapes bench.py runs them (M = 8 x 2436 = 19 488 rows, p = 0.05, rank padded to
    64 / 128 with 1-3 live 32-column blocks): lora_down_drop / lora_up_drop / wgrad_tn(drop=...) against fp32 torch ON THE GPU with the
    masks the kernels regenerate exported by bra_dropout_mask (the small-M parametrisations above top out at M = 130)"""
    dev = hip_device
    M, p = 8 * 2436, 0.05
    seeds = [101, 202, 303][:nlive]
    x, A, dts = rnd(M, K, dev=dev, seed=1), rnd(R, K, dev=dev, scale=K ** -0.5, seed=2), rnd(M, R, dev=dev, seed=3)
    A[32 * nlive:] = 0
    dts[:, 32 * nlive:] = 0
    masks = [ops.dropout_mask(M, K, p, seeds[j], dev).float() for j in range(nlive)]
    keep = torch.stack(masks).mean().item()
    assert abs(keep - (1 - p)) < 2e-3, keep                              # 40-120 M draws: the keep rate is tight
    xd = [(x.float() * mk / (1 - p)).to(BF).float() for mk in masks]     # torch: scale in fp32, round once
    Af = A.float()
    t = ops.lora_down_drop(x, A, 2.0, p, seeds)
    want_t = torch.zeros(M, R, device=dev)
    for j in range(nlive):
        want_t[:, 32 * j:32 * j + 32] = 2.0 * xd[j] @ Af[32 * j:32 * j + 32].T
    assert rel(t, want_t.to(BF)) < 4e-3 and (t[:, 32 * nlive:] == 0).all(), what
    up = ops.lora_up_drop(dts, A.T.contiguous(), p, seeds)
    want_up = sum((dts.float()[:, 32 * j:32 * j + 32] @ Af[32 * j:32 * j + 32]) * masks[j] / (1 - p) for j in range(nlive))
    assert rel(up, want_up.to(BF)) < 4e-3, what
    dA = torch.zeros(R, K, device=dev)
    ops.wgrad_tn(x, dts, dA, transposed_out=True, drop=(p, seeds))
    want_dA = torch.zeros(R, K, device=dev)
    for j in range(nlive):
        want_dA[32 * j:32 * j + 32] = dts.float()[:, 32 * j:32 * j + 32].T @ xd[j]
    assert rel(dA, want_dA) < 2e-4 and (dA[32 * nlive:] == 0).all(), what       # fp32 atomics over 19 488 rows: summation order


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,R,nlive", [(2180, 2048, 128, 3), (2048, 6144, 64, 1), (2180, 2048, 64, 2)])
def test_lora_down_drop_split_k_at_bench_shapes(hip_device, M, K, R, nlive):
    """the split-K form at the shapes of the shared-prompt passes (one prompt = 2180 rows, eight completions = 2048 rows)"""
    dev, p = hip_device, 0.05
    seeds = [101, 202, 303][:nlive]
    x, A = rnd(M, K, dev=dev, seed=1), rnd(R, K, dev=dev, scale=K ** -0.5, seed=2)
    A[32 * nlive:] = 0
    want = torch.zeros(M, R, device=dev)
    for j in range(nlive):
        mk = ops.dropout_mask(M, K, p, seeds[j], dev).float()
        want[:, 32 * j:32 * j + 32] = 2.0 * (x.float() * mk / (1 - p)).to(BF).float() @ A.float()[32 * j:32 * j + 32].T
    assert ops.LORA_DOWN_SPLITK
    t = ops.lora_down_drop(x, A, 2.0, p, seeds)
    assert rel(t, want.to(BF)) < 4e-3 and (t[:, 32 * nlive:] == 0).all()
    assert torch.equal(t, ops.lora_down_drop(x, A, 2.0, p, seeds))
