"""fp8 x fp8 GEMM on the fp8 MFMA path (round 6; BASELINE config 5 "fp8 weights (CDNA4 fp8 MFMA)"): bra_quant_rows_fp8,
bra_swiglu_quant_fp8, bra_gemm_fp8_nt against an oracle that holds the SAME quantised weights AND activations (torch float8_e4m3fn
decode, fp64 products).  Tolerances: quantised bytes exact up to reciprocal-rounding ties (device `1 / a` may be one ulp off the exact
quotient); GEMM outputs within one rounding of the output type."""
import pytest
import torch

from bioreason_amd import ops

BF16 = torch.bfloat16


def decode_e4m3(q: torch.Tensor) -> torch.Tensor:
    return q.cpu().view(torch.float8_e4m3fn).to(torch.float64)


def ref_quant(x: torch.Tensor, colw=None):
    """the rule of k_quant.hip in fp32: a = max |x colw| / 448 (1 for a zero row); q = e4m3(x colw * (1 / a)), nearest even"""
    xf = x.float()
    if colw is not None:
        xf = xf * colw.float()
    mx = xf.abs().amax(dim=1)
    a = torch.where(mx > 0, mx * (1.0 / 448.0), torch.ones_like(mx))
    q = (xf * (1.0 / a)[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    return q, a


@pytest.mark.parametrize("M,K,colw,rms", [(9, 256, False, False), (5, 512, True, False), (7, 384, False, True), (6, 4096, True, True),
                                          (5, 6144, False, False), (4, 9728, False, True)])        # (every register-chunk count of the kernel)
def test_quant_rows_fp8(backend, M, K, colw, rms):
    g = torch.Generator().manual_seed(M * K)
    x = (torch.randn(M, K, generator=g) * 3).to(BF16)
    x[1] = 0                                                    # an all-zero row: scale 1, bytes 0
    x[2, 5] = 1000.0                                            # one outlier sets the row's scale
    w = (torch.rand(K, generator=g) + 0.5).to(BF16) if colw else None
    eps = 1e-6
    q, s = ops.quant_rows_fp8(x.to(backend), None if w is None else w.to(backend), rms_eps=eps if rms else None)
    qr, a = ref_quant(x, w)
    want_s = a * torch.rsqrt(x.float().pow(2).mean(dim=1) + eps) if rms else a
    assert torch.allclose(s.cpu(), want_s, rtol=1e-5, atol=0)    # (device rsqrt / reciprocal: an ulp)
    diff = (q.cpu().to(torch.int16) - qr.to(torch.int16)).abs()
    assert int((diff > 1).sum()) == 0 and float((diff > 0).float().mean()) < 1e-3
    assert int(q.cpu()[1].sum()) == 0 and abs(float(s.cpu()[1]) / float(want_s[1]) - 1) < 1e-5
    # the largest element of a row lands on +-448
    assert int(q.cpu()[2, 5]) == 0x7e


def test_quant_encoder_is_nearest_even_over_every_bf16_value(backend):
    """every finite bf16 magnitude up to 448 (and a few beyond: saturation) through the activation quantiser with the row scale pinned to
    exactly 1 by a 448 element: the bytes are torch's float8_e4m3fn round-to-nearest-even codes — the device's v_cvt_pk_fp8_f32 and the
    emulator's integer encoder agree with each other and with the weight packer (bra_dec_pack_weights_fp8 uses the integer form)"""
    bits = torch.arange(0, 0x43e0 + 1, dtype=torch.int32)                       # 0 .. 448.0 (0x43e0) as bf16 bit patterns
    mags = bits.to(torch.int16).view(BF16)
    vals = torch.cat([mags, -mags])
    K = 512
    rows = (vals.numel() + K - 2) // (K - 1)
    x = torch.zeros(rows, K, dtype=BF16)
    x[:, 0] = 448.0                                                              # a = 448 / 448 = 1
    flat = torch.zeros(rows * (K - 1), dtype=BF16)
    flat[:vals.numel()] = vals
    x[:, 1:] = flat.view(rows, K - 1)
    q, s = ops.quant_rows_fp8(x.to(backend))
    assert torch.equal(s.cpu(), torch.ones(rows))
    want = x.float().to(torch.float8_e4m3fn).view(torch.uint8)
    got = q.cpu()
    # +0 / -0: both encoders keep the sign of a negative zero; torch does too
    assert torch.equal(got, want), (got != want).nonzero()[:5]


@pytest.mark.parametrize("F", [256, 6144, 9728])
def test_swiglu_quant_fp8(backend, F):
    M = 6
    g = torch.Generator().manual_seed(3)
    gu = (torch.randn(M, 2 * F, generator=g) * 2).to(BF16)
    q, s = ops.swiglu_quant_fp8(gu.to(backend))
    act = ops.swiglu_fwd(gu.to(backend)).cpu()                  # the bf16 path's roundings
    qr, a = ref_quant(act)
    assert torch.allclose(s.cpu(), a, rtol=2e-6, atol=0)
    diff = (q.cpu().to(torch.int16) - qr.to(torch.int16)).abs()
    assert int((diff > 1).sum()) == 0 and float((diff > 0).float().mean()) < 1e-3


def _rand_fp8(g, rows, cols):
    q = torch.randint(0, 256, (rows, cols), generator=g, dtype=torch.int32)
    q = torch.where((q & 0x7f) == 0x7f, q & 0x80, q)           # no NaN codes
    return q.to(torch.uint8)


@pytest.mark.parametrize("M,N,K,res,f32", [(70, 132, 256, False, False), (33, 260, 128, True, False), (256, 128, 384, False, True),
                                           (200, 144, 512, True, False)])
def test_gemm_fp8_nt_against_decoded_products(backend, M, N, K, res, f32):
    """every product of two e4m3 values is exact in fp32; the kernel's fp32 accumulation is compared with fp64 sums of the same
    decoded operands, then one rounding of the output type"""
    g = torch.Generator().manual_seed(M + N + K)
    a8, b8 = _rand_fp8(g, M, K), _rand_fp8(g, N, K)
    # keep magnitudes moderate: mask the exponent's top bit so that values stay within +-1.875 and sums do not swamp bf16's range
    a8, b8 = a8 & 0xbf, b8 & 0xbf
    sa = torch.rand(M, generator=g) + 0.5
    sb = torch.rand(N, generator=g) * 0.1 + 0.01
    r = (torch.randn(M, N, generator=g)).to(BF16) if res else None
    c = ops.gemm_fp8_nt(a8.to(backend), sa.to(backend), b8.to(backend), sb.to(backend), res=None if r is None else r.to(backend), out_f32=f32)
    want = (decode_e4m3(a8) @ decode_e4m3(b8).T) * sa.double()[:, None] * sb.double()[None, :]
    # the matrix pipe does NOT sum the 128 products of an instruction exactly: measured on an MI355X (tools/fp8_accum_probe.py,
    # profiles/r6_r_fp8_accum_probe.txt) the fp32 result is off by up to 1.5e-4 x sum |products| (mean 1e-5..2e-5) — products are
    # aligned to a common exponent and truncated inside the instruction.  The emulator sums in fp32; both sit inside this bound.
    sabs = (decode_e4m3(a8).abs() @ decode_e4m3(b8).abs().T) * sa.double()[:, None] * sb.double()[None, :]
    mag = want.abs()
    if r is not None:
        want = want + r.double()
        mag = mag + r.double().abs()
    got = c.cpu().double()
    scale = mag.max()
    if f32:
        assert float(((got - want).abs() / sabs.clamp_min(1e-30)).max()) < 3e-4
    else:
        # one bf16 rounding of the product, a second one after the residual add (the shared epilogue: rnd(rnd(acc) + res), as a bf16
        # nn.Linear followed by a bf16 add), relative to the terms that were added: a residual may cancel the product
        assert float(((got - want).abs() / (mag * 2 ** -8 * (2.2 if res else 1.2) + 3e-4 * sabs + 1e-6 * scale)).max()) < 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(2180, 4096, 2048), (2048, 2048, 6144), (2180, 12288, 2048), (17440, 2048, 2048),
                                   (2180, 2560, 9728), (2181, 6144, 2560)])          # (the last two: Qwen3-4B widths, a ragged row count)
def test_gemm_fp8_nt_model_shapes(hip_device, M, N, K):
    """the projections of one prompt / eight completions / the SFT rows: quantised from bf16 operands by the kernels themselves,
    against the oracle with the same fake-quantised weights and activations"""
    dev = hip_device
    g = torch.Generator().manual_seed(N)
    x = (torch.randn(M, K, generator=g)).to(BF16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.02).to(BF16).to(dev)
    nw = (torch.rand(K, generator=g) + 0.5).to(BF16).to(dev)
    xq, xs = ops.quant_rows_fp8(x, rms_eps=1e-6)
    wq, ws = ops.quant_rows_fp8(w, colw=nw)
    r = torch.randn(M, N, generator=g).to(BF16).to(dev)
    c = ops.gemm_fp8_nt(xq, xs, wq, ws, res=r)
    A = xq.view(torch.float8_e4m3fn).float() * xs[:, None]
    B = wq.view(torch.float8_e4m3fn).float() * ws[:, None]
    prod = A.double() @ B.double().T
    want = prod + r.double()
    mag = prod.abs() + r.double().abs()
    got = c.double()
    assert float(((got - want).abs() / (mag + 1e-3 * mag.max())).max()) < 2 ** -8 * 2.2           # two roundings: product, then + residual
    # and the quantisation itself is the usual W8A8 distance from the bf16 product (a sanity bound, not a parity claim)
    full = (x.float() * torch.rsqrt(x.float().pow(2).mean(1, keepdim=True) + 1e-6)) @ (w.float() * nw.float()).T + r.float()
    rel = float((c.float() - full).norm() / full.norm())
    assert rel < 0.05, rel


def _fake_quant_rows(t: torch.Tensor) -> torch.Tensor:
    """per-token W8A8 activation quantisation as k_quant.hip states it, in the tensor's own dtype arithmetic: a = absmax / 448, e4m3(x / a) a"""
    tf = t.float()
    mx = tf.abs().amax(dim=-1, keepdim=True)
    a = torch.where(mx > 0, mx * (1.0 / 448.0), torch.ones_like(mx))
    return ((tf * (1.0 / a)).clamp(-448, 448).to(torch.float8_e4m3fn).float() * a).to(t.dtype)


@pytest.mark.parametrize("widths", ["tiny", pytest.param("qwen3_1p7b", marks=pytest.mark.gpu)])
def test_layer_stack_fp8_against_the_oracle_with_fake_quantised_weights_and_activations(backend, widths):
    """the no-grad layer forward on the fp8 MFMA path (engine.use_fp8: the prompt pass of an fp8 rollout, the reference pass under
    GRPOConfig.ref_fp8) against the installed HF Qwen3 whose linears hold the DECODED e4m3 weights (norm weights folded as the device folds
    them) and whose linear INPUTS are fake-quantised per token by a pre-hook.  Criterion: rel(hip, oracle fp32) <= 1.5 x rel(oracle bf16,
    oracle fp32) — a quantiser is discontinuous, so two runs whose activations differ by bf16 rounding pick different codes here and
    there; the bf16 run of the same oracle is the yardstick for how much that moves the result."""
    import copy
    from bioreason_amd import configs
    from bioreason_amd.engine import SeqMeta
    from bioreason_amd.modeling import Qwen3ForCausalLM
    from oracle import dna_llm_oracle as O
    if widths == "tiny":
        tc = dict(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                  head_dim=64, rope_theta=1e6, max_position_embeddings=512)
        B, S = 2, 40
    else:
        if backend.type != "cuda":
            pytest.skip("full widths run on the GPU only")
        tc = dict(vocab_size=8192, hidden_size=2048, intermediate_size=6144, num_hidden_layers=2, num_attention_heads=16, num_key_value_heads=8,
                  head_dim=128, rope_theta=1e6, max_position_embeddings=4096)
        B, S = 1, 600
    H = tc["hidden_size"]
    m = Qwen3ForCausalLM(configs.qwen3_config(**tc), device=backend)
    m.init_weights(0.05 if widths == "tiny" else 0.02, seed=1)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if "norm" in n_ and n_.endswith("weight"):
                p_.copy_((1.0 + 0.1 * torch.randn(p_.shape, generator=g)).to(p_.dtype).to(backend))
    eng = m.ensure_packed()
    emb = (torch.randn(B, S, H, generator=g) * 0.5).to(BF16)
    meta = SeqMeta(B=B, S=S, pos=torch.arange(S, dtype=torch.int32, device=backend).repeat(B),
                   kmask=torch.ones(B, S, dtype=torch.uint8, device=backend), lora_on=False, max_pos=S)
    W8 = eng.fp8_weight_images()
    with torch.no_grad():
        with eng.use_fp8(W8):
            got, _ = eng.forward_hidden(emb.to(backend).reshape(B * S, H).contiguous(), meta, save=False)
        plain, _ = eng.forward_hidden(emb.to(backend).reshape(B * S, H).contiguous(), meta, save=False)
    # ---- oracle: decoded weights + fake-quantised linear inputs
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    ora = O.make_qwen3(tc, "eager")
    missing, _ = ora.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "lm_head" not in k], missing[:3]
    ora.eval()

    def dq(qs):
        return qs[0].cpu().view(torch.float8_e4m3fn).float() * qs[1].float().cpu()[:, None]
    Nq, Nkv, F_ = tc["num_attention_heads"] * tc["head_dim"], tc["num_key_value_heads"] * tc["head_dim"], tc["intermediate_size"]
    for li, R in enumerate(W8):
        lay = ora.model.layers[li]
        wqkv, wgu = dq(R["Wqkv"]), dq(R["Wgu"])
        lay.self_attn.q_proj.weight.data, lay.self_attn.k_proj.weight.data, lay.self_attn.v_proj.weight.data = \
            wqkv[:Nq].clone(), wqkv[Nq:Nq + Nkv].clone(), wqkv[Nq + Nkv:].clone()
        lay.input_layernorm.weight.data.fill_(1.0)                                # (folded into the quantised weights)
        lay.self_attn.o_proj.weight.data = dq(R["Wo"])
        lay.mlp.gate_proj.weight.data, lay.mlp.up_proj.weight.data = wgu[:F_].clone(), wgu[F_:].clone()
        lay.post_attention_layernorm.weight.data.fill_(1.0)
        lay.mlp.down_proj.weight.data = dq(R["Wd"])

    def run(dtype):
        o = copy.deepcopy(ora).to(dtype)
        for mod in o.modules():
            if isinstance(getattr(mod, "inv_freq", None), torch.Tensor):
                mod.inv_freq = mod.inv_freq.float()
            if isinstance(mod, torch.nn.Linear) and mod is not getattr(o, "lm_head", None):
                mod.register_forward_pre_hook(lambda _m, args: (_fake_quant_rows(args[0]),))
        with torch.no_grad():
            return o.model(inputs_embeds=emb.float().to(dtype), attention_mask=torch.ones(B, S, dtype=torch.long)).last_hidden_state.float().reshape(B * S, H)
    ref32, ref16 = run(torch.float32), run(torch.bfloat16)

    def rel(a, b):
        return float((a.float().cpu() - b).norm() / b.norm())
    e_hip, e_ref = rel(got, ref32), rel(ref16, ref32)
    assert e_hip <= 1.5 * e_ref + 2e-3, f"fp8 layer stack: rel(hip, fake-quant fp32) {e_hip:.3e} vs oracle bf16 {e_ref:.3e}"
    # and it is the quantised path that ran: it differs from the bf16 stack by W8A8 noise — not by nothing, not by garbage
    d = rel(got, plain.float().cpu())
    assert 5e-3 < d < 0.2, d
