"""fp8 (OCP e4m3) rollout weights — BASELINE config 5 "GRPO fp8 weights"; opt-in, never the bf16 headline.

Parity criterion (stated): the kernels against the fp32 statement over THE SAME quantised weights — q and scale are read back from
the device, decoded on the host (torch.float8_e4m3fn: the OCP encoding gfx950 uses) and multiplied in fp32, so only the kernel's own
arithmetic is under test, within one bf16 output rounding (4e-3) like every other decode projection; the quantiser itself is checked
separately: every code is the nearest e4m3 value of w / scale (ties to even), scale = row amax / 448.  End to end: greedy tokens and
logits of a rollout over fp8 weights against the op-by-op bf16 decode path run on the fake-quantised (decoded) weights."""
import os

import pytest
import torch

from bioreason_amd import ops

BF = torch.bfloat16
F8 = torch.float8_e4m3fn


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def rnd(*shape, dev, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(dev)


def unpack_fp8(q: torch.Tensor, N: int, K: int, diag: bool) -> torch.Tensor:
    """inverse of bra_dec_pack_weights_fp8's fragment order -> the e4m3 codes as a row-major uint8 [N, K] matrix"""
    KS, NCOL = (64, 8) if diag else (32, 16)
    npair = K // KS // 2
    c = q.cpu().reshape(N // NCOL, npair, 64, 2, 8)                           # [tile][pair][lane][half][8]
    lane = torch.arange(64)
    fr, fq = lane % 16, lane // 16
    row_in = (fr % 8) if diag else fr
    koff = ((fr // 8) * 32 + fq * 8) if diag else fq * 8
    out = torch.zeros((N, K), dtype=torch.uint8)
    for t in range(N // NCOL):
        for p in range(npair):
            for h in range(2):
                cols = (2 * p + h) * KS + koff[:, None] + torch.arange(8)[None, :]      # [64, 8]
                rows = (t * NCOL + row_in)[:, None].expand(64, 8)
                out[rows, cols] = c[t, p, :, h]
    return out


def test_e4m3_codec_matches_torch_float8():
    """the software encoder of bra_device.h (shared by the emulator and the device pack kernel) against torch's float8_e4m3fn over every
    code and over values around every rounding boundary — exercised through a 1-row pack with scale pinned by a 448 element"""
    codes = torch.arange(256, dtype=torch.uint8)
    vals = codes.view(F8).float()
    finite = torch.isfinite(vals)
    assert finite.sum() == 254 and vals[finite].abs().max() == 448.0           # OCP e4m3fn: no infinities, two NaN codes
    # round trip of every finite value through torch's own encoder: identity (sanity of the reference used below)
    assert torch.equal(vals[finite].to(F8).view(torch.uint8), codes[finite])


@pytest.mark.parametrize("N,K,act,f32", [(64, 2048, 0, 0), (64, 6144, 0, 0), (48, 2048, 1, 0), (80, 2048, 0, 1), (256, 2048, 0, 0)])
def test_fp8_pack_is_nearest_even_quantisation(backend, N, K, act, f32):
    W = rnd(N, K, dev=backend, scale=0.05, seed=1)
    W[3, 5] = 0.0
    W[1] = 0                                                                     # an all-zero row: scale 1, codes 0
    nw = (1.0 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(2))).to(BF).to(backend)
    for norm in (None, nw):
        got = ops.dec_pack_weights_fp8(W, act=bool(act), out_f32=bool(f32), norm_w=norm)
        assert got is not None
        q, scale = got
        diag = (not act) and (not f32) and N % 8 == 0 and K % 64 == 0 and (N + 15) // 16 < 256 and N // 8 <= 256
        codes = unpack_fp8(q, N, K, diag)
        wf = W.float().cpu() * (norm.float().cpu() if norm is not None else 1.0)
        amax = wf.abs().amax(1)
        want_scale = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
        assert torch.allclose(scale.cpu(), want_scale, rtol=1e-6, atol=0)
        # every code is the e4m3 value nearest to w / scale: compare with torch's round-to-nearest-even encoder; the device divides with a
        # reciprocal (1 ulp), so a value within 1e-6 relative of a rounding boundary may land on either neighbour
        t = wf * (1.0 / scale.cpu())[:, None]                                    # the pack kernel multiplies by the reciprocal
        want = t.to(F8).view(torch.uint8)
        same = codes == want
        if not bool(same.all()):
            dec_got, dec_want = codes.view(F8).float(), want.view(F8).float()
            bad = ~same
            mid = 0.5 * (dec_got[bad] + dec_want[bad])
            assert ((t[bad] - mid).abs() <= 2e-6 * t[bad].abs() + 1e-12).all(), "a code is not the nearest e4m3 value"
            assert bad.sum() <= max(4, N * K // 20000)
        assert (codes[1] == 0).all() and scale[1].item() == 1.0
        assert int(codes.view(F8).float().abs().max()) == 448                   # the row maximum uses the full range
    assert ops.dec_pack_weights_fp8(rnd(40, 48, dev=backend)) is None          # not a tile multiple
    assert ops.dec_pack_weights_fp8(rnd(64, 96, dev=backend)) is None          # an odd number of 32-deep k-steps


@pytest.mark.parametrize("M,N,K,act,f32", [(8, 64, 2048, 0, 0), (5, 2048, 2048, 0, 0), (8, 64, 6144, 0, 0), (8, 96, 2048, 1, 0),
                                            (7, 80, 2048, 0, 1), (8, 4096, 2048, 0, 0), (8, 8208, 2048, 0, 1), (8, 12288, 2048, 1, 0),
                                            (8, 2048, 6144, 0, 0)])
def test_dec_gemm2_fp8_against_fp32_over_the_same_quantised_weights(backend, M, N, K, act, f32):
    if backend.type == "cpu" and N * K > 3000000:
        pytest.skip("emulator: small shapes only")
    x, W = rnd(M, K, dev=backend, seed=3), rnd(N, K, dev=backend, scale=0.05, seed=4)
    nw = (1.0 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(5))).to(BF).to(backend)
    diag = (not act) and (not f32) and N % 8 == 0 and K % 64 == 0 and (N + 15) // 16 < 256 and N // 8 <= 256
    xf = x.float().cpu()

    def statement(q, scale, folded, res):
        wq = unpack_fp8(q, N, K, diag).view(F8).float()                          # decoded codes, row-major
        y = (xf @ wq.T) * scale.cpu()[None, :]
        if folded:
            y = y * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + 1e-6)
        if act:
            y = y.view(M, N // 16, 2, 8)
            y = (torch.nn.functional.silu(y[:, :, 0].to(BF).float()).to(BF).float() * y[:, :, 1].to(BF).float()).reshape(M, N // 2)
        if res is not None:
            y = y.to(BF).float() + res.float().cpu()
        return y

    # (1) norm folded: qkv / gate-up / lm_head forms
    q, sc = ops.dec_pack_weights_fp8(W, act=bool(act), out_f32=bool(f32), norm_w=nw)
    ss = ops.row_sumsq(x, 256)
    tm = torch.full((M, N // 16), -7.0, device=backend) if f32 else None
    got = ops.dec_gemm2_fp8(x, q, sc, ss_in=ss, act=bool(act), out_f32=bool(f32), tile_max=tm)
    assert got is not None
    y, _ = got
    want = statement(q, sc, True, None)
    assert rel(y, want) < (2e-5 if f32 else 4e-3), rel(y, want)
    if f32:
        assert torch.equal(tm.cpu(), y.float().view(M, N // 16, 16).amax(-1).cpu())
    # (2) no norm, residual + statistics: o / down forms
    if not act and not f32:
        q2, sc2 = ops.dec_pack_weights_fp8(W)
        r = rnd(M, N, dev=backend, seed=6)
        y2, s2 = ops.dec_gemm2_fp8(x, q2, sc2, res=r, want_ss=True)
        want2 = statement(q2, sc2, False, r)
        assert rel(y2, want2) < 4e-3
        assert rel(s2[:M].sum(1), (y2.float() ** 2).sum(1)) < 1e-2
        # against the bf16 kernel on the decoded weights (what "fake-quantised" means for the bf16 path): same products
        wdq = (unpack_fp8(q2, N, K, diag).view(F8).float()).to(BF).to(backend)   # codes are exact in bf16
        yb, _ = ops.dec_gemm2(x, wdq, res=None)
        yq, _ = ops.dec_gemm2_fp8(x, q2, sc2)
        assert rel(yq.float().cpu(), yb.float().cpu() * sc2.cpu()[None, :]) < 6e-3
    # quantisation itself costs what e4m3 costs: a few percent against the unquantised product
    yref = ((xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + 1e-6) * nw.float().cpu()) @ W.float().cpu().T)
    if not act:
        assert 5e-3 < rel(y, yref) < 6e-2


def test_fp8_shapes_outside_the_single_round_form_are_refused(backend):
    x = rnd(8, 9728, dev=backend)
    W = rnd(64, 9728, dev=backend, scale=0.05)
    assert ops.dec_pack_weights_fp8(W) is None                                    # K = 9728 needs two register rounds: no image, bf16 stays
    assert ops.dec_pack_weights_fp8(rnd(64, 256, dev=backend)) is None            # K = 256: less than one round of the 4-wave form
    x12 = rnd(12, 2048, dev=backend)
    q, sc = ops.dec_pack_weights_fp8(rnd(64, 2048, dev=backend))
    assert ops.dec_gemm2_fp8(x12, q, sc) is None                                  # more than 8 rows


def unpack_fp8_fast(q: torch.Tensor, N: int, K: int, diag: bool) -> torch.Tensor:
    """vectorised `unpack_fp8` (the loop form above is the readable statement; both are compared in the test below)"""
    c = q.cpu()
    if diag:      # [tile][pair][fq][kh][r][half][e] -> row = tile * 8 + r, col = ((pair * 2 + half) * 2 + kh) * 32 + fq * 8 + e
        c = c.reshape(N // 8, K // 128, 4, 2, 8, 2, 8).permute(0, 4, 1, 5, 3, 2, 6)
    else:         # [tile][pair][fq][fr][half][e]     -> row = tile * 16 + fr, col = ((pair * 2 + half) * 4 + fq) * 8 + e
        c = c.reshape(N // 16, K // 64, 4, 16, 2, 8).permute(0, 3, 1, 4, 2, 5)
    return c.reshape(N, K).contiguous()


def test_unpack_helpers_agree(backend):
    for (N, K, act) in [(64, 2048, 0), (48, 2048, 1)]:
        W = rnd(N, K, dev=backend, scale=0.05, seed=9)
        q, _ = ops.dec_pack_weights_fp8(W, act=bool(act))
        diag = not act
        assert torch.equal(unpack_fp8(q, N, K, diag), unpack_fp8_fast(q, N, K, diag))


@pytest.mark.gpu
def test_rollout_over_fp8_weights_against_the_oracle_with_the_same_fake_quantised_weights(hip_device, monkeypatch):
    """End to end at Qwen3-1.7B widths (2 layers, V = 8192): 8 rollouts of one prompt, greedy, teacher-forced.  The token loop streams
    e4m3 weights (`model.rollout_fp8`); the prompt pass stays bf16 (MFMA-bound: nothing to gain).  Oracle = the installed HF Qwen3 in
    fp32: prompt K / V from the bf16 weights, every decode step through a copy whose linears hold the DECODED quantised weights
    (scale x e4m3 code, norm weights folded as the device folded them, lm_head untied).  Criterion as everywhere: rel(hip, fp32) <=
    1.25 x rel(oracle in bf16, fp32) on the step logits; a greedy token may differ only inside the oracle's own near-tie margin."""
    import copy
    from bioreason_amd import configs, generation
    from bioreason_amd.modeling import Qwen3ForCausalLM
    from oracle import dna_llm_oracle as O
    monkeypatch.setenv("BRA_FP8_PREFILL", "0")        # this test pins the TOKEN LOOP (W8A16) against an oracle whose prompt K / V are bf16;
    dev = hip_device                                  # the W8A8 prompt pass has its own oracle: tests/test_fp8_gemm.py
    L, V, P, T, copies = 2, 8192, 200, 12, 8
    tc = dict(vocab_size=V, hidden_size=2048, intermediate_size=6144, num_hidden_layers=L, num_attention_heads=16, num_key_value_heads=8,
              head_dim=128, rope_theta=1e6, max_position_embeddings=4096)
    m = Qwen3ForCausalLM(configs.qwen3_config(**tc), device=dev)
    m.init_weights(0.02, seed=1)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if "norm" in n_ and n_.endswith("weight"):
                p_.copy_((1.0 + 0.1 * torch.randn(p_.shape, generator=g)).to(p_.dtype).to(dev))
    m.ensure_packed()
    emb = (torch.randn(1, P, 2048, generator=g) * 0.02).to(BF).to(dev).repeat(copies, 1, 1)
    mask = torch.ones(copies, P, dtype=torch.long, device=dev)
    mask[:, :5] = 0
    kw = dict(max_new_tokens=T, do_sample=False, eos_token_id=None, prompt_alias=[0] * copies, use_graph=False)
    # ---- the bf16 rollout first (tokens to force), then the fp8 rollout
    tok_bf = generation.generate(m, emb, mask, **kw)
    m.rollout_fp8 = True
    tr8 = []
    tok_f8 = generation.generate(m, emb, mask, force_tokens=tok_bf, trace_logits=tr8, **kw)
    rw = generation.rollout_weights(m, rows=copies)
    assert all(R.get("fp8") for R in rw), "the fp8 images were not built for these shapes"
    head_q, head_s = generation.packed_head_fp8(m)
    assert len(tr8) == T - 1
    # ---- oracle
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    ora_bf = O.make_qwen3(tc, "eager")
    missing, unexpected = ora_bf.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "lm_head" not in k], missing[:3]
    ora_bf.tie_weights()
    ora_bf.eval()
    ora_fq = copy.deepcopy(ora_bf)
    ora_fq.config.tie_word_embeddings = False

    def dq(q, s, N, K, diag):
        return unpack_fp8_fast(q, N, K, diag).view(F8).float() * s.float().cpu()[:, None]
    H, F_, Nq, Nkv = 2048, 6144, 2048, 1024
    for li, R in enumerate(rw):
        lay = ora_fq.model.layers[li]
        wqkv = dq(R["Wqkv_q"], R["Wqkv_s"], Nq + 2 * Nkv, H, False)             # N = 4096: 16-column tiles
        lay.self_attn.q_proj.weight.data, lay.self_attn.k_proj.weight.data, lay.self_attn.v_proj.weight.data = \
            wqkv[:Nq].clone(), wqkv[Nq:Nq + Nkv].clone(), wqkv[Nq + Nkv:].clone()
        lay.input_layernorm.weight.data.fill_(1.0)                                # (folded into the quantised weights)
        lay.self_attn.o_proj.weight.data = dq(R["Wo_q"], R["Wo_s"], H, Nq, True)
        wgu = dq(R["Wgu_q"], R["Wgu_s"], 2 * F_, H, False).view(F_ // 8, 2, 8, H)  # gate / up rows interleaved in blocks of 8
        lay.mlp.gate_proj.weight.data, lay.mlp.up_proj.weight.data = wgu[:, 0].reshape(F_, H).clone(), wgu[:, 1].reshape(F_, H).clone()
        lay.post_attention_layernorm.weight.data.fill_(1.0)
        lay.mlp.down_proj.weight.data = dq(R["Wd_q"], R["Wd_s"], H, F_, True)
    ora_fq.model.norm.weight.data.fill_(1.0)
    ora_fq.lm_head.weight = torch.nn.Parameter(dq(head_q, head_s, V, H, False))
    assert ora_fq.lm_head.weight.data_ptr() != ora_fq.model.embed_tokens.weight.data_ptr()

    def oracle_steps(dtype):
        a, b = copy.deepcopy(ora_bf).to(dtype), copy.deepcopy(ora_fq).to(dtype)
        for mod in list(a.modules()) + list(b.modules()):                          # rotary buffers stay fp32, as from_pretrained leaves them
            if isinstance(getattr(mod, "inv_freq", None), torch.Tensor):
                mod.inv_freq = mod.inv_freq.float()
        e1 = emb[:1].float().cpu().to(dtype)
        mk = mask[:1].cpu()
        logits = []
        with torch.no_grad():
            pre = a.model(inputs_embeds=e1, attention_mask=mk, use_cache=True)
            past = pre.past_key_values
            am = mk
            for t in range(T - 1):
                tok = tok_bf[0, t].cpu().view(1, 1)
                am = torch.cat([am, torch.ones(1, 1, dtype=am.dtype)], 1)
                out = b(inputs_embeds=a.model.embed_tokens(tok), attention_mask=am, past_key_values=past, use_cache=True)
                past = out.past_key_values
                logits.append(out.logits[0, -1].float())
        return torch.stack(logits)                                                 # [T - 1, V]: logits that choose tokens 1 .. T - 1
    ref32, ref16 = oracle_steps(torch.float32), oracle_steps(torch.bfloat16)
    got = torch.stack([t_[0].float().cpu() for t_ in tr8])
    e_hip, e_ref = rel(got, ref32), rel(ref16, ref32)
    assert e_hip <= 1.25 * e_ref, f"fp8 rollout: rel(hip, fp32) {e_hip:.3e} > 1.25 x rel(oracle bf16, fp32) {e_ref:.3e}"
    # greedy choices of the fp8 rollout (recorded while teacher-forced) against the oracle's arg-max, near-tie tolerant
    n_tie = 0
    for t in range(T - 1):
        ours, theirs = int(tok_f8[0, t + 1]), int(ref32[t].argmax())
        if ours != theirs:
            margin = (ref32[t, theirs] - ref32[t, ours]).item()
            assert 0 <= margin <= 3.0 * e_ref * ref32[t].norm().item() / V ** 0.5 + 1e-3, (t, ours, theirs, margin)
            n_tie += 1
    assert n_tie <= 2
    for j in range(1, copies):
        assert torch.equal(tok_f8[j], tok_f8[0])                                    # copies of one prompt, teacher-forced: identical
    # the fp8 path really ran and really differs from the bf16 rollout by quantisation noise (not by nothing, not by garbage)
    m.rollout_fp8 = False
    trb = []
    generation.generate(m, emb, mask, force_tokens=tok_bf, trace_logits=trb, **kw)
    d = rel(torch.stack([t_[0].float().cpu() for t_ in tr8]), torch.stack([t_[0].float().cpu() for t_ in trb]))
    assert 2e-3 < d < 0.15, d
    if os.environ.get("BRA_FP8_REPORT"):
        print(f"\\n[fp8 rollout] rel(hip, fp32 fake-quant oracle) {e_hip:.3e}, oracle bf16 {e_ref:.3e}, fp8 vs bf16 rollout logits {d:.3e}, near ties {n_tie}")


def test_rollout_fp8_falls_back_to_bf16_where_the_shapes_do_not_fit(backend):
    """`GRPOConfig.rollout_fp8` on a model whose projections are not whole fp8 tiles (the tiny fixtures): the layers keep their bf16
    images and the rollout is the bf16 rollout, token for token — the flag never fails a run"""
    from test_model_parity import GOLD, build, to_dev
    from bioreason_amd import generation
    fix = torch.load(os.path.join(GOLD, "tiny_b.pt"), weights_only=False)
    m = build(fix, backend, True)
    b = to_dev(fix["batch"], backend)
    mm = {"dna_tokenized": b["dna_tokenized"], "batch_idx_map": b["batch_idx_map"]}
    kw = dict(max_new_tokens=5, do_sample=False, eos_token_id=None, use_graph=False)
    want = m.generate(input_ids=b["input_ids"], attention_mask=b["attention_mask"], **mm, **kw)
    m.text_model.rollout_fp8 = True
    assert generation.rollout_fp8_enabled(m.text_model)
    got = m.generate(input_ids=b["input_ids"], attention_mask=b["attention_mask"], **mm, **kw)
    assert torch.equal(got, want)
    assert not any(R.get("fp8") for R in generation.rollout_weights(m.text_model, rows=got.shape[0]))
    from bioreason_amd.trainer import GRPOConfig
    assert GRPOConfig().rollout_fp8 is False                                      # opt-in only


def test_ref_fp8_runs_the_reference_pass_on_the_fp8_path_and_falls_back_where_the_widths_do_not_fit(backend):
    """`GRPOConfig.ref_fp8` on the tiny fixture (hidden 128: the fp8 GEMM's K % 128 == 0 holds): the no-grad reference pass runs W8A8
    (engine.use_fp8 over e4m3 images of the base weights) — its log-probs differ from the bf16 reference pass by quantisation noise, not by
    nothing and not by garbage, and the step trains; the rollout keeps its bf16 weights there (the token loop's fp8 kernel refuses shapes
    outside its single-register-round form) and samples the same tokens.  Widths the fp8 GEMM does not take report `fp8_supported() ==
    False` and keep the bf16 reference pass: the flags never fail a run."""
    from test_model_parity import GOLD, build, to_dev
    from bioreason_amd import configs
    from bioreason_amd.modeling import Qwen3ForCausalLM
    from bioreason_amd.trainer import GRPOConfig, GRPOStepRunner
    fix = torch.load(os.path.join(GOLD, "tiny_a.pt"), weights_only=False)
    res = []
    for fp8 in (False, True):
        m = build(fix, backend, True)
        assert m.text_model.ensure_packed().fp8_supported()
        b = to_dev(fix["batch"], backend)
        b.pop("labels")
        runner = GRPOStepRunner(m, GRPOConfig(num_generations=2, max_completion_length=4, eos_token_id=None, seed=7, learning_rate=1e-3,
                                              rollout_fp8=fp8, ref_fp8=fp8),
                                lambda ids, mask: torch.stack([(ids[:, 0] % 5).float(), (ids[:, 1] % 3).float()], dim=1))
        out = runner.step(b)
        inp = runner._buffered_inputs[0]
        res.append((inp["ref_per_token_logps"].float().cpu().clone(), inp["completion_ids"].cpu().clone(), float(out["loss_t"])))
    (lp16, ids16, l16), (lp8, ids8, l8) = res
    assert torch.equal(ids16, ids8)                                    # same rollout: the token loop kept bf16 weights on these shapes
    assert not torch.equal(lp16, lp8)                                  # the fp8 reference pass ran ...
    assert float((lp16 - lp8).abs().mean()) < 0.3 and torch.isfinite(lp8).all()       # ... and differs by W8A8 noise only
    assert abs(l8) < 10 and l8 == l8
    small = Qwen3ForCausalLM(configs.qwen3_config(vocab_size=64, hidden_size=64, intermediate_size=96, num_hidden_layers=1, num_attention_heads=2,
                                                  num_key_value_heads=1, head_dim=32, rope_theta=1e4, max_position_embeddings=64), device=backend)
    small.init_weights(0.02, seed=1)
    assert not small.ensure_packed().fp8_supported()


@pytest.mark.gpu
def test_fp8_prompt_pass_and_reference_pass_in_a_step_at_qwen3_widths(hip_device, monkeypatch):
    """Qwen3-1.7B widths x 2 layers, one prompt x 8 rollouts: (a) the rollout's prompt pass on the fp8 MFMA path (default with rollout_fp8)
    against the bf16 prompt pass (BRA_FP8_PREFILL=0) — the first-step logits differ by W8A8 noise, not by nothing and not by garbage, the
    K / V cache is written; (b) `per_token_logps_shared_prefix` under engine.use_fp8 (what GRPOConfig.ref_fp8 runs) against the bf16 reference
    pass: log-probs within the same noise band; both deterministic from call to call."""
    from bioreason_amd import configs, generation, grpo
    from bioreason_amd.modeling import Qwen3ForCausalLM
    dev = hip_device
    L, V, P, T, copies = 2, 8192, 300, 6, 8
    tc = dict(vocab_size=V, hidden_size=2048, intermediate_size=6144, num_hidden_layers=L, num_attention_heads=16, num_key_value_heads=8,
              head_dim=128, rope_theta=1e6, max_position_embeddings=4096)
    m = Qwen3ForCausalLM(configs.qwen3_config(**tc), device=dev)
    m.init_weights(0.02, seed=1)
    eng = m.ensure_packed()
    assert eng.fp8_supported()
    g = torch.Generator().manual_seed(3)
    emb = (torch.randn(1, P, 2048, generator=g) * 0.02).to(BF).to(dev).repeat(copies, 1, 1)
    mask = torch.ones(copies, P, dtype=torch.long, device=dev)
    kw = dict(max_new_tokens=T, do_sample=False, eos_token_id=None, prompt_alias=[0] * copies, use_graph=False)
    m.rollout_fp8 = True
    tr_a, tr_b, tr_c = [], [], []
    monkeypatch.setenv("BRA_FP8_PREFILL", "0")
    tok0 = generation.generate(m, emb, mask, trace_logits=tr_a, **kw)
    monkeypatch.setenv("BRA_FP8_PREFILL", "1")
    m.ensure_packed()._rollout = None                                   # (the weight set is cached: rebuild it with the row-major images)
    tok1 = generation.generate(m, emb, mask, force_tokens=tok0, trace_logits=tr_b, **kw)
    tok2 = generation.generate(m, emb, mask, force_tokens=tok0, trace_logits=tr_c, **kw)
    rw = generation.rollout_weights(m, rows=copies)
    assert all(R.get("rm8") is not None for R in rw), "the row-major e4m3 images were not built"
    la, lb, lc = (torch.stack([t_[0].float().cpu() for t_ in tr]) for tr in (tr_a, tr_b, tr_c))
    assert torch.equal(lb, lc)                                          # deterministic
    d = rel(lb, la)
    assert 1e-3 < d < 0.25, d                                           # W8A8 prompt K / V under a W8A16 token loop vs bf16 prompt K / V
    # (b) reference-type pass (no adapters) over prompt + completion, bf16 vs fp8 MFMA path
    m.rollout_fp8 = False

    class Wrap:                                                         # the DNALLMModel surface per_token_logps_shared_prefix uses
        text_model = m

        @staticmethod
        def _inputs_embeds(ids, *a):
            return emb
    pids = torch.zeros(copies, P, dtype=torch.long, device=dev)
    cids = tok0[:, :T].to(torch.long)
    cmask = torch.ones(copies, T, dtype=torch.int32, device=dev)
    with torch.no_grad():
        lp16 = grpo.per_token_logps_shared_prefix(Wrap, pids, mask, cids, cmask, [0] * copies)
        w8 = eng.fp8_weight_images()
        with eng.use_fp8(w8):
            lp8 = grpo.per_token_logps_shared_prefix(Wrap, pids, mask, cids, cmask, [0] * copies)
            lp8b = grpo.per_token_logps_shared_prefix(Wrap, pids, mask, cids, cmask, [0] * copies)
    assert torch.equal(lp8, lp8b)
    assert torch.isfinite(lp8).all()
    dlp = float((lp8.float() - lp16.float()).abs().mean())
    assert 1e-4 < dlp < 0.5, dlp
