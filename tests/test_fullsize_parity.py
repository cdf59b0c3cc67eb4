"""Parity at the BASELINE dimensions: NT-v2-500M (29 x 1024) + Qwen3-1.7B (28 x 2048, V = 151936) + LoRA r=32,
prompt P = 2180, against the CPU oracle run LIVE on the host cores (fp32, and in bf16 as the noise yard-stick).

Batch: 2 rows — row 0 the cfg-2/3 sample (2 DNA sequences x 1024 NT tokens + 128 text tokens), row 1 the same shape with its
second DNA sequence right-padded to 700 valid tokens, hence 324 fewer placeholders and a left-padded text row — followed by
C = 32 completion tokens.  Compared:
  a3   encoder `hidden_states[-1]` of the 4 DNA sequences                                   (dna_llm.py:150-156)
  a5/6 logits at the last 64 positions and the loss (labels on the last 64)                 (dna_llm.py:181-244)
  a4/7 gradients of dna_projection and of the layer-0 / layer-27 LoRA factors               (train_dna_qwen.py:155-167)
  a9   per-token log-probs of the completion, policy and reference (adapters off)           (grpo_trainer.py:510-520, 636-640)
  a8   16 greedy tokens on the FUSED SHARED-PREFIX decode path the bench uses               (dna_llm.py:246-306)
Tolerance, measured inside the test: err(hip, oracle_fp32) <= FACTOR x err(oracle_bf16, oracle_fp32) per quantity — the HIP
path must sit as close to the exact answer as the reference's own bf16 execution of the same weights does; greedy tokens must
equal the fp32 oracle's except where the oracle's own top-2 margin is inside that bf16 noise.
The fp32 oracle uses sdpa (fp32 throughout; `test_oracle.py::test_sdpa_fp32_equals_eager_fp32` pins it to the eager path) so
that 28 layers of [16, S, S] attention weights are not kept for the backward.
"""
import json
import os
import time

import pytest
import torch

from bioreason_amd import configs

FACTOR = float(os.environ.get("BRA_PARITY_FACTOR", "1.25"))
TC = dict(vocab_size=151936, hidden_size=2048, intermediate_size=6144, num_hidden_layers=28, num_attention_heads=16,
          num_key_value_heads=8, head_dim=128, rope_theta=1e6, max_position_embeddings=40960)
DC = dict(vocab_size=4107, hidden_size=1024, intermediate_size=4096, num_hidden_layers=29, num_attention_heads=16,
          max_position_embeddings=2050)
DNA_ID = 151670
# round 4: the completion now has the bench's length (C = 256: log-prob rows up to position 2436) and the teacher-forced greedy
# decode runs 128 tokens — past two 64-key chunk boundaries of the completion cache at positions > 2308 — plus a 16-row decode
SD, TEXT, C, TAIL, NGREEDY, NGREEDY16 = 1024, 128, 256, 64, 128, 32
PRESET = os.environ.get("BRA_FULLSIZE_PRESET", "")
COPIES_WIDE = 8                                # rollouts per prompt of the many-row decode check (2 prompts x 8 = 16 rows)
if PRESET == "qwen3_4b":
    # Qwen3-4B WIDTHS (README.md:84 "NT-500M + Qwen3-4B"; SURVEY section 8 preamble: 2560 / 36 / 32 / 8 / 128 / 9728) on 3 layers and a
    # small encoder: what changes against 1.7B is per-layer geometry — G = 4 query heads per kv-head (8 rollouts = 32 query rows per
    # (prompt, kv-head) in the decode attention), K = 2560 / 4096 / 9728 in the streaming projections — not depth
    TC.update(hidden_size=2560, intermediate_size=9728, num_hidden_layers=3, num_attention_heads=32, num_key_value_heads=8,
              vocab_size=32768)
    DC.update(hidden_size=256, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4)
    DNA_ID = 32000
    SD, TEXT, C, NGREEDY, NGREEDY16 = 160, 60, 96, 80, 72
if os.environ.get("BRA_FULLSIZE_SMALL"):       # plumbing check on a box without the time for the real sizes
    TC.update(num_hidden_layers=2, vocab_size=8192)
    DC.update(num_hidden_layers=2)
    DNA_ID = 8000
    SD, TEXT, C, NGREEDY, NGREEDY16 = 96, 40, 70, 66, 6
    if PRESET == "qwen3_4b":
        TC.update(hidden_size=256, intermediate_size=512, num_attention_heads=8, num_key_value_heads=2, head_dim=64)
RATIOS = {}


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _fill(model, seed):
    g = torch.Generator().manual_seed(seed)
    for n, p in model.named_parameters():
        if p.dim() >= 2:
            p.data = (torch.randn(p.shape, generator=g) * 0.02).to(torch.bfloat16).float()
        elif "norm" in n.lower() and n.endswith("weight"):
            p.data = (1.0 + 0.1 * torch.randn(p.shape, generator=g)).to(torch.bfloat16).float()
        else:
            p.data = (0.02 * torch.randn(p.shape, generator=g)).to(torch.bfloat16).float()


def _batch():
    from oracle import dna_llm_oracle as O
    b = O.synth_batch(seed=5, B=2, n_dna_per_sample=2, Sd=SD, text_len=TEXT, vocab_text=DNA_ID - 30, vocab_dna=DC["vocab_size"],
                      dna_token_id=DNA_ID, left_pad=[0, 0], dna_pad={3: SD * 700 // 1024}, label_tail=TAIL)
    g = torch.Generator().manual_seed(9)
    comp = torch.randint(0, DNA_ID - 30, (2, C), generator=g)
    return b, comp


def _oracle_run(ora, text, b, comp, want_decode):
    """forward/backward over prompt + completion with labels on the last TAIL positions; log-probs of the completion (policy
    from the same logits, reference = one more no-grad pass with adapters off); optionally greedy decode from the prompt"""
    from oracle import dna_llm_oracle as O
    out = {}
    ids = torch.cat([b["input_ids"], comp], 1)
    mask = torch.cat([b["attention_mask"], torch.ones_like(comp)], 1)
    labels = torch.full_like(ids, -100)
    labels[:, -TAIL:] = ids[:, -TAIL:]
    mm = {"dna_tokenized": b["dna_tokenized"], "batch_idx_map": b["batch_idx_map"]}
    with torch.no_grad():
        out["enc"] = ora.dna_model(input_ids=b["dna_tokenized"]["input_ids"], attention_mask=b["dna_tokenized"]["attention_mask"],
                                   output_hidden_states=True).hidden_states[-1].float().clone()
    ora.zero_grad(set_to_none=True)
    fw = ora(input_ids=ids, attention_mask=mask, labels=labels, **mm)
    out["loss"] = fw.loss.detach().float().clone()
    out["logits_tail"] = fw.logits[:, -TAIL:].detach().float().clone()
    lg = fw.logits[:, -C - 1:-1].detach().float()
    out["logps"] = torch.gather(lg.log_softmax(-1), 2, comp.unsqueeze(-1)).squeeze(-1).clone()
    fw.loss.backward()
    del fw, lg
    out["grad_proj_w"] = ora.dna_projection.weight.grad.detach().float().clone()
    out["grad_proj_b"] = ora.dna_projection.bias.grad.detach().float().clone()
    L = len(text.model.layers)
    for li in (0, L - 1):
        lay = text.model.layers[li]
        for nm, mod in (("q", lay.self_attn.q_proj), ("v", lay.self_attn.v_proj), ("o", lay.self_attn.o_proj),
                        ("gate", lay.mlp.gate_proj), ("down", lay.mlp.down_proj)):
            out[f"grad_l{li}_{nm}_A"] = mod.lora_A["default"].weight.grad.detach().float().clone()
            out[f"grad_l{li}_{nm}_B"] = mod.lora_B["default"].weight.grad.detach().float().clone()
    ora.zero_grad(set_to_none=True)
    O.set_adapters(text, False)
    with torch.no_grad():
        lg = ora(input_ids=ids, attention_mask=mask, **mm).logits[:, -C - 1:-1].float()
        out["ref_logps"] = torch.gather(lg.log_softmax(-1), 2, comp.unsqueeze(-1)).squeeze(-1).clone()
    O.set_adapters(text, True)
    if want_decode:
        embeds = ora._inputs_embeds(b["input_ids"], b["dna_tokenized"], b["batch_idx_map"]).detach()
        full = text.generate(inputs_embeds=embeds, attention_mask=b["attention_mask"], use_cache=True, max_new_tokens=NGREEDY,
                             do_sample=False, eos_token_id=None, pad_token_id=0, output_scores=True, return_dict_in_generate=True)
        out["greedy_ids"] = full.sequences.clone()
        out["greedy_scores"] = torch.stack([s.float() for s in full.scores], dim=1).clone()
    return out


@pytest.fixture(scope="module")
def any_device():
    """cuda:0 through libbioreason_hip.so; BRA_FULLSIZE_EMU=1 (with BRA_FULLSIZE_SMALL=1) runs the same test body on the
    kernel-source emulator so that its plumbing can be exercised in a container without a GPU"""
    from bioreason_amd import _lib
    if os.environ.get("BRA_FULLSIZE_EMU"):
        from conftest import EMU_LIB, _build_emu
        _build_emu()
        _lib.use_library_for_tests(EMU_LIB)
        yield torch.device("cpu")
        _lib.reset_library()
        return
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _lib.reset_library()
    assert not _lib.get_lib().emulated
    yield torch.device("cuda:0")
    torch.cuda.synchronize()


@pytest.fixture(scope="module")
def runs(any_device):
    hip_device = any_device
    from oracle import dna_llm_oracle as O
    from bioreason_amd.dna_llm import DNALLMModel
    from transformers.initialization import no_init_weights
    dev = hip_device
    t0 = time.time()
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    with no_init_weights():
        text = O.make_qwen3(TC, "sdpa")
        dna = O.make_nt_v2(DC, "sdpa")
    _fill(text, 1)
    _fill(dna, 2)
    text.tie_weights()
    O.apply_lora(text, r=32, alpha=64.0, dropout=0.0)
    g = torch.Generator().manual_seed(3)
    lora_state = {}
    for n, p in text.named_parameters():
        if "lora_" in n:
            p.data = (torch.randn(p.shape, generator=g) * (0.5 / p.shape[1] ** 0.5)).to(torch.bfloat16).float()
            lora_state[n] = p.data.clone()
    ora = O.OracleDNALLM(text, dna, DNA_ID)
    _fill(ora.dna_projection, 4)
    ora.eval()
    b, comp = _batch()
    t_build = time.time() - t0
    # ---- HIP model from the same numbers
    m = DNALLMModel(configs.qwen3_config(**TC), configs.nt_v2_config(**DC), device=dev, dna_token_id=DNA_ID)
    base = {k.replace(".base_layer", ""): v.to(torch.bfloat16) for k, v in text.state_dict().items() if "lora_" not in k}
    missing, _ = m.text_model.load_state_dict(base, strict=False)
    assert not [k for k in missing if "lora" not in k], missing[:4]
    md, _ = m.dna_model.load_state_dict({k: v.to(torch.bfloat16) for k, v in dna.state_dict().items()
                                         if "lm_head" not in k and "contact_head" not in k and "inv_freq" not in k}, strict=False)
    assert not md, md[:4]
    m.dna_projection.weight.data.copy_(ora.dna_projection.weight.data)
    m.dna_projection.bias.data.copy_(ora.dna_projection.bias.data)
    m.text_model.apply_lora(r=32, alpha=64.0, dropout=0.0, arena=m.arena)
    own = dict(m.text_model.named_parameters())
    for k, v in lora_state.items():
        own[k].data.copy_(v.to(dev))
    m.arena.pack()
    m.eval()
    # ---- oracle: fp32 (the exact answer), then the same modules in bf16 (the reference's own execution noise)
    t0 = time.time()
    fp32 = _oracle_run(ora, text, b, comp, want_decode=True)
    t_fp32 = time.time() - t0
    keep = [(mod, mod.inv_freq.clone()) for mod in ora.modules() if isinstance(getattr(mod, "inv_freq", None), torch.Tensor)]
    ora.to(torch.bfloat16)
    for mod, buf in keep:                       # from_pretrained(torch_dtype=bf16) leaves the rotary buffer in fp32
        mod.inv_freq = buf.clone()
    t0 = time.time()
    bf16 = _oracle_run(ora, text, b, comp, want_decode=False)
    t_bf16 = time.time() - t0
    # ---- the same bf16 oracle modules on THIS GPU through stock PyTorch-ROCm eager kernels: what the reference itself
    # executes on a GPU box — the HIP path is compared with that run directly in test_hip_vs_hf_on_the_same_gpu
    hf_gpu, t_gpu = None, 0.0
    if dev.type == "cuda":
        t0 = time.time()
        ora.to(dev)
        bg = {"input_ids": b["input_ids"].to(dev), "attention_mask": b["attention_mask"].to(dev), "labels": b["labels"].to(dev),
              "dna_tokenized": {k: v.to(dev) for k, v in b["dna_tokenized"].items()}, "batch_idx_map": list(b["batch_idx_map"])}
        hf_gpu = {k: v.cpu() for k, v in _oracle_run(ora, text, bg, comp.to(dev), want_decode=True).items()}
        torch.cuda.synchronize()
        t_gpu = time.time() - t0
    del ora, text, dna
    if dev.type == "cuda":
        torch.cuda.empty_cache()
    print(f"\n[fullsize] oracle build {t_build:.0f}s, fp32 run {t_fp32:.0f}s, bf16 run {t_bf16:.0f}s on {torch.get_num_threads()} threads; "
          f"bf16 oracle on the GPU (stock PyTorch-ROCm) {t_gpu:.0f}s")
    return {"m": m, "b": b, "comp": comp, "fp32": fp32, "bf16": bf16, "hf_gpu": hf_gpu, "dev": dev}


def _dev_batch(b, dev):
    return {"input_ids": b["input_ids"].to(dev), "attention_mask": b["attention_mask"].to(dev),
            "dna_tokenized": {k: v.to(dev) for k, v in b["dna_tokenized"].items()}, "batch_idx_map": list(b["batch_idx_map"])}


def _check(name, got, runs, key=None, sel=None):
    key = key or name
    want, ref = runs["fp32"][key], runs["bf16"][key]
    got = got.float().cpu()
    if sel is not None:
        got, want, ref = got[sel], want[sel], ref[sel]
    e_hip, e_ref = rel(got, want), rel(ref, want)
    RATIOS[name] = {"hip_vs_fp32": e_hip, "refbf16_vs_fp32": e_ref, "ratio": e_hip / max(e_ref, 1e-30)}
    assert e_hip <= FACTOR * e_ref, f"{name}: rel(hip, fp32) = {e_hip:.3e} > {FACTOR} x rel(ref_bf16, fp32) = {e_ref:.3e}"


@pytest.mark.gpu
def test_encoder_hidden_states(runs):
    """a3: NT-v2-500M forward, 29 layers, one right-padded sequence; valid rows only (padded rows are never read, dna_llm.py:168)"""
    m, b, dev = runs["m"], runs["b"], runs["dev"]
    d = {k: v.to(dev) for k, v in b["dna_tokenized"].items()}
    hid = m.dna_model(input_ids=d["input_ids"], attention_mask=d["attention_mask"]).hidden_states[-1]
    _check("enc", hid, runs, sel=b["dna_tokenized"]["attention_mask"].bool())


@pytest.mark.gpu
def test_forward_backward_fullsize(runs):
    m, b, comp, dev = runs["m"], runs["b"], runs["comp"], runs["dev"]
    db = _dev_batch(b, dev)
    ids = torch.cat([db["input_ids"], comp.to(dev)], 1)
    mask = torch.cat([db["attention_mask"], torch.ones_like(comp).to(dev)], 1)
    labels = torch.full_like(ids, -100)
    labels[:, -TAIL:] = ids[:, -TAIL:]
    m.arena.zero_grad()
    m.train()                                    # dropout 0: train mode only switches the backward bookkeeping on
    out = m(input_ids=ids, attention_mask=mask, labels=labels, dna_tokenized=db["dna_tokenized"], batch_idx_map=db["batch_idx_map"])
    _check("logits_tail", out.logits[:, -TAIL:], runs)
    noise = RATIOS["logits_tail"]["refbf16_vs_fp32"]
    dl, dl_ref = abs(out.loss.item() - runs["fp32"]["loss"].item()), abs(runs["bf16"]["loss"].item() - runs["fp32"]["loss"].item())
    RATIOS["loss"] = {"hip_abs": dl, "refbf16_abs": dl_ref, "loss": runs["fp32"]["loss"].item()}
    # a scalar's own bf16 deviation can be accidentally tiny: bound it by the logit noise the loss is a mean over
    assert dl <= FACTOR * max(dl_ref, noise * max(1.0, abs(runs["fp32"]["loss"].item())))
    out.loss.backward()
    _check("grad_proj_w", m.dna_projection.weight.grad, runs)
    _check("grad_proj_b", m.dna_projection.bias.grad, runs)
    L = len(m.text_model.model.layers)
    for li in (0, L - 1):
        lay = m.text_model.model.layers[li]
        for nm, mod in (("q", lay.self_attn.q_proj), ("v", lay.self_attn.v_proj), ("o", lay.self_attn.o_proj),
                        ("gate", lay.mlp.gate_proj), ("down", lay.mlp.down_proj)):
            _check(f"grad_l{li}_{nm}_A", mod.lora_A["default"].weight.grad, runs)
            _check(f"grad_l{li}_{nm}_B", mod.lora_B["default"].weight.grad, runs)
    m.eval()


@pytest.mark.gpu
def test_per_token_logps_fullsize(runs):
    """a9: fused lm_head + LSE + gather over the completion rows at V = 151936, policy and adapter-off reference, through
    both the full-sequence pass and the shared-prefix pass the GRPO step uses for the reference policy"""
    from bioreason_amd import grpo
    m, b, comp, dev = runs["m"], runs["b"], runs["comp"], runs["dev"]
    db = _dev_batch(b, dev)
    cm = torch.ones((2, C), dtype=torch.int32, device=dev)
    mm = {"dna_tokenized": db["dna_tokenized"], "batch_idx_map": db["batch_idx_map"]}
    with torch.no_grad():
        lp = grpo.per_token_logps(m, db["input_ids"], db["attention_mask"], comp.to(dev), cm, **mm)
        with m.text_model.disable_adapter():
            rlp = grpo.per_token_logps(m, db["input_ids"], db["attention_mask"], comp.to(dev), cm, **mm)
            # shared-prefix form: rows [0, 0, 1, 1] = 2 prompts x 2 copies, each copy with its own completion
            rows = [0, 0, 1, 1]
            ids4, mask4 = db["input_ids"][rows], db["attention_mask"][rows]
            dna4 = {k: torch.cat([v[0:2], v[0:2], v[2:4], v[2:4]], 0) for k, v in db["dna_tokenized"].items()}
            comp4 = comp.to(dev)[[0, 1, 1, 0]]
            shared = grpo.per_token_logps_shared_prefix(m, ids4, mask4, comp4, torch.ones((4, C), dtype=torch.int32, device=dev),
                                                        [0, 0, 2, 2], dna_tokenized=dna4, batch_idx_map=[0, 0, 1, 1, 2, 2, 3, 3],
                                                        dna_alias=[0, 1, 0, 1, 4, 5, 4, 5])
    _check("logps", lp, runs)
    _check("ref_logps", rlp, runs)
    # rows of `shared`: (p0,c0) (p0,c1) (p1,c1) (p1,c0); the oracle holds (p0,c0) and (p1,c1)
    got = torch.stack([shared[0], shared[2]])
    _check("ref_logps_shared_prefix", got, runs, key="ref_logps")
    for k in ("logps", "ref_logps"):
        d_hip = (({"logps": lp, "ref_logps": rlp}[k]).float().cpu() - runs["fp32"][k]).abs().max().item()
        d_ref = (runs["bf16"][k] - runs["fp32"][k]).abs().max().item()
        RATIOS[k + "_maxabs"] = {"hip": d_hip, "refbf16": d_ref}
        assert d_hip <= 2.0 * d_ref + 1e-3, (k, d_hip, d_ref)      # max statistics are heavier-tailed than the norms above


@pytest.mark.gpu
def test_shared_policy_pass_fullsize(runs):
    """a9 / a11 on the path the GRPO step runs: the differentiable shared-prompt policy pass (2 prompts x 2 copies, P = 2180, one
    left-padded prompt) — log-probs against the fp32 oracle like the full pass, and the gradient of EVERY trainable parameter
    against the full-sequence pass on the same four rows (the sum over the copies through the shared prompt K / V rows)"""
    from bioreason_amd import grpo
    m, b, comp, dev = runs["m"], runs["b"], runs["comp"], runs["dev"]
    db = _dev_batch(b, dev)
    rows = [0, 0, 1, 1]
    ids4, mask4 = db["input_ids"][rows], db["attention_mask"][rows]
    dna4 = {k: torch.cat([v[0:2], v[0:2], v[2:4], v[2:4]], 0) for k, v in db["dna_tokenized"].items()}
    mm4 = dict(dna_tokenized=dna4, batch_idx_map=[0, 0, 1, 1, 2, 2, 3, 3], dna_alias=[0, 1, 0, 1, 4, 5, 4, 5])
    comp4 = comp.to(dev)[[0, 1, 1, 0]]
    cm4 = torch.ones((4, C), dtype=torch.int32, device=dev)
    w = torch.randn(4, C, generator=torch.Generator().manual_seed(9)).to(dev)
    m.train()

    def run(fn):
        m.arena.zero_grad()
        lp = fn()
        (lp * w).sum().backward()
        return lp.detach().clone(), m.arena.grads.clone()
    lp_sh, g_sh = run(lambda: grpo.per_token_logps_shared_policy(m, ids4, mask4, comp4, cm4, [0, 0, 2, 2], **mm4))
    lp_full, g_full = run(lambda: grpo.per_token_logps(m, ids4, mask4, comp4, cm4, **mm4))
    m.eval()
    _check("logps_shared_policy", torch.stack([lp_sh[0], lp_sh[2]]), runs, key="logps")
    d = (lp_sh - lp_full).abs().max().item()
    gr = ((g_sh - g_full).norm() / g_full.norm()).item()
    noise = max(v.get("refbf16_vs_fp32", 0.0) for k, v in RATIOS.items() if k.startswith("grad_l")) if any(k.startswith("grad_l") for k in RATIOS) else 3e-2
    RATIOS["shared_policy_vs_full"] = {"logps_maxabs": d, "grads_rel": gr, "grad_noise_refbf16": noise}
    # two bf16 executions of the same pass: their largest log-prob difference over the 4 x C positions is bounded by the reference's
    # OWN largest bf16 deviation from fp32 on these rows (measured in test_per_token_logps_fullsize; a max over 8x more positions
    # than round 3's C = 32 is heavier-tailed than the fixed 0.06 it was held to)
    assert "logps_maxabs" in RATIOS, "test_per_token_logps_fullsize must run first (it measures the reference's own bf16 deviation)"
    d_ref = RATIOS["logps_maxabs"]["refbf16"]
    RATIOS["shared_policy_vs_full"]["logps_maxabs_refbf16"] = d_ref
    assert d <= 2.0 * d_ref + 1e-3, (d, d_ref)
    assert gr <= 1.5 * noise, (gr, noise)       # the two passes differ by where bf16 roundings of dK / dV fall: inside the gradients' own bf16 noise


@pytest.mark.gpu
def test_greedy_decode_fused_shared_prefix_fullsize(runs):
    """a8: prefill once per prompt + fused decode steps against one shared copy of the prompt K/V (the bench's rollout path),
    2 prompts x 2 copies, teacher-forced with the oracle's tokens: every choice equals the oracle's arg-max unless the
    oracle's own margin between the two candidates is inside bf16 noise; copies of a prompt must agree exactly"""
    m, b, dev = runs["m"], runs["b"], runs["dev"]
    db = _dev_batch(b, dev)
    rows = [0, 0, 1, 1]
    ids4, mask4 = db["input_ids"][rows], db["attention_mask"][rows]
    dna4 = {k: torch.cat([v[0:2], v[0:2], v[2:4], v[2:4]], 0) for k, v in db["dna_tokenized"].items()}
    want = runs["fp32"]["greedy_ids"][rows]
    scores = runs["fp32"]["greedy_scores"][rows]
    kw = dict(input_ids=ids4, attention_mask=mask4, dna_tokenized=dna4, batch_idx_map=[0, 0, 1, 1, 2, 2, 3, 3],
              dna_alias=[0, 1, 0, 1, 4, 5, 4, 5], prompt_alias=[0, 0, 2, 2], max_new_tokens=NGREEDY, do_sample=False, eos_token_id=None)
    prof = {}
    forced = m.generate(**kw, force_tokens=want.to(dev), profile=prof).cpu()
    free = m.generate(**kw).cpu()
    assert torch.equal(forced[0], forced[1]) and torch.equal(forced[2], forced[3])
    assert torch.equal(free[0], free[1]) and torch.equal(free[2], free[3])
    noise = RATIOS.get("logits_tail", {}).get("refbf16_vs_fp32", 1e-2)
    n_tie = 0
    for bi in range(4):
        for t in range(NGREEDY):
            ours, theirs = int(forced[bi, t]), int(want[bi, t])
            if ours != theirs:
                margin = (scores[bi, t, theirs] - scores[bi, t, ours]).item()
                assert 0 <= margin <= 3.0 * noise * scores[bi, t].norm().item() / scores.shape[-1] ** 0.5 + 1e-3, (bi, t, ours, theirs, margin)
                n_tie += 1
    RATIOS["greedy"] = {"near_ties": n_tie, "positions": 4 * NGREEDY, "tokens": NGREEDY, "last_position": int(ids4.shape[1]) + NGREEDY,
                        "free_run_equal": bool(torch.equal(free, want))}
    # how MANY positions fall inside the margin depends on which equally accurate rounding path produced the logits (random-init weights:
    # the arg-max margins are tiny).  Measured, every flip individually inside the margin asserted above: 4 of 512 (1.7B) with both
    # prefill attention kernels; at Qwen3-4B widths 2 of 320 with the round 1-5 kernel and 8 of 320 with the pipelined one, whose
    # row-level error against fp32 is the same to three digits (tools/attn_pad_check.py, profiles/r6_d_attn_pad_check.txt)
    assert n_tie <= max(4, (4 * NGREEDY) // 32)
    if n_tie == 0:
        assert torch.equal(free, want)


@pytest.mark.gpu
def test_greedy_decode_many_rows_fullsize(runs):
    """a8 at 2 prompts x 8 rollouts = 16 sequences (`bench.py --prompts-per-gpu 2`; at Qwen3-4B widths also 1 x 8 = 32 query rows
    per (prompt, kv-head), split into virtual prompts): the 16-row streaming projections + the shared-prefix attention, teacher-forced
    with the fp32 oracle's tokens of each prompt; all copies of a prompt must agree exactly"""
    m, b, dev = runs["m"], runs["b"], runs["dev"]
    db = _dev_batch(b, dev)
    noise = RATIOS.get("logits_tail", {}).get("refbf16_vs_fp32", 1e-2)
    for tag, rows in (("16rows", [0] * COPIES_WIDE + [1] * COPIES_WIDE), ("8rows_one_prompt", [1] * COPIES_WIDE)):
        n = len(rows)
        ids, mask = db["input_ids"][rows], db["attention_mask"][rows]
        dna = {k: torch.cat([v[2 * r:2 * r + 2] for r in rows], 0) for k, v in db["dna_tokenized"].items()}
        first = {r: rows.index(r) for r in set(rows)}
        want = runs["fp32"]["greedy_ids"][rows][:, :NGREEDY16]
        scores = runs["fp32"]["greedy_scores"][rows]
        kw = dict(input_ids=ids, attention_mask=mask, dna_tokenized=dna, batch_idx_map=[j // 2 for j in range(2 * n)],
                  dna_alias=[2 * first[rows[j // 2]] + j % 2 for j in range(2 * n)], prompt_alias=[first[r] for r in rows],
                  max_new_tokens=NGREEDY16, do_sample=False, eos_token_id=None)
        forced = m.generate(**kw, force_tokens=want.to(dev)).cpu()
        for j, r in enumerate(rows):
            assert torch.equal(forced[j], forced[first[r]]), (tag, j)
        n_tie = 0
        for bi in sorted(first.values()):
            for t in range(NGREEDY16):
                ours, theirs = int(forced[bi, t]), int(want[bi, t])
                if ours != theirs:
                    margin = (scores[bi, t, theirs] - scores[bi, t, ours]).item()
                    assert 0 <= margin <= 3.0 * noise * scores[bi, t].norm().item() / scores.shape[-1] ** 0.5 + 1e-3, (tag, bi, t, ours, theirs, margin)
                    n_tie += 1
        RATIOS["greedy_" + tag] = {"near_ties": n_tie, "positions": len(first) * NGREEDY16, "sequences": n}
        assert n_tie <= max(2, (len(first) * NGREEDY16) // 16)      # measured: 1 - 3 of 64, 0 - 1 of 32; 4B widths 1 - 4 of 144, 1 - 3 of 72 (see above)


@pytest.mark.gpu
def test_hip_vs_hf_on_the_same_gpu(runs):
    """the HIP path against the reference's own bf16 execution ON THIS GPU (the oracle modules through stock
    PyTorch-ROCm eager kernels; `runs["hf_gpu"]`), not only against the CPU oracle: log-probs of the policy and of the adapter-off
    reference, the logits of the last 64 positions, and greedy tokens teacher-forced with the HF-on-GPU run's own tokens.
    Two bf16 executions of one network differ by about the sum of their distances from the exact answer: the bound is
    rel(hip, hf_gpu) <= rel(hip, fp32) + rel(hf_gpu, fp32), each side measured here; a greedy token may differ only where the HF run's
    own top-2 margin is inside that noise."""
    from bioreason_amd import grpo
    if runs["hf_gpu"] is None:
        pytest.skip("no GPU run of the oracle (emulator plumbing run)")
    m, b, comp, dev, hf, fp32 = runs["m"], runs["b"], runs["comp"], runs["dev"], runs["hf_gpu"], runs["fp32"]
    db = _dev_batch(b, dev)
    cm = torch.ones((2, C), dtype=torch.int32, device=dev)
    mm = {"dna_tokenized": db["dna_tokenized"], "batch_idx_map": db["batch_idx_map"]}
    with torch.no_grad():
        lp = grpo.per_token_logps(m, db["input_ids"], db["attention_mask"], comp.to(dev), cm, **mm).float().cpu()
        with m.text_model.disable_adapter():
            rlp = grpo.per_token_logps(m, db["input_ids"], db["attention_mask"], comp.to(dev), cm, **mm).float().cpu()
    out = {}
    for name, got in (("logps", lp), ("ref_logps", rlp)):
        e_pair, e_hip, e_hf = rel(got, hf[name]), rel(got, fp32[name]), rel(hf[name], fp32[name])
        d_pair, d_hf = (got - hf[name]).abs().max().item(), (hf[name] - fp32[name]).abs().max().item()
        out[name] = {"hip_vs_hfgpu": e_pair, "hip_vs_fp32": e_hip, "hfgpu_vs_fp32": e_hf, "maxabs_hip_vs_hfgpu": d_pair, "maxabs_hfgpu_vs_fp32": d_hf}
        assert e_hf <= 2.0 * rel(runs["bf16"][name], fp32[name]) + 1e-6, (name, "the GPU run of the oracle is not a bf16 execution of the same network", e_hf)
        assert e_pair <= FACTOR * (e_hip + e_hf), (name, e_pair, e_hip, e_hf)
        assert d_pair <= 3.0 * d_hf + 1e-3, (name, d_pair, d_hf)
    # greedy: teacher-forced with the HF-on-GPU run's tokens on the fused shared-prefix path (2 prompts x 2 copies)
    rows = [0, 0, 1, 1]
    ids4, mask4 = db["input_ids"][rows], db["attention_mask"][rows]
    dna4 = {k: torch.cat([v[0:2], v[0:2], v[2:4], v[2:4]], 0) for k, v in db["dna_tokenized"].items()}
    want, scores = hf["greedy_ids"][rows], hf["greedy_scores"][rows]
    forced = m.generate(input_ids=ids4, attention_mask=mask4, dna_tokenized=dna4, batch_idx_map=[0, 0, 1, 1, 2, 2, 3, 3],
                        dna_alias=[0, 1, 0, 1, 4, 5, 4, 5], prompt_alias=[0, 0, 2, 2], max_new_tokens=NGREEDY, do_sample=False,
                        eos_token_id=None, force_tokens=want.to(dev)).cpu()
    noise = max(RATIOS.get("logits_tail", {}).get("refbf16_vs_fp32", 1e-2), rel(hf["logits_tail"], fp32["logits_tail"]))
    n_diff = 0
    for bi in (0, 2):
        for t in range(NGREEDY):
            ours, theirs = int(forced[bi, t]), int(want[bi, t])
            if ours != theirs:
                margin = (scores[bi, t, theirs] - scores[bi, t, ours]).item()
                assert 0 <= margin <= 3.0 * noise * scores[bi, t].norm().item() / scores.shape[-1] ** 0.5 + 1e-3, (bi, t, ours, theirs, margin)
                n_diff += 1
    # the HF run's own agreement with the fp32 oracle over the same positions (it free-runs: equal prefixes only)
    agree_fp32 = int((hf["greedy_ids"] == fp32["greedy_ids"]).all(1).sum().item())
    out["greedy"] = {"positions": 2 * NGREEDY, "hip_differs_from_hfgpu_inside_its_margin": n_diff, "hfgpu_rows_equal_to_fp32_oracle": agree_fp32,
                     "logits_tail_hfgpu_vs_fp32": rel(hf["logits_tail"], fp32["logits_tail"])}
    assert n_diff <= max(4, (2 * NGREEDY) // 32)
    RATIOS["vs_hf_on_gpu"] = out


@pytest.mark.gpu
def test_zz_dump_ratios(runs):
    """not a check: leaves the measured error ratios where the round's evidence is collected (gpurun_out/)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    name = "fullsize_parity_ratios.json" if not PRESET else f"{PRESET}_parity_ratios.json"
    with open(os.path.join(root, "gpurun_out", name), "w") as fh:
        json.dump({"factor": FACTOR, "preset": PRESET or "qwen3_1.7b", "text_config": TC, "completion_len": C, "greedy_tokens": NGREEDY,
                   "ratios": RATIOS}, fh, indent=1)
    print("\n[fullsize] " + ", ".join(f"{k}: {v.get('ratio', float('nan')):.2f}" for k, v in RATIOS.items() if "ratio" in v))
