"""The CPU oracle against the golden fixtures written from the reference itself (oracle/make_golden.py), and — when
/root/reference is present (build container) — against the reference's own class, bit for bit."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import dna_llm_oracle as O   # noqa: E402
from oracle import grpo_math as G        # noqa: E402


def rebuild(fix, lora):
    cfg = fix["config"]
    text, dna = O.make_qwen3(cfg["text"], "eager"), O.make_nt_v2(cfg["dna"], "eager")
    text.load_state_dict({k: v.float() for k, v in fix["state"]["text"].items()}, strict=False)
    dna.load_state_dict({k: v.float() for k, v in fix["state"]["dna"].items() if "inv_freq" not in k}, strict=False)   # (the bf16 fixture would round the rotary buffer)
    text.tie_weights()
    if lora:
        O.apply_lora(text, r=32, alpha=64.0)
        text.load_state_dict({k: v.float() for k, v in fix["state"]["lora"].items()}, strict=False)
    m = O.OracleDNALLM(text, dna, cfg["dna_token_id"])
    m.dna_projection.load_state_dict({k: v.float() for k, v in fix["state"]["proj"].items()})
    return m.eval()


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b"])
@pytest.mark.parametrize("lora", [False, True])
def test_oracle_reproduces_golden(name, lora):
    fix = torch.load(os.path.join(GOLD, f"{name}.pt"), weights_only=False)
    ref = fix["fp32_lora" if lora else "fp32"]
    m = rebuild(fix, lora)
    b = fix["batch"]
    out = m(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()})
    assert torch.allclose(out.logits, ref["logits"], rtol=1e-4, atol=1e-4)
    assert abs(out.loss.item() - ref["loss"].item()) < 1e-4 * max(1, abs(ref["loss"].item()))
    gb = {k: v for k, v in b.items() if k != "labels"}
    gen = m.generate(**gb, max_new_tokens=fix["config"]["gen_tokens"], do_sample=False, eos_token_id=fix["config"]["eos_token_id"],
                     pad_token_id=fix["config"]["eos_token_id"])
    assert torch.equal(gen, ref["greedy_ids"])
    bad = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()}
    bad["input_ids"][0, -1] = fix["config"]["dna_token_id"]
    with pytest.raises(ValueError):
        m(**bad)


def test_grpo_math_known_answers():
    ids = torch.tensor([[4, 5, 2, 7, 2], [1, 1, 1, 1, 1]])
    assert G.completion_mask(ids, 2).tolist() == [[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]]
    r = torch.tensor([[1.0], [3.0], [2.0], [2.0]])
    adv, mean, std = G.group_advantages(r, 2)
    assert torch.allclose(adv, torch.tensor([-0.7071, 0.7071, 0.0, 0.0]), atol=1e-3)
    lp = torch.zeros(1, 3, requires_grad=True)
    loss, kl, clip = G.grpo_loss(lp, None, torch.full((1, 3), -0.1), torch.tensor([2.0]), torch.ones(1, 3), 0.2, 0.2, 0.04)
    d = -0.1
    want = -2.0 + 0.04 * (torch.exp(torch.tensor(d)) - d - 1)
    assert abs(loss.item() - want.item()) < 1e-6 and clip.item() == 0.0
    idx = G.repeat_sampler_indices(6, 4, batch_size=1, seed=3)
    assert len(idx) == 24 and all(idx[i] == idx[i - i % 4] for i in range(24))
    p = G.warp_probs(torch.tensor([[2.0, 1.0, 0.0, -5.0]]), 1.0, 3, 0.9)
    assert p[0, 3] == 0 and abs(p.sum().item() - 1) < 1e-6


@pytest.mark.skipif(not os.path.isdir("/root/reference/bioreason"), reason="reference checkout only exists in the build container")
def test_restatement_equals_reference_class():
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_golden.py"), "--check-only"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_sdpa_fp32_equals_eager_fp32():
    """tests/test_fullsize_parity.py runs the fp32 oracle with attn_implementation="sdpa" (no [H, S, S] tensors kept for the
    backward at S = 2212): in fp32 the two HF attention paths are the same arithmetic up to summation order"""
    import torch
    from oracle import dna_llm_oracle as O
    tc = dict(vocab_size=300, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
              head_dim=16, rope_theta=1e6, max_position_embeddings=256)
    dc = dict(vocab_size=40, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=66)
    outs = []
    for impl in ("eager", "sdpa"):
        torch.manual_seed(0)
        text, dna = O.make_qwen3(tc, impl), O.make_nt_v2(dc, impl)
        g = torch.Generator().manual_seed(1)
        for mdl in (text, dna):
            for p in mdl.parameters():
                p.data = torch.randn(p.shape, generator=g) * (0.5 / max(p.shape[-1], 1) ** 0.5) if p.dim() >= 2 else 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        text.tie_weights()
        m = O.OracleDNALLM(text, dna, 290)
        for p in m.dna_projection.parameters():
            p.data = torch.randn(p.shape, generator=g) * 0.1
        m.eval()
        b = O.synth_batch(seed=3, B=2, n_dna_per_sample=2, Sd=10, text_len=20, vocab_text=280, vocab_dna=40, dna_token_id=290,
                          left_pad=[2, 0], dna_pad={1: 6}, label_tail=8)
        out = m(**b)
        out.loss.backward()
        outs.append((out.logits.detach(), out.loss.detach(), m.dna_projection.weight.grad.clone()))
    keep = b["attention_mask"].bool()
    assert (outs[0][0][keep] - outs[1][0][keep]).abs().max() < 1e-4 * outs[0][0][keep].abs().max()
    assert abs(outs[0][1] - outs[1][1]) < 1e-5 * abs(outs[0][1])
    assert (outs[0][2] - outs[1][2]).abs().max() < 1e-4 * outs[0][2].abs().max()
