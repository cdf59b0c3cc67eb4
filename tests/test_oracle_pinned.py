"""Pins the GRPO / sampling restatement `oracle/grpo_math.py` (the checker of the GPU tests) to the reference itself:

* against tests/golden/grpo_ref.pt — written by oracle/make_grpo_golden.py from the REFERENCE'S OWN statements
  (grpo_trainer.py:510-520, 605-609, 679-699, 751-814, ast-extracted and executed unmodified) and from the installed HF
  warpers (TF:generation/logits_process.py Temperature / TopK / TopP, HF's order) — runs everywhere, GPU box included;
* when /root/reference is present (build container): against a fresh execution of those statements on new seeded inputs, and
  the committed fixture against a fresh run (no drift);
* `warp_probs` against the installed HF warpers directly (transformers is installed on both boxes).
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import grpo_math as G            # noqa: E402
from oracle import ref_exec as R             # noqa: E402

FIX = torch.load(os.path.join(ROOT, "tests", "golden", "grpo_ref.pt"), weights_only=False)
needs_ref = pytest.mark.skipif(not R.available(), reason="reference checkout only exists in the build container")


def _restated_loss(kw, logits):
    """oracle.grpo_math on the inputs of a reference compute_loss case: log-probs of the completion rows, then the loss"""
    ids = torch.cat([kw["prompt_ids"], kw["completion_ids"]], dim=1)
    am = torch.cat([kw["prompt_mask"], kw["completion_mask_"]], dim=1)
    lp = G.per_token_logps(R._FakeModel(logits), ids, am)[:, kw["prompt_ids"].size(1) - 1:]
    return G.grpo_loss(lp, kw["old_per_token_logps"], kw["ref_per_token_logps"], kw["advantages_"], kw["completion_mask_"],
                       kw["epsilon_low"], kw["epsilon_high"], kw["beta"])


@pytest.mark.parametrize("name", list(FIX["loss"]))
def test_grpo_loss_equals_reference_compute_loss(name):
    c = FIX["loss"][name]
    kw = c["in"]
    lg = kw["logits"].clone().requires_grad_(True)
    loss, kl, clip = _restated_loss(kw, lg)
    loss.backward()
    assert torch.equal(loss.detach(), c["loss"])                     # same torch ops in the same order: bit-equal
    assert torch.equal(lg.grad, c["dlogits"])
    assert abs(clip.item() - c["clip_ratio"][0]) == 0
    if kw["beta"] > 0:
        assert abs(kl.item() - c["kl"][0]) == 0
    else:
        assert kl is None and c["kl"] is None


def test_completion_mask_equals_reference_lines():
    c = FIX["mask"]["a"]
    got = G.completion_mask(c["ids"], c["eos"])
    assert got.dtype == c["mask"].dtype and torch.equal(got, c["mask"])
    assert got[3].all() and got[4].tolist() == [1] + [0] * 12        # no EOS: whole row; EOS first: one position


@pytest.mark.parametrize("name", list(FIX["adv"]))
def test_group_advantages_equal_reference_lines(name):
    c = FIX["adv"][name]
    adv, mean, std = G.group_advantages(c["rewards_per_func"], c["G"])
    assert torch.equal(adv, c["advantages"]) and torch.equal(mean, c["mean"]) and torch.equal(std, c["std"])
    n = c["rewards_per_func"].shape[0]
    adv1, _, _ = G.group_advantages(c["rewards_per_func"], c["G"], process_index=1, local_n=n // 2)
    assert torch.equal(adv1, c["rank1_of_2"])                        # the local slice of rank 1 of 2 (a group may span the ranks)
    if name == "g8_zero_std":
        # (the mean of eight equal floats is off by an ulp, and that ulp is divided by the 1e-4 floor: reference behaviour)
        assert (c["std"][8:16] == 0).all() and (adv[8:16].abs() < 0.05).all()


def test_per_token_logps_equal_reference_method():
    c = FIX["logps"]["a"]
    got = G.per_token_logps(R._FakeModel(c["logits"]), c["input_ids"], torch.ones_like(c["input_ids"]))
    assert torch.equal(got, c["logps"])


@pytest.mark.parametrize("name", list(FIX["warp"]))
def test_warp_probs_equal_hf_warpers_fixture(name):
    c = FIX["warp"][name]
    got = G.warp_probs(c["logits"], c["temperature"], c["top_k"], c["top_p"])
    assert torch.equal(got == 0, c["probs"] == 0), "support differs from the HF warpers"
    assert torch.allclose(got, c["probs"], rtol=0, atol=1e-7)


def test_warp_probs_equal_installed_hf_warpers():
    """fresh seeded rows through the installed warpers (ties, top_p cut exactly on a cumulative value, top_k above the support)"""
    from oracle.make_grpo_golden import hf_warped_probs
    g = torch.Generator().manual_seed(77)
    rows = [
        (torch.randn(16, 211, generator=g) * 2.5, 0.6, 20, 0.95),            # grpo_trainer.py:384-391 defaults
        (torch.randn(8, 50, generator=g).round(), 1.0, 7, 0.9),              # integer logits: many exact ties at the k-th value
        (torch.log(torch.tensor([[0.4, 0.3, 0.2, 0.1]])), 1.0, 0, 0.6),      # 1 - top_p = 0.4 = cum of the two smallest... boundary
        (torch.log(torch.tensor([[0.4, 0.3, 0.2, 0.1]])), 1.0, 0, 0.7),
        (torch.randn(4, 9, generator=g), 0.9, 64, 0.8),                      # k > vocabulary
        (torch.randn(4, 9, generator=g), 1.3, 3, 1.0),
    ]
    for lg, T, k, p in rows:
        want = hf_warped_probs(lg, T, k, p)
        got = G.warp_probs(lg, T, k, p)
        assert torch.equal(got == 0, want == 0), (T, k, p)
        assert torch.allclose(got, want, rtol=0, atol=1e-7), (T, k, p)


@needs_ref
def test_fixture_is_a_fresh_run_of_the_reference_statements():
    from oracle.make_grpo_golden import _same, build
    assert _same(build(), FIX), "tests/golden/grpo_ref.pt is stale: python oracle/make_grpo_golden.py"


@needs_ref
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_restatement_equals_reference_statements_on_fresh_inputs(seed):
    g = torch.Generator().manual_seed(1000 + seed)
    B, P, C, V = 4 + 2 * seed, 3, 5 + seed, 13 + seed
    mu, beta = 1 + seed % 2, 0.04 * ((seed + 1) % 2)
    kw = dict(logits=torch.randn(B, P + C, V, generator=g), prompt_ids=torch.randint(0, V, (B, P), generator=g),
              prompt_mask=torch.ones(B, P, dtype=torch.long), completion_ids=torch.randint(0, V, (B, C), generator=g),
              completion_mask_=(torch.rand(B, C, generator=g) > 0.2).int(), advantages_=torch.randn(B, generator=g),
              ref_per_token_logps=torch.randn(B, C, generator=g) - 2 if beta > 0 else None,
              old_per_token_logps=torch.randn(B, C, generator=g) * 0.1 - 2 if mu > 1 else None, beta=beta, epsilon_low=0.2,
              epsilon_high=0.2, num_iterations=mu)
    kw["completion_mask_"][:, 0] = 1                                # every row keeps at least one token (EOS itself is kept)
    lg_ref = kw["logits"].clone().requires_grad_(True)
    loss_ref, met = R.compute_loss(**{**kw, "logits": lg_ref})
    loss_ref.backward()
    lg = kw["logits"].clone().requires_grad_(True)
    loss, kl, clip = _restated_loss(kw, lg)
    loss.backward()
    assert torch.equal(loss.detach(), loss_ref.detach()) and torch.equal(lg.grad, lg_ref.grad)
    assert clip.item() == met["clip_ratio"][0] and (kl is None or kl.item() == met["kl"][0])
    ids = torch.randint(0, 4, (7, 11), generator=g)
    assert torch.equal(G.completion_mask(ids, 1), R.completion_mask(ids, 1))
    r = torch.rand(4 * (seed + 2), 2 + seed, generator=g)
    a_ref, loc = R.advantages(r, seed + 2)
    a, m, s = G.group_advantages(r, seed + 2)
    assert torch.equal(a, a_ref) and torch.equal(m, loc["mean_grouped_rewards"]) and torch.equal(s, loc["std_grouped_rewards"])
