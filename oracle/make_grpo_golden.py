"""CPU ORACLE fixture writer (test infrastructure; run in the build container, where /root/reference exists).

Executes the reference's own GRPO statements (oracle/ref_exec.py: grpo_trainer.py:510-520, 605-609, 679-699, 751-814) and the
installed HF logits warpers (TF:generation/logits_process.py TemperatureLogitsWarper / TopKLogitsWarper / TopPLogitsWarper, in
the order GenerationMixin._get_logits_processor applies them for grpo_trainer.py:384-391) on seeded inputs and writes inputs +
outputs to tests/golden/grpo_ref.pt, so that `oracle/grpo_math.py` stays pinned on the GPU box, where the reference is absent.

    python oracle/make_grpo_golden.py            # rewrite the fixture
    python oracle/make_grpo_golden.py --check    # recompute and compare with the committed fixture (exit 1 on drift)
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_exec as R      # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "grpo_ref.pt")


def loss_cases():
    """(name, kwargs of ref_exec.compute_loss) — μ > 1, β = 0, a fully masked-after-first-token row, a zero advantage group"""
    out = []
    for name, seed, B, P, C, V, beta, mu, el, eh in [
        ("mu1_beta", 11, 4, 5, 7, 23, 0.04, 1, 0.2, 0.2),
        ("mu2_beta", 12, 4, 3, 6, 17, 0.04, 2, 0.2, 0.28),
        ("mu1_beta0", 13, 2, 4, 5, 19, 0.0, 1, 0.2, 0.2),
        ("mu3_beta0_asym", 14, 6, 2, 9, 31, 0.0, 3, 0.1, 0.3),
    ]:
        g = torch.Generator().manual_seed(seed)
        logits = torch.randn(B, P + C, V, generator=g) * 2.0
        prompt_ids = torch.randint(0, V, (B, P), generator=g)
        completion_ids = torch.randint(0, V, (B, C), generator=g)
        prompt_mask = torch.ones(B, P, dtype=torch.long)
        prompt_mask[0, :1] = 0                                   # one left-padded row
        cmask = torch.ones(B, C, dtype=torch.int32)
        cmask[1, 1:] = 0                                         # a row that ended after its first token
        cmask[-1, C // 2:] = 0
        adv = torch.randn(B, generator=g)
        adv[B // 2] = 0.0                                        # member of a zero-std group
        ref = torch.randn(B, C, generator=g) * 0.3 - 3.0 if beta > 0 else None
        old = (torch.randn(B, C, generator=g) * 0.2 - 3.0) if mu > 1 else None
        out.append((name, dict(logits=logits, prompt_ids=prompt_ids, prompt_mask=prompt_mask, completion_ids=completion_ids,
                               completion_mask_=cmask, advantages_=adv, ref_per_token_logps=ref, old_per_token_logps=old,
                               beta=beta, epsilon_low=el, epsilon_high=eh, num_iterations=mu)))
    return out


def warp_cases():
    """(name, logits [rows, V], temperature, top_k, top_p): ties at the k-th value, top_p boundary hit exactly, k > support"""
    g = torch.Generator().manual_seed(5)
    cases = [
        ("grpo_defaults", torch.randn(6, 97, generator=g) * 3.0, 0.6, 20, 0.95),
        ("ties_at_kth", torch.tensor([[2.0, 1.0, 1.0, 1.0, 0.5, -1.0], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]]), 1.0, 2, 1.0),
        ("top_p_boundary", torch.log(torch.tensor([[0.5, 0.25, 0.125, 0.125], [0.9, 0.05, 0.03, 0.02]])), 1.0, 0, 0.75),
        ("k_gt_support", torch.randn(3, 5, generator=g), 0.8, 64, 0.9),
        ("top_p_tiny", torch.randn(4, 33, generator=g) * 4.0, 0.7, 20, 0.05),
        ("temp_only", torch.randn(2, 11, generator=g), 1.7, 0, 1.0),
    ]
    return cases


def hf_warped_probs(logits, temperature, top_k, top_p):
    """softmax of the installed HF warpers in HF's order (temperature, top-k, top-p; TF:generation/utils.py _get_logits_processor)"""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    scores = logits.float().clone()
    ids = torch.zeros((scores.shape[0], 1), dtype=torch.long)
    if temperature is not None and temperature != 1.0:
        scores = TemperatureLogitsWarper(temperature)(ids, scores)
    if top_k is not None and top_k != 0:
        scores = TopKLogitsWarper(top_k=top_k, min_tokens_to_keep=1)(ids, scores)
    if top_p is not None and top_p < 1.0:
        scores = TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=1)(ids, scores)
    return torch.softmax(scores, dim=-1)             # _sample: probs = softmax(next_token_scores) (TF:generation/utils.py:2917)


def build():
    fix = {"loss": {}, "mask": {}, "adv": {}, "logps": {}, "warp": {}}
    for name, kw in loss_cases():
        lg = kw["logits"].clone().requires_grad_(True)
        loss, metrics = R.compute_loss(**{**kw, "logits": lg})
        loss.backward()
        fix["loss"][name] = {"in": kw, "loss": loss.detach(), "dlogits": lg.grad.clone(),
                             "kl": metrics.get("kl"), "clip_ratio": metrics["clip_ratio"]}
    g = torch.Generator().manual_seed(21)
    ids = torch.randint(0, 6, (9, 13), generator=g)
    ids[3] = 4                                                     # a row without EOS (eos = 2)
    ids[4, 0] = 2                                                  # EOS at the first position
    fix["mask"]["a"] = {"ids": ids, "eos": 2, "mask": R.completion_mask(ids, 2)}
    for name, G, n, F, zero in [("g4", 4, 12, 3, None), ("g8_zero_std", 8, 16, 5, 1), ("g2", 2, 6, 1, None)]:
        r = torch.rand(n, F, generator=g) * 2.0
        if zero is not None:
            r[zero * G:(zero + 1) * G] = r[zero * G]               # a group whose rewards are all equal: std = 0
        adv, loc = R.advantages(r, G)
        parts = [r[:n // 2], r[n // 2:]]                          # two ranks (a group may span them), accelerator.gather = cat
        adv1, _ = R.advantages(parts[1], G, gather=lambda x: torch.cat(parts), process_index=1)
        fix["adv"][name] = {"rewards_per_func": r, "G": G, "advantages": adv, "mean": loc["mean_grouped_rewards"],
                            "std": loc["std_grouped_rewards"], "rank1_of_2": adv1}
    lg = torch.randn(3, 8, 29, generator=g)
    ii = torch.randint(0, 29, (3, 8), generator=g)
    fix["logps"]["a"] = {"logits": lg, "input_ids": ii, "logps": R.per_token_logps(R._FakeModel(lg), ii, torch.ones_like(ii))}
    for name, lgts, T, k, p in warp_cases():
        fix["warp"][name] = {"logits": lgts, "temperature": T, "top_k": k, "top_p": p, "probs": hf_warped_probs(lgts, T, k, p)}
    return fix


def _same(a, b):
    if torch.is_tensor(a):
        return torch.is_tensor(b) and a.shape == b.shape and torch.equal(a, b)
    if isinstance(a, dict):
        return isinstance(b, dict) and a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    return a == b


if __name__ == "__main__":
    if not R.available():
        sys.exit("needs /root/reference (build container)")
    fix = build()
    if "--check" in sys.argv:
        old = torch.load(OUT, weights_only=False)
        ok = _same(fix, old)
        print("grpo_ref.pt", "matches the reference's statements" if ok else "DIFFERS from a fresh run")
        sys.exit(0 if ok else 1)
    torch.save(fix, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
