"""Monte-Carlo evidence for the one semantic deviation of the headline.

Under LoRA dropout the shared-prompt policy pass (grpo.per_token_logps_shared_policy) gives the prompt rows of all G copies of a
prompt ONE mask stream; the reference (`compute_loss`, grpo_trainer.py:777-779 over [B, P + C] rows with PEFT's nn.Dropout) draws
an independent mask per copy.  Claim: every copy's marginal mask distribution is unchanged, so the EXPECTED gradient is the
same; the copies' gradient noise becomes correlated, which can only change the VARIANCE of the group gradient.

This script measures both on the kernel-source emulator (CPU; the kernels' own mask hash and arithmetic), tiny dims, p = 0.2 (4 x
the bench's 0.05, to magnify): for N seeds it runs the GRPO loss backward through (i) the full-row pass with per-copy masks and
(ii) the shared-prompt pass, on the SAME batch / completions / advantages / reference log-probs, and reports
  * || mean_i - mean_s || against its sampling error  (expectation),
  * tr Cov_s / tr Cov_i of the flat gradient           (variance cost), also per parameter family,
  * the same ratio for the loss value.
Usage: python tools/shared_mask_mc.py [--seeds 240] [--p 0.2] [--fixture tiny_b] [--copies 4] [--workers 8] [--out profiles/...json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402


def setup(fixture, copies, p):
    from bioreason_amd import _lib
    _lib.use_library_for_tests(os.path.join(ROOT, "tests", "emu", "libbioreason_emu.so"))
    from test_model_parity import GOLD, build                    # noqa: E402
    from test_shared_policy import _group_batch                   # noqa: E402
    from bioreason_amd import grpo
    dev = torch.device("cpu")
    fix = torch.load(os.path.join(GOLD, f"{fixture}.pt"), weights_only=False)
    m = build(fix, dev, True)
    ids, mask, mm, alias = _group_batch(fix, dev, copies)
    B = ids.shape[0]
    C = 8
    g = torch.Generator().manual_seed(11)
    comp = torch.randint(3, fix["config"]["text"]["vocab_size"] - 8, (B, C), generator=g)
    cmask = torch.ones((B, C), dtype=torch.int32)
    cmask[1, 5:] = 0
    # GRPO advantages: group-normalised rewards, (r - mean_group) / (std_group + 1e-4) (grpo_trainer.py:682-691) — they sum to zero
    # inside every prompt group, which is what makes a SHARED prompt mask act as common random numbers for the group's gradient
    rewards = torch.rand(B, 1, generator=g) * 2.0
    adv, _, _ = grpo.group_advantages(rewards, copies)
    m.eval()
    with torch.no_grad(), m.text_model.disable_adapter():
        ref_lp = grpo.per_token_logps(m, ids, mask, comp, cmask, **mm).detach()
    m.train()
    m.text_model.lora_dropout_p = p
    return m, grpo, (ids, mask, comp, cmask, adv, ref_lp, mm, alias)


def one(m, grpo, data, seed, shared):
    ids, mask, comp, cmask, adv, ref_lp, mm, alias = data
    m.text_model.set_dropout_seed(seed * 7919 + (1 if shared else 2) * 104729)
    m.arena.zero_grad()
    if shared:
        lp = grpo.per_token_logps_shared_policy(m, ids, mask, comp, cmask, alias, **mm)
    else:
        lp = grpo.per_token_logps(m, ids, mask, comp, cmask, **mm)
    loss, _ = grpo.grpo_loss(lp, None, ref_lp, adv, cmask, 0.2, 0.2, 0.04)
    loss.backward()
    return float(loss), m.arena.grads.clone()


def worker(args):
    fixture, copies, p, seeds = args
    torch.set_num_threads(1)
    m, grpo, data = setup(fixture, copies, p)
    out = []
    for s in seeds:
        li, gi = one(m, grpo, data, s, False)
        ls, gs = one(m, grpo, data, s, True)
        out.append((s, li, gi, ls, gs))
    fam = {}
    a = m.arena
    for k in a._offsets:
        f = "dna_projection" if k.startswith("dna_projection") else ("lora_A" if k.endswith(".A") else ("lora_B" if k.endswith(".B") else "other"))
        fam.setdefault(f, []).append((a._offsets[k], a._offsets[k] + a.param(k).numel()))
    return out, fam


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=240)
    ap.add_argument("--p", type=float, default=0.2)
    ap.add_argument("--fixture", default="tiny_b")
    ap.add_argument("--copies", type=int, default=4)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import multiprocessing as mp
    chunks = [list(range(w, a.seeds, a.workers)) for w in range(a.workers)]
    with mp.get_context("spawn").Pool(a.workers) as pool:
        res = pool.map(worker, [(a.fixture, a.copies, a.p, c) for c in chunks])
    fam = res[0][1]
    rows = sorted([r for out, _ in res for r in out], key=lambda r: r[0])
    Li = torch.tensor([r[1] for r in rows], dtype=torch.float64)
    Ls = torch.tensor([r[3] for r in rows], dtype=torch.float64)
    Gi = torch.stack([r[2] for r in rows]).double()
    Gs = torch.stack([r[4] for r in rows]).double()
    N = Gi.shape[0]
    mi, ms = Gi.mean(0), Gs.mean(0)
    vi, vs = Gi.var(0, unbiased=True), Gs.var(0, unbiased=True)
    # || mean_i - mean_s ||^2 has expectation (tr Cov_i + tr Cov_s) / N when the two schemes share their expectation
    diff2 = float(((mi - ms) ** 2).sum())
    expect2 = float((vi.sum() + vs.sum()) / N)
    # its sampling spread, from a split-half bootstrap of the same statistic under the null (each scheme against ITSELF)
    gen = torch.Generator().manual_seed(0)
    null = []
    for _ in range(200):
        perm = torch.randperm(N, generator=gen)
        h1, h2 = perm[: N // 2], perm[N // 2:]
        null.append(float(((Gi[h1].mean(0) - Gi[h2].mean(0)) ** 2).sum()) / 2 + float(((Gs[h1].mean(0) - Gs[h2].mean(0)) ** 2).sum()) / 2)
    null = torch.tensor(null) / 2          # halves have N/2 samples each: E = 2 * tr Cov / (N/2) = 4 tr Cov / N; the statistic above: (tr Cov_i + tr Cov_s)/N
    rep = {"fixture": a.fixture, "copies": a.copies, "p": a.p, "seeds": N, "n_params": int(Gi.shape[1]),
           "grad_norm_mean_per_copy_masks": float(mi.norm()), "grad_norm_mean_shared_mask": float(ms.norm()),
           "mean_diff_sq": diff2, "mean_diff_sq_expected_if_equal": expect2, "mean_diff_ratio": diff2 / expect2,
           "null_ratio_p05_p50_p95": [float(torch.quantile(null, q)) / expect2 for q in (0.05, 0.5, 0.95)],
           "rel_mean_diff": float((mi - ms).norm() / mi.norm()),
           "trace_cov_per_copy": float(vi.sum()), "trace_cov_shared": float(vs.sum()), "variance_ratio_shared_over_per_copy": float(vs.sum() / vi.sum()),
           "noise_to_signal_per_copy": float(vi.sum().sqrt() / mi.norm()), "noise_to_signal_shared": float(vs.sum().sqrt() / ms.norm()),
           "loss_mean_per_copy": float(Li.mean()), "loss_mean_shared": float(Ls.mean()),
           "loss_var_ratio": float(Ls.var() / Li.var()), "loss_mean_diff_over_se": float((Li.mean() - Ls.mean()) / ((Li.var() + Ls.var()) / N).sqrt()),
           "families": {}}
    for f, spans in fam.items():
        idx = torch.cat([torch.arange(lo, hi) for lo, hi in spans])
        if float(vi[idx].sum()) == 0:
            continue
        rep["families"][f] = {"variance_ratio": float(vs[idx].sum() / vi[idx].sum()),
                              "mean_diff_ratio": float(((mi[idx] - ms[idx]) ** 2).sum() / ((vi[idx].sum() + vs[idx].sum()) / N))}
    print(json.dumps(rep, indent=1))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(rep, fh, indent=1)


if __name__ == "__main__":
    main()
