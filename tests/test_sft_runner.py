"""The SFT step around the model (train_dna_qwen.py:179-213, 393-411, 985-1005): cosine-with-warm-up learning rate, gradient
accumulation (`accumulate_grad_batches`), clip at 1.0 — `bioreason_amd.trainer.SFTStepRunner`."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_model_parity import GOLD, build, rel, to_dev   # noqa: E402
from test_oracle import rebuild                          # noqa: E402


@pytest.mark.parametrize("total", [37, 10, 3, 1])
def test_cosine_schedule_equals_the_installed_scheduler(total):
    """the reference's `configure_optimizers` (train_dna_qwen.py:393-411): rate of optimiser step i = what the installed
    `get_cosine_schedule_with_warmup(opt, int(0.1 total), total)` has in force before its i-th `step()`"""
    from transformers import get_cosine_schedule_with_warmup
    from bioreason_amd.trainer import cosine_schedule_with_warmup
    base = 3e-4
    p = torch.zeros(1, requires_grad=True)
    opt = torch.optim.AdamW([p], lr=base)
    sched = get_cosine_schedule_with_warmup(opt, num_warmup_steps=int(0.1 * total), num_training_steps=total)
    mine = cosine_schedule_with_warmup(base, total)
    for i in range(total + 2):
        assert abs(opt.param_groups[0]["lr"] - mine(i)) <= 1e-12 + 1e-9 * base, (i, opt.param_groups[0]["lr"], mine(i))
        opt.step()
        sched.step()


def test_accumulated_gradients_and_schedule(backend):
    """two micro-batches under gradient_accumulation_steps = 2: the gradient handed to the optimiser equals the oracle's
    (loss_1 / 2 + loss_2 / 2).backward(); one optimiser step per cycle, at the scheduled rate"""
    from bioreason_amd.trainer import SFTStepRunner, cosine_schedule_with_warmup
    fix = torch.load(os.path.join(GOLD, "tiny_b.pt"), weights_only=False)
    ora = rebuild(fix, True)
    m = build(fix, backend, True)
    m.train()
    b1 = fix["batch"]
    b2 = {k: (v.flip(0) if isinstance(v, torch.Tensor) else v) for k, v in b1.items()}
    n_per = [b1["batch_idx_map"].count(i) for i in range(b1["input_ids"].shape[0])]
    assert len(set(n_per)) == 1                        # equal DNA sequences per sample: flipping the rows flips whole blocks
    k = n_per[0]
    B = b1["input_ids"].shape[0]
    order = [j for i in reversed(range(B)) for j in range(i * k, (i + 1) * k)]
    b2["dna_tokenized"] = {kk: v[order] for kk, v in b1["dna_tokenized"].items()}
    b2["batch_idx_map"] = list(b1["batch_idx_map"])
    b2["labels"] = b2["labels"].clone()
    b2["labels"][:, -3:] = -100                        # a different number of supervised positions in the second micro-batch
    # ---- oracle
    ora.zero_grad(set_to_none=True)
    for bb in (b1, b2):
        out = ora(input_ids=bb["input_ids"], attention_mask=bb["attention_mask"], labels=bb["labels"], dna_tokenized=bb["dna_tokenized"],
                  batch_idx_map=bb["batch_idx_map"])
        (out.loss / 2).backward()
    want = {n: p.grad.detach().clone() for n, p in ora.named_parameters() if p.grad is not None}
    # ---- runner
    sched = cosine_schedule_with_warmup(1e-3, 20)
    runner = SFTStepRunner(m, learning_rate=1e-3, weight_decay=0.0, gradient_accumulation_steps=2, lr_schedule=sched)
    seen = []
    orig = m.arena.adamw_step

    def spy(lr, *a, **kw):
        seen.append((lr, {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}))
        return orig(lr, *a, **kw)
    m.arena.adamw_step = spy
    try:
        r1 = runner.step(to_dev(b1, backend))
        assert not seen and "stepped" not in r1 and runner.global_step == 0
        r2 = runner.step(to_dev(b2, backend))
        assert len(seen) == 1 and r2.get("stepped") and runner.global_step == 1
        runner.step(to_dev(b1, backend))
        runner.step(to_dev(b2, backend))
        runner.step(to_dev(b1, backend))
        runner.step(to_dev(b2, backend))
    finally:
        m.arena.adamw_step = orig
    assert [round(lr / 1e-3, 9) for lr, _ in seen] == [round(sched(i) / 1e-3, 9) for i in range(3)] and seen[0][0] == 0.0 and seen[1][0] > 0
    got = seen[0][1]
    checked = 0
    for n, g in got.items():
        key = n if n in want else n.replace("text_model.", "text_model.base_model.model.", 1)
        cands = [w for w in want if w.endswith(n.split("text_model.", 1)[-1])] if key not in want else [key]
        if not cands:
            continue
        w = want[cands[0]]
        if w.shape != g.shape or float(w.norm()) == 0.0:
            continue
        assert rel(g, w) < 3e-2, (n, rel(g, w))
        checked += 1
    assert checked >= 8, checked
