"""The differentiable shared-prompt policy pass (grpo.per_token_logps_shared_policy: the G rollouts of a GRPO group run their
prompt ONCE, forward and backward) against the full-sequence pass it replaces (grpo.per_token_logps = `_get_per_token_logps`,
grpo_trainer.py:510-520 with the slice of :779): same log-probs, same gradients of every trainable parameter (LoRA A / B of all
layers, dna_projection) for a loss with random per-token weights — the chain rule through the shared prompt K / V rows is a sum
over the copies.  Under LoRA dropout the two passes are compared with the oracle under INJECTED masks: the shared prompt rows carry
one mask stream (all copies of a prompt see it), the completion rows their own."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_model_parity import GOLD, build, rel, to_dev      # noqa: E402


def _group_batch(fix, dev, copies):
    b = to_dev(fix["batch"], dev)
    b.pop("labels")
    nb = b["input_ids"].shape[0]
    bmap0 = b["batch_idx_map"]
    rows = [r for r in range(nb) for _ in range(copies)]
    ids, mask = b["input_ids"][rows], b["attention_mask"][rows]
    dsel, bmap = [], []
    for j, r in enumerate(rows):
        for i, s in enumerate(bmap0):
            if s == r:
                dsel.append(i)
                bmap.append(j)
    dna = {k: v[dsel] for k, v in b["dna_tokenized"].items()}
    alias = [(j // copies) * copies for j in range(len(rows))]
    return ids, mask, {"dna_tokenized": dna, "batch_idx_map": bmap}, alias


@pytest.mark.parametrize("name,copies", [("tiny_a", 2), ("tiny_b", 3)])
def test_shared_policy_pass_equals_full_pass(backend, name, copies):
    from bioreason_amd import grpo
    fix = torch.load(os.path.join(GOLD, f"{name}.pt"), weights_only=False)
    m = build(fix, backend, True)
    m.train()
    ids, mask, mm, alias = _group_batch(fix, backend, copies)
    B = ids.shape[0]
    C = 7
    g = torch.Generator().manual_seed(11)
    comp = torch.randint(3, fix["config"]["text"]["vocab_size"] - 8, (B, C), generator=g).to(backend)
    cmask = torch.ones((B, C), dtype=torch.int32, device=backend)
    cmask[1, 4:] = 0
    w = torch.randn(B, C, generator=g).to(backend)

    def run(fn):
        m.arena.zero_grad()
        lp = fn()
        (lp * w * cmask).sum().backward()
        return lp.detach().clone(), m.arena.grads.clone()

    lp_full, g_full = run(lambda: grpo.per_token_logps(m, ids, mask, comp, cmask, **mm))
    lp_sh, g_sh = run(lambda: grpo.per_token_logps_shared_policy(m, ids, mask, comp, cmask, alias, **mm))
    keep = cmask.bool().cpu()
    assert (lp_full.cpu()[keep] - lp_sh.cpu()[keep]).abs().max() < 0.06                 # bf16 activations: rounding placed differently
    assert rel(lp_sh.cpu()[keep], lp_full.cpu()[keep]) < 1e-2
    # every trainable parameter's gradient: LoRA A / B of every layer and target, dna_projection weight / bias
    assert g_full.abs().max() > 0
    assert rel(g_sh, g_full) < 3e-2, rel(g_sh, g_full)
    for key in ("dna_projection.weight", "dna_projection.bias"):
        m.arena.grads.copy_(g_full)
        a = m.arena.grad(key).clone()
        m.arena.grads.copy_(g_sh)
        b_ = m.arena.grad(key).clone()
        assert a.abs().max() > 0 and rel(b_, a) < 5e-2, key
    # groups that are not uniform consecutive copies: the caller falls back to the full pass
    assert grpo.per_token_logps_shared_policy(m, ids, mask, comp, cmask, list(range(B)), **mm) is None


def test_group_sum_kernel(backend):
    from bioreason_amd import ops
    g = torch.Generator().manual_seed(0)
    big = torch.randn(6, 10, 4, 8, generator=g).to(torch.bfloat16).to(backend)
    src = big[:, :7]                                     # strided over dim 0, as dK[:, :P] of a [B, P + C, Hkv, hd] tensor is
    add = torch.randn(2, 7, 4, 8, generator=g).to(torch.bfloat16).to(backend)
    want = (src.float().view(2, 3, 7, 4, 8).sum(1) + add.float()).to(torch.bfloat16)
    assert torch.equal(ops.group_sum(src, 3, add).cpu(), want.cpu())
    assert torch.equal(ops.group_sum(src, 3).cpu(), src.float().view(2, 3, 7, 4, 8).sum(1).to(torch.bfloat16).cpu())


def test_group_broadcast_kernel(backend):
    from bioreason_amd import ops
    g = torch.Generator().manual_seed(1)
    src = torch.randn(2, 3, 5, 8, generator=g).to(torch.bfloat16).to(backend)
    out = torch.full((6, 3, 9, 8), 7.0, dtype=torch.bfloat16, device=backend)
    ops.group_broadcast(src, out, 3)
    want = torch.full((6, 3, 9, 8), 7.0, dtype=torch.bfloat16)
    want.view(2, 3, 3, 9, 8)[:, :, :, :5] = src.cpu()[:, None]
    assert torch.equal(out.cpu(), want)


def test_runner_uses_the_shared_policy_pass(backend):
    """GRPOStepRunner.compute_loss takes the shared pass for grouped batches (cfg.share_policy_prompt) and the step still trains"""
    from bioreason_amd.trainer import GRPOConfig, GRPOStepRunner
    from bioreason_amd import grpo
    fix = torch.load(os.path.join(GOLD, "tiny_a.pt"), weights_only=False)
    m = build(fix, backend, True)
    ids, mask, mm, alias = _group_batch(fix, backend, 2)
    batch = {"input_ids": ids, "attention_mask": mask, "dna_tokenized": mm["dna_tokenized"], "batch_idx_map": mm["batch_idx_map"],
             "prompt_alias": alias}
    calls = {"n": 0}
    orig = grpo.per_token_logps_shared_policy

    def spy(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    grpo.per_token_logps_shared_policy = spy
    try:
        runner = GRPOStepRunner(m, GRPOConfig(num_generations=2, max_completion_length=4, eos_token_id=None, seed=3, learning_rate=1e-3))
        p0 = m.arena.params.clone()
        out = runner.step(batch)
        assert calls["n"] == 1 and torch.isfinite(out["loss_t"]).all() and (m.arena.params - p0).abs().max() > 0
        runner2 = GRPOStepRunner(m, GRPOConfig(num_generations=2, max_completion_length=4, eos_token_id=None, seed=3, learning_rate=1e-3,
                                               share_policy_prompt=False))
        runner2.step(batch)
        assert calls["n"] == 1
        # default (None): the adapters' dropout decides.  With lora_dropout > 0 the shared rows would carry one mask stream for all
        # copies of a prompt, which is not how the reference draws them (grpo_trainer.py:777-779) -> the full-row pass, unless opted in
        m.text_model.lora_dropout_p = 0.05
        runner3 = GRPOStepRunner(m, GRPOConfig(num_generations=2, max_completion_length=4, eos_token_id=None, seed=3, learning_rate=1e-3))
        assert runner3.cfg.share_policy_prompt is None and not runner3.shares_policy_prompt()
        runner3.step(batch)
        assert calls["n"] == 1
        runner4 = GRPOStepRunner(m, GRPOConfig(num_generations=2, max_completion_length=4, eos_token_id=None, seed=3, learning_rate=1e-3,
                                               share_policy_prompt=True))
        assert runner4.shares_policy_prompt()
        runner4.step(batch)
        assert calls["n"] == 2
        m.text_model.lora_dropout_p = 0.0
        assert runner3.shares_policy_prompt()
    finally:
        grpo.per_token_logps_shared_policy = orig


class _FixedMask(torch.nn.Module):
    """stands in for a LoraLayer's nn.Dropout with a given keep mask: x * mask / (1 - p)"""

    def __init__(self, mask, p):
        super().__init__()
        self.mask, self.p = mask, p

    def forward(self, x):
        return x * self.mask.to(x.dtype).view(x.shape) / (1.0 - self.p)


def test_shared_policy_pass_under_dropout_matches_oracle_with_injected_masks(backend):
    """training mode, lora_dropout > 0: the oracle (grpo_math.per_token_logps on the full [B, P + C] rows, reference arithmetic)
    gets, for every LoRA target, the keep mask the HIP pass regenerates — the prompt rows of ALL copies of a prompt read the
    shared segment's mask rows, the completion rows their own segment's — and must then give the same log-probs and LoRA /
    projection gradients.  This pins the semantics of the one stated deviation: which rows share a mask."""
    from bioreason_amd import grpo, ops
    from bioreason_amd.engine import lora_drop_seeds
    from oracle import dna_llm_oracle as O
    from oracle import grpo_math as GM
    from test_oracle import rebuild
    p, copies, C = 0.2, 2, 5
    fix = torch.load(os.path.join(GOLD, "tiny_a.pt"), weights_only=False)
    m = build(fix, backend, False)
    m.text_model.apply_lora(r=32, alpha=64.0, dropout=p, arena=m.arena)
    own = dict(m.text_model.named_parameters())
    for k, v in fix["state"]["lora"].items():
        own[k].data.copy_(v.to(backend))
    m.arena.pack()
    m.train()
    m.text_model.set_dropout_seed(77)
    seed_p = (77 * 0x9E3779B1 + 1 * 0x85EBCA6B) & 0xFFFFFFFF               # first pass after set_dropout_seed
    seed_c = (seed_p * 0x2C1B3C6D + 0x5BD1E995) & 0xFFFFFFFF
    ids, mask, mm, alias = _group_batch(fix, backend, copies)
    B, P = ids.shape
    R = B // copies
    g = torch.Generator().manual_seed(4)
    comp = torch.randint(3, fix["config"]["text"]["vocab_size"] - 8, (B, C), generator=g)
    cmask = torch.ones((B, C), dtype=torch.int32)
    w = torch.randn(B, C, generator=g)
    m.arena.zero_grad()
    lp = grpo.per_token_logps_shared_policy(m, ids, mask, comp.to(backend), cmask.to(backend), alias, **mm)
    (lp * w.to(backend)).sum().backward()
    # ---- the oracle on the full rows with the same masks
    ora = rebuild(fix, True).train()
    where = {"q_proj": ("qkv", 0, 3), "k_proj": ("qkv", 1, 3), "v_proj": ("qkv", 2, 3), "o_proj": ("o", 0, 1),
             "gate_proj": ("gu", 0, 2), "up_proj": ("gu", 1, 2), "down_proj": ("d", 0, 1)}
    for li, layer in enumerate(ora.text_model.model.layers):
        for holder in (layer.self_attn, layer.mlp):
            for nm, (grp, j, n) in where.items():
                mod = getattr(holder, nm, None)
                if isinstance(mod, O.LoraLinear):
                    K = mod.base_layer.in_features
                    mp_ = ops.dropout_mask(R * P, K, p, lora_drop_seeds(seed_p, li, grp, n)[j], backend).cpu().view(R, P, K)
                    mc_ = ops.dropout_mask(B * C, K, p, lora_drop_seeds(seed_c, li, grp, n)[j], backend).cpu().view(B, C, K)
                    full = torch.cat([mp_.repeat_interleave(copies, dim=0), mc_], dim=1)                # [B, P + C, K]
                    mod.dropout = _FixedMask(full, p)
    ids_c, mask_c = ids.cpu(), mask.cpu()
    mmc = {"dna_tokenized": {k: v.cpu() for k, v in mm["dna_tokenized"].items()}, "batch_idx_map": mm["batch_idx_map"]}
    full_ids = torch.cat([ids_c, comp], dim=1)
    full_mask = torch.cat([mask_c, cmask.to(mask_c.dtype)], dim=1)
    want = GM.per_token_logps(ora, full_ids, full_mask, **mmc)[:, P - 1:]
    (want * w).sum().backward()
    tol = 2.5e-2
    assert rel(lp.detach().cpu(), want.detach()) < tol
    assert rel(m.dna_projection.weight.grad, ora.dna_projection.weight.grad) < 3 * tol
    l0, r0 = m.text_model.model.layers[0], ora.text_model.model.layers[0]
    l1, r1 = m.text_model.model.layers[-1], ora.text_model.model.layers[-1]
    for nm, mod, ref in (("q0", l0.self_attn.q_proj, r0.self_attn.q_proj), ("v0", l0.self_attn.v_proj, r0.self_attn.v_proj),
                         ("up0", l0.mlp.up_proj, r0.mlp.up_proj), ("down0", l0.mlp.down_proj, r0.mlp.down_proj),
                         ("k1", l1.self_attn.k_proj, r1.self_attn.k_proj), ("o1", l1.self_attn.o_proj, r1.self_attn.o_proj)):
        assert rel(mod.lora_A["default"].weight.grad, ref.lora_A["default"].weight.grad) < 3 * tol, nm
        assert rel(mod.lora_B["default"].weight.grad, ref.lora_B["default"].weight.grad) < 3 * tol, nm


@pytest.mark.gpu
def test_ref_pass_on_side_stream_equals_main_stream(hip_device):
    """GRPOConfig.overlap_ref_pass: the reference-policy pass issued on a second HIP stream (beside the policy forward) gives the
    same bits as on the main stream, and the step that consumes it equals the step without the overlap (forward quantities are
    deterministic: no atomics on that path)"""
    from bioreason_amd.trainer import GRPOConfig, GRPOStepRunner
    dev = hip_device
    fix = torch.load(os.path.join(GOLD, "tiny_a.pt"), weights_only=False)
    ids, mask, mm, alias = _group_batch(fix, dev, 2)
    batch = {"input_ids": ids, "attention_mask": mask, "dna_tokenized": mm["dna_tokenized"], "batch_idx_map": mm["batch_idx_map"],
             "prompt_alias": alias}
    outs = []
    for overlap in (True, False):
        m = build(fix, dev, True)
        runner = GRPOStepRunner(m, GRPOConfig(num_generations=2, max_completion_length=6, eos_token_id=None, seed=3, learning_rate=1e-3,
                                              overlap_ref_pass=overlap))
        inputs = runner.generate_and_score(batch, defer_ref_join=True)
        if overlap:
            assert inputs["ref_join"] is not None
            torch.cuda.current_stream(dev).wait_stream(inputs["ref_join"])
        else:
            assert inputs["ref_join"] is None
        torch.cuda.synchronize()
        loss, stats = runner.compute_loss(dict(inputs, ref_join=None))
        outs.append((inputs["completion_ids"].clone(), inputs["ref_per_token_logps"].clone(), loss.detach().clone(), stats.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1]), (outs[0][1] - outs[1][1]).abs().max()
    assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][3], outs[1][3])


@pytest.mark.gpu
def test_rollout_weights_on_side_stream_give_the_same_rollout(hip_device):
    """GRPOConfig.overlap_rollout_weights: the merged / packed weight set of the rollout built on a side stream beside the DNA
    encoder (twice in a row: the second build replaces the first set) — same sampled tokens, same reference log-probs as with the
    set built on the main stream inside generate()"""
    from bioreason_amd.trainer import GRPOConfig, GRPOStepRunner
    dev = hip_device
    fix = torch.load(os.path.join(GOLD, "tiny_a.pt"), weights_only=False)
    ids, mask, mm, alias = _group_batch(fix, dev, 2)
    batch = {"input_ids": ids, "attention_mask": mask, "dna_tokenized": mm["dna_tokenized"], "batch_idx_map": mm["batch_idx_map"],
             "prompt_alias": alias}
    outs = []
    for overlap in (True, False):
        m = build(fix, dev, True)
        runner = GRPOStepRunner(m, GRPOConfig(num_generations=2, max_completion_length=6, eos_token_id=None, seed=3, learning_rate=1e-3,
                                              overlap_rollout_weights=overlap, overlap_ref_pass=False))
        got = []
        for _ in range(2):                       # the second step rebuilds the set from the updated adapters
            out = runner.step(batch)
            inputs = runner._buffered_inputs[0]
            got.append((inputs["completion_ids"].clone(), inputs["ref_per_token_logps"].clone(), out["loss_t"].clone()))
        torch.cuda.synchronize()
        outs.append(got)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(outs[0][0][2], outs[1][0][2])          # first step's loss: identical inputs, no atomics on the forward path


@pytest.mark.gpu
@pytest.mark.parametrize("name,copies", [("tiny_a", 2), ("tiny_b", 3)])
def test_two_stream_chains_equal_one_stream(hip_device, name, copies):
    """the prompt chain and the completion chain of the shared policy pass on two HIP streams (engine.forward_hidden_shared
    `side`): log-probs bit-identical to the one-stream pass, gradients equal up to the order of the fp32 atomics of the LoRA
    weight gradients (both chains add into the same arena)"""
    from bioreason_amd import grpo
    dev = hip_device
    fix = torch.load(os.path.join(GOLD, f"{name}.pt"), weights_only=False)
    m = build(fix, dev, True)
    m.train()
    ids, mask, mm, alias = _group_batch(fix, dev, copies)
    B, C = ids.shape[0], 9
    g = torch.Generator().manual_seed(2)
    comp = torch.randint(3, fix["config"]["text"]["vocab_size"] - 8, (B, C), generator=g).to(dev)
    cmask = torch.ones((B, C), dtype=torch.int32, device=dev)
    w = torch.randn(B, C, generator=g).to(dev)
    side = torch.cuda.Stream(device=dev)
    res = []
    for s_ in (None, side, side, None):
        m.arena.zero_grad()
        lp = grpo.per_token_logps_shared_policy(m, ids, mask, comp, cmask, alias, side=s_, **mm)
        (lp * w).sum().backward()
        torch.cuda.synchronize()
        res.append((lp.detach().clone(), m.arena.grads.clone()))
    for lp, gr in res[1:]:
        assert torch.equal(lp, res[0][0])
        assert rel(gr, res[0][1]) < 1e-4, rel(gr, res[0][1])
