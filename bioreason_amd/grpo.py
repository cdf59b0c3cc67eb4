"""GRPO arithmetic and the training step on top of the HIP DNA-LLM (mirror of
bioreason/trainer/grpo_trainer.py: `_get_per_token_logps` :510-520, `_generate_and_score_completions` :535-749,
`compute_loss` :751-814).  The HF-Trainer scaffolding of the reference (logging, callbacks, checkpoints) is not
re-created; the numerics, the data-parallel partitioning and the two collectives are.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import ops


# ------------------------------------------------------------------------------------------- tensor math
def completion_mask(completion_ids: torch.Tensor, eos_token_id: int) -> torch.Tensor:
    """1 up to and including the first EOS (grpo_trainer.py:605-609); int32 [B, C]."""
    mask, _ = ops.eos_mask(completion_ids.to(torch.int32).contiguous(), int(eos_token_id))
    return mask


def per_token_logps(model, prompt_ids: torch.Tensor, prompt_mask: torch.Tensor, completion_ids: torch.Tensor,
                    completion_mask_: torch.Tensor, **multimodal) -> torch.Tensor:
    """log pi(completion token | prefix) for every completion position -> fp32 [B, C].

    Same quantity as `_get_per_token_logps(model, cat(prompt, completion), cat(masks))[:, P-1:]`
    (grpo_trainer.py:510-520 with the slice of :624/:640/:779), but the tied lm_head, the log-softmax and the
    gather run fused over the C kept rows only instead of materialising [B, P+C, V] logits."""
    B, P = prompt_ids.shape
    C = completion_ids.shape[1]
    ids = torch.cat([prompt_ids, completion_ids.to(prompt_ids.dtype)], dim=1)
    mask = torch.cat([prompt_mask, completion_mask_.to(prompt_mask.dtype)], dim=1)
    embeds = model._inputs_embeds(ids, multimodal.get("dna_tokenized"), multimodal.get("batch_idx_map"),
                                  multimodal.get("dna_alias"), multimodal.get("dna_enc"))
    hid = model.text_model.hidden_states(embeds, mask)                     # [B*(P+C), H]
    S = P + C
    dev = ids.device
    rows = (torch.arange(B, device=dev, dtype=torch.int32)[:, None] * S
            + torch.arange(P - 1, S - 1, device=dev, dtype=torch.int32)[None, :]).reshape(-1).contiguous()
    tgt = completion_ids.to(torch.int32).reshape(-1).contiguous()
    return model.text_model.token_logprobs(hid, rows, tgt).view(B, C)


class _GrpoLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, old_logp, ref_logp, adv, mask, eps_lo, eps_hi, beta):
        out3, dlogp = ops.grpo_loss(logp.contiguous(), old_logp, ref_logp, adv.contiguous().float(), mask.contiguous(),
                                    eps_lo, eps_hi, beta, need_grad=True)
        ctx.save_for_backward(dlogp)
        ctx.mark_non_differentiable(out3)
        return out3[0].clone(), out3

    @staticmethod
    def backward(ctx, g, _g3):
        (dlogp,) = ctx.saved_tensors
        return dlogp * g, None, None, None, None, None, None, None


def grpo_loss(logp, old_logp, ref_logp, advantages, mask, epsilon_low=0.2, epsilon_high=0.2, beta=0.04):
    """-> (loss, stats[3] = {loss, mean_kl, clip_ratio}); old_logp=None <=> num_iterations == 1 (:786)."""
    rl = ref_logp.contiguous() if (ref_logp is not None and beta != 0.0) else None
    ol = old_logp.contiguous() if old_logp is not None else None
    return _GrpoLossFn.apply(logp, ol, rl, advantages, mask.to(torch.int32), epsilon_low, epsilon_high, beta)


def group_advantages(rewards_per_func: torch.Tensor, num_generations: int, rank: int = 0, local_n: Optional[int] = None):
    """rewards_per_func: ALL-GATHERED [N, F] fp32 -> (advantages [local slice], group mean, group std) (:682-699)."""
    adv, gm, gs = ops.group_advantage(rewards_per_func.contiguous().float(), num_generations)
    if local_n is not None:
        adv = adv[rank * local_n:(rank + 1) * local_n]
    return adv, gm, gs


def repeat_sampler_indices(num_samples: int, mini_repeat_count: int, batch_size: int = 1, repeat_count: int = 1, seed: int = 0) -> List[int]:
    """the index stream of `RepeatRandomSampler` (grpo_trainer.py:72-119; class in grpo_trainer.py of this package): every
    rank draws the same permutation, each prompt index is emitted `mini_repeat_count` (= G) times consecutively; the launcher
    then shards the stream across ranks."""
    from .grpo_trainer import RepeatRandomSampler
    return list(RepeatRandomSampler(range(num_samples), mini_repeat_count, batch_size, repeat_count, seed))


def per_token_logps_shared_policy(model, prompt_ids: torch.Tensor, prompt_mask: torch.Tensor, completion_ids: torch.Tensor,
                                  completion_mask_: torch.Tensor, prompt_alias: Sequence[int], side=None, **multimodal) -> Optional[torch.Tensor]:
    """`per_token_logps` WITH gradients (the policy pass of compute_loss, grpo_trainer.py:777-779) when the rows are groups of
    consecutive copies of a prompt (RepeatRandomSampler, :107-116): the prompt rows run once per distinct prompt — forward and
    backward — and the completion rows attend to [their prompt's K / V | their own].  Same log-probs as the full-sequence pass
    (rows of a batched forward are independent; a prompt position never sees a completion); the gradients that the copies send
    into the prompt rows are summed where they enter them (the prompt's K / V rows and its last hidden state), which is what
    the chain rule gives for the sum of the copies' losses.  ~4.6x fewer rows than B x (P + C) at cfg-3.
    Deviation (stated in DESIGN.md): under LoRA dropout the shared prompt rows carry ONE mask stream for all copies of a prompt,
    the reference draws an independent mask per copy — the same marginal distribution per copy, correlated across the copies.
    Returns None when the aliases are not uniform consecutive groups (the caller then runs `per_token_logps`)."""
    from .generation import _uniform_groups
    grp = _uniform_groups(list(prompt_alias))
    if grp is None or grp[1] < 2:
        return None
    R, copies = grp
    tm = model.text_model
    B, P = prompt_ids.shape
    C = completion_ids.shape[1]
    dev = prompt_ids.device
    sel = torch.arange(R, device=dev) * copies
    embeds = model._inputs_embeds(prompt_ids, multimodal.get("dna_tokenized"), multimodal.get("batch_idx_map"),
                                  multimodal.get("dna_alias"), multimodal.get("dna_enc"))
    hid_last, hid_c = tm.hidden_states_shared(embeds.index_select(0, sel), prompt_mask.index_select(0, sel), completion_ids,
                                              completion_mask_, copies, side=side)
    from .modeling import _ExpandGroupsFn, _LogProbFn
    H = hid_c.shape[-1]
    first = _ExpandGroupsFn.apply(hid_last, copies)                                          # [B, H]: predicts completion token 0
    hsel = torch.cat([first[:, None, :], hid_c.view(B, C, H)[:, : C - 1, :]], dim=1).reshape(B * C, H).contiguous()
    tgt = completion_ids.to(torch.int32).reshape(-1).contiguous()
    return _LogProbFn.apply(hsel, tm, tgt).view(B, C)


@torch.no_grad()
def per_token_logps_shared_prefix(model, prompt_ids: torch.Tensor, prompt_mask: torch.Tensor, completion_ids: torch.Tensor,
                                  completion_mask_: torch.Tensor, prompt_alias: Sequence[int], side=None, **multimodal) -> torch.Tensor:
    """`per_token_logps` for a no-grad pass (the reference policy, grpo_trainer.py:628-640) when several rows share a
    prompt (GRPO's G copies): the prompt is run ONCE per distinct prompt with its K/V kept, then only the C completion
    tokens of every row are run against [shared prompt K/V | own completion K/V].  Rows of a batched forward are
    independent and causal attention never lets a prompt position see a completion, so the result equals the
    full-sequence pass; ~7/8 of the prompt work of a group of 8 is not repeated."""
    from . import ops
    from .engine import BF16, SeqMeta
    from .generation import KVCache

    tm = model.text_model
    eng = tm.ensure_packed()
    B, P = prompt_ids.shape
    C = completion_ids.shape[1]
    S = P + C
    dev = prompt_ids.device
    from .generation import _uniform_groups
    grp = _uniform_groups(list(prompt_alias))
    if grp is not None and grp[1] >= 2:
        # consecutive groups of equal size (RepeatRandomSampler's order): the two-segment forward of the policy pass, nothing kept
        R, copies = grp
        sel = torch.arange(R, device=dev) * copies
        embeds = model._inputs_embeds(prompt_ids, multimodal.get("dna_tokenized"), multimodal.get("batch_idx_map"), multimodal.get("dna_alias"),
                                      multimodal.get("dna_enc"))
        mp = SeqMeta(B=R, S=P, pos=torch.arange(P, dtype=torch.int32, device=dev).repeat(R),
                     kmask=prompt_mask.index_select(0, sel).to(torch.uint8).contiguous(), lora_on=tm._lora_enabled, max_pos=S)
        kfull = torch.cat([prompt_mask, completion_mask_.to(prompt_mask.dtype)], dim=1).to(torch.uint8).contiguous()
        mc = SeqMeta(B=B, S=C, pos=(torch.arange(C, dtype=torch.int32, device=dev) + P).repeat(B), kmask=kfull,
                     lora_on=tm._lora_enabled, max_pos=S)
        ids32 = completion_ids.to(torch.int32).reshape(-1).contiguous()
        xc = torch.empty((B * C, eng.H), dtype=BF16, device=dev)
        ops.embed_scatter_fwd(ids32, None, eng.E, None, xc)
        xp = embeds.index_select(0, sel).reshape(R * P, -1).to(BF16).contiguous()
        hid_last, hid_c, _ = eng.forward_hidden_shared(xp, mp, xc, mc, copies, save=False, side=side)
        first = hid_last.repeat_interleave(copies, dim=0)
        hsel = torch.cat([first[:, None, :], hid_c.view(B, C, -1)[:, : C - 1, :]], dim=1).reshape(B * C, -1).contiguous()
        logp, _ = ops.lmhead_logprob(hsel, eng.E, ids32)
        return logp.view(B, C)
    reps = sorted(set(prompt_alias))
    where = {r: i for i, r in enumerate(reps)}
    sel = torch.tensor(reps, device=dev)
    gmap = torch.tensor([where[a] for a in prompt_alias], device=dev)
    R = len(reps)
    embeds = model._inputs_embeds(prompt_ids, multimodal.get("dna_tokenized"), multimodal.get("batch_idx_map"), multimodal.get("dna_alias"),
                                  multimodal.get("dna_enc"))
    # (1) distinct prompts, K/V captured (positions = arange, as Qwen3Model.forward assigns them, TF:qwen3:391-394)
    cache_r = KVCache(eng, R, S, dev)
    meta = SeqMeta(B=R, S=P, pos=torch.arange(P, dtype=torch.int32, device=dev).repeat(R),
                   kmask=prompt_mask[sel].to(torch.uint8).contiguous(), lora_on=tm._lora_enabled, max_pos=S)
    x = embeds[sel].reshape(R * P, -1).to(BF16).contiguous()
    for li in range(eng.L):
        x, _ = eng.layer_fwd(li, x, meta, save=False, kv_out=(cache_r.k[li], cache_r.v[li], 0))
    last = torch.arange(R, device=dev, dtype=torch.int32) * P + (P - 1)
    hid_last = ops.rmsnorm_fwd(ops.gather_rows(last, x), eng.norm_w, eng.eps).index_select(0, gmap)      # [B, H]
    # (2) every row gets its prompt's K/V, (3) completion tokens only
    kfull = torch.cat([prompt_mask, completion_mask_.to(prompt_mask.dtype)], dim=1).to(torch.uint8).contiguous()
    meta2 = SeqMeta(B=B, S=C, pos=(torch.arange(C, dtype=torch.int32, device=dev) + P).repeat(B), kmask=kfull,
                    lora_on=tm._lora_enabled, max_pos=S)
    ids32 = completion_ids.to(torch.int32).reshape(-1).contiguous()
    x2 = torch.empty((B * C, eng.H), dtype=BF16, device=dev)
    ops.embed_scatter_fwd(ids32, None, eng.E, None, x2)
    for li in range(eng.L):
        kc = cache_r.k[li].index_select(0, gmap)
        vc = cache_r.v[li].index_select(0, gmap)
        x2, _ = eng.layer_fwd(li, x2, meta2, save=False, kv_out=(kc, vc, P))
    hid2 = ops.rmsnorm_fwd(x2, eng.norm_w, eng.eps).view(B, C, -1)
    hsel = torch.cat([hid_last[:, None, :], hid2[:, : C - 1, :]], dim=1).reshape(B * C, -1).contiguous()
    logp, _ = ops.lmhead_logprob(hsel, eng.E, ids32)
    return logp.view(B, C)
