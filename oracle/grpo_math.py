"""CPU ORACLE (test infrastructure) — the GRPO arithmetic of bioreason/trainer/grpo_trainer.py restated as
free functions (the trainer class itself cannot be imported: trl / peft / deepspeed are absent).

  per_token_logps        grpo_trainer.py:510-520
  completion_mask        grpo_trainer.py:605-609
  group_advantages       grpo_trainer.py:682-699
  grpo_loss              grpo_trainer.py:786-814
  repeat_sampler_indices grpo_trainer.py:72-119 (RepeatRandomSampler)
  sample_next_token      HF warpers as configured at grpo_trainer.py:384-391
                         (TF:generation/logits_process.py:238 temperature, :473 top-p, :542 top-k)

PINNED (round 3): tests/test_oracle_pinned.py holds every function above bit-equal to the reference's OWN statements — the
cited lines are ast-extracted from grpo_trainer.py and executed unmodified against a stub `self` (oracle/ref_exec.py; whole
`compute_loss` incl. its buffering branch and `_get_per_token_logps`, the EOS-mask and advantage statements of
`_generate_and_score_completions`) on seeded inputs incl. num_iterations > 1, beta = 0, a row masked after its first token, a
zero-std group and a group spanning two ranks — and `warp_probs` equal to softmax(TemperatureLogitsWarper -> TopKLogitsWarper ->
TopPLogitsWarper) of the installed transformers (ties at the k-th value, top_p boundary, k > support).  The outputs of the
reference statements are committed as tests/golden/grpo_ref.pt (oracle/make_grpo_golden.py), so the pin also holds on the GPU
box, where /root/reference does not exist.
"""
from __future__ import annotations

from typing import List, Optional

import torch


def per_token_logps(model, input_ids, attention_mask, **multimodal):
    logits = model(input_ids=input_ids, attention_mask=attention_mask, **multimodal).logits
    logits = logits[:, :-1, :]                       # the last position predicts a token we do not have
    targets = input_ids[:, 1:]
    rows = []
    for lg, tg in zip(logits, targets):              # row loop as in the reference (memory peak)
        rows.append(torch.gather(lg.log_softmax(dim=-1), 1, tg.unsqueeze(1)).squeeze(1))
    return torch.stack(rows)


def completion_mask(completion_ids: torch.Tensor, eos_token_id: int) -> torch.Tensor:
    is_eos = completion_ids == eos_token_id
    B, C = completion_ids.shape
    first = torch.full((B,), C, dtype=torch.long, device=completion_ids.device)
    has = is_eos.any(dim=1)
    first[has] = is_eos.int().argmax(dim=1)[has]
    idx = torch.arange(C, device=completion_ids.device).expand(B, -1)
    return (idx <= first.unsqueeze(1)).int()


def group_advantages(rewards_per_func: torch.Tensor, num_generations: int, process_index: int = 0, local_n: Optional[int] = None):
    """rewards_per_func: the ALL-GATHERED [N, F] matrix; returns (advantages for the local slice, mean, std)."""
    rewards = rewards_per_func.sum(dim=1)
    grouped = rewards.view(-1, num_generations)
    mean = grouped.mean(dim=1).repeat_interleave(num_generations, dim=0)
    std = grouped.std(dim=1).repeat_interleave(num_generations, dim=0)
    adv = (rewards - mean) / (std + 1e-4)
    if local_n is not None:
        adv = adv[process_index * local_n:(process_index + 1) * local_n]
    return adv, mean, std


def grpo_loss(per_token_logps_, old_per_token_logps, ref_per_token_logps, advantages, completion_mask_, epsilon_low=0.2,
              epsilon_high=0.2, beta=0.04):
    """-> (loss, mean_kl or None, clip_ratio).  old_per_token_logps=None means num_iterations == 1 (:786)."""
    old = per_token_logps_.detach() if old_per_token_logps is None else old_per_token_logps
    coef_1 = torch.exp(per_token_logps_ - old)
    coef_2 = torch.clamp(coef_1, 1 - epsilon_low, 1 + epsilon_high)
    l1 = coef_1 * advantages.unsqueeze(1)
    l2 = coef_2 * advantages.unsqueeze(1)
    per_token_loss = -torch.min(l1, l2)
    mean_kl = None
    m = completion_mask_
    if beta > 0:
        d = ref_per_token_logps - per_token_logps_
        kl = torch.exp(d) - d - 1
        per_token_loss = per_token_loss + beta * kl
        mean_kl = ((kl * m).sum(dim=1) / m.sum(dim=1)).mean()
    loss = ((per_token_loss * m).sum(dim=1) / m.sum(dim=1)).mean()
    clip_ratio = ((l1 < l2).float() * m).sum() / m.sum()
    return loss, mean_kl, clip_ratio


def repeat_sampler_indices(num_samples: int, mini_repeat_count: int, batch_size: int = 1, repeat_count: int = 1, seed: int = 0) -> List[int]:
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(num_samples, generator=g).tolist()
    chunks = [perm[i:i + batch_size] for i in range(0, len(perm), batch_size)]
    chunks = [c for c in chunks if len(c) == batch_size]
    out: List[int] = []
    for chunk in chunks:
        for _ in range(repeat_count):
            for index in chunk:
                out.extend([index] * mini_repeat_count)
    return out


def warp_probs(logits: torch.Tensor, temperature: float, top_k: int, top_p: float) -> torch.Tensor:
    """Distribution HF samples from after temperature -> top-k -> top-p (order of GenerationMixin._get_logits_processor)."""
    scores = logits.float() / temperature
    if top_k and top_k > 0:
        kth = torch.topk(scores, min(top_k, scores.shape[-1]))[0][..., -1, None]
        scores = scores.masked_fill(scores < kth, float("-inf"))
    if top_p < 1.0:
        sorted_logits, sorted_idx = torch.sort(scores, descending=False)
        cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove = cum <= (1 - top_p)
        remove[..., -1:] = False
        remove = remove.scatter(-1, sorted_idx, remove)
        scores = scores.masked_fill(remove, float("-inf"))
    return torch.softmax(scores, dim=-1)
