"""One GRPO training step on the HIP DNA-LLM, data-parallel over the GPUs of a node.

Mirrors DNALLMGRPOTrainer.compute_loss / _generate_and_score_completions (grpo_trainer.py:535-814) for
num_iterations == 1 (the reference's default):
    rollout (generate)  ->  EOS mask  ->  reference log-probs (adapters disabled, :636-640)  ->  rewards
    ->  all-gather of rewards over ranks (:679)  ->  group advantages, local slice (:682-699)
    ->  policy log-probs (:777-779)  ->  clipped objective + beta*KL (:786-807)  ->  backward
    ->  gradient all-reduce  ->  AdamW (+ grad clip).
Parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).  The path has exactly two
exchange steps, both on tiny or single-bucket payloads: the reward all-gather (one packed [B_local, F] fp32
message) and ONE all-reduce over the flat gradient arena (≈148 MB fp32 for Qwen3-1.7B r=32).  No model sharding:
NT-500M + Qwen3-1.7B + KV cache + activations use < 60 GB of the 288 GB HBM.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist

from . import grpo


@dataclass
class GRPOConfig:
    """the fields of DNALLMGRPOConfig (grpo_config.py:146-365) that enter the arithmetic"""
    num_generations: int = 8
    max_completion_length: int = 256
    temperature: float = 0.6
    top_p: float = 0.95
    top_k: int = 20
    beta: float = 0.04
    epsilon: float = 0.2
    epsilon_high: Optional[float] = None
    learning_rate: float = 1e-5
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 1.0
    eos_token_id: Optional[int] = None
    pad_token_id: Optional[int] = None
    seed: int = 42
    # execution knobs of the rollout (no effect on the arithmetic): hipGraph replay of the token loop, and decode
    # attention over one shared copy of each prompt's K/V
    rollout_graph: Optional[bool] = None
    rollout_shared_prefix: bool = True


def token_stat_rewards(completion_ids: torch.Tensor, completion_mask: torch.Tensor) -> torch.Tensor:
    """Stand-in reward functions for synthetic runs (the reference's are CPU regexes over decoded text,
    reason.py:193-230, and there is no tokenizer offline): two cheap statistics of the sampled ids -> [B, 2] fp32."""
    m = completion_mask.float()
    n = m.sum(1).clamp(min=1)
    r0 = ((completion_ids % 7 == 0).float() * m).sum(1) / n * 2.0
    r1 = ((completion_ids % 2 == 0).float() * m).sum(1) / n * 0.5
    return torch.stack([r0, r1], dim=1)


class GRPOStepRunner:
    def __init__(self, model, cfg: GRPOConfig, reward_fn: Callable = token_stat_rewards):
        self.model, self.cfg, self.reward_fn = model, cfg, reward_fn
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.step_idx = 0
        if hasattr(model.text_model, "set_dropout_seed"):          # every rank draws its own LoRA dropout masks, as separate
            model.text_model.set_dropout_seed(cfg.seed * 1000003 + self.rank)   # processes with their own RNG streams do
        self.timers: Dict[str, float] = {}
        self.rollout_profile: Dict[str, float] = {}      # filled by step(timing=True): phases inside generate(), ms

    def step(self, batch: Dict, timing: bool = False) -> Dict[str, float]:
        m, c = self.model, self.cfg
        dev = batch["input_ids"].device
        ev = None
        marks: List = []

        def mark(name):
            if timing and dev.type == "cuda":
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append((name, e))

        if timing:
            self.rollout_profile.clear()
        mm = {"dna_tokenized": batch["dna_tokenized"], "batch_idx_map": batch["batch_idx_map"], "dna_alias": batch.get("dna_alias")}
        prompt_ids, prompt_mask = batch["input_ids"], batch["attention_mask"]
        B = prompt_ids.shape[0]
        mark("start")
        # ---- rollout (unwrapped_model.generate, :579-596): only completion ids come back
        completion_ids = m.generate(input_ids=prompt_ids, attention_mask=prompt_mask, **mm,
                                    max_new_tokens=c.max_completion_length, do_sample=True, temperature=c.temperature,
                                    top_k=c.top_k, top_p=c.top_p, eos_token_id=c.eos_token_id, pad_token_id=c.pad_token_id,
                                    seed=c.seed + 1000003 * self.step_idx + self.rank, return_full_length=True,
                                    prompt_alias=batch.get("prompt_alias"), use_graph=c.rollout_graph,
                                    shared_prefix_decode=c.rollout_shared_prefix,
                                    profile=self.rollout_profile if timing else None)
        mark("rollout")
        if c.eos_token_id is not None:
            cmask = grpo.completion_mask(completion_ids, c.eos_token_id)
        else:
            cmask = torch.ones(completion_ids.shape, dtype=torch.int32, device=dev)
        # ---- reference policy = the same network with adapters disabled (:636-640)
        ref_lp = None
        if c.beta != 0.0:
            with torch.no_grad(), m.text_model.disable_adapter():
                if batch.get("prompt_alias") is not None:
                    ref_lp = grpo.per_token_logps_shared_prefix(m, prompt_ids, prompt_mask, completion_ids, cmask,
                                                                batch["prompt_alias"], **mm)
                else:
                    ref_lp = grpo.per_token_logps(m, prompt_ids, prompt_mask, completion_ids, cmask, **mm)
        mark("ref_logps")
        # ---- rewards, all-gather over ranks, group statistics, local slice (:651-699)
        rewards = self.reward_fn(completion_ids, cmask).float().contiguous()
        if self.world > 1:
            gathered = [torch.empty_like(rewards) for _ in range(self.world)]
            dist.all_gather(gathered, rewards)
            all_rewards = torch.cat(gathered, dim=0)
        else:
            all_rewards = rewards
        adv, gmean, gstd = grpo.group_advantages(all_rewards, c.num_generations, self.rank, B)
        mark("rewards")
        # ---- policy forward / loss / backward (:777-814)
        m.arena.zero_grad()
        lp = grpo.per_token_logps(m, prompt_ids, prompt_mask, completion_ids, cmask, **mm)
        eps_hi = c.epsilon_high if c.epsilon_high is not None else c.epsilon
        loss, stats = grpo.grpo_loss(lp, None, ref_lp, adv, cmask, c.epsilon, eps_hi, c.beta)
        mark("policy_fwd")
        loss.backward()
        mark("policy_bwd")
        # ---- gradient reduction: one flat bucket (DDP averages)
        if self.world > 1:
            dist.all_reduce(m.arena.grads)
            scale = 1.0 / self.world
        else:
            scale = 1.0
        m.arena.adamw_step(c.learning_rate, (c.adam_beta1, c.adam_beta2), c.adam_epsilon, c.weight_decay,
                           max_grad_norm=c.max_grad_norm, grad_scale=scale)
        mark("optimizer")
        self.step_idx += 1
        out = {"loss_t": loss.detach(), "stats_t": stats, "reward_mean_t": all_rewards.sum(1).mean()}
        if timing and marks:
            torch.cuda.synchronize()
            for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
                self.timers[n1] = e0.elapsed_time(e1)
        return out
