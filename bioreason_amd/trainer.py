"""Training steps on the HIP DNA-LLM, data-parallel over the GPUs of a node.

`GRPOStepRunner` mirrors DNALLMGRPOTrainer.compute_loss / _generate_and_score_completions (grpo_trainer.py:535-814):
    rollout (generate)  ->  EOS mask  ->  [old policy log-probs when num_iterations > 1, :620-626]
    ->  reference log-probs (adapters disabled, :636-640)  ->  rewards
    ->  all-gather of rewards over ranks (:679)  ->  group advantages, local slice (:682-699)
    ->  policy log-probs (:777-779)  ->  clipped objective + beta*KL (:786-807)  ->  backward
    ->  gradient all-reduce  ->  AdamW (+ grad clip),
with the reference's rollout buffering: one buffered rollout per gradient-accumulation slot, regenerated every
`num_iterations` optimiser steps (:757-762).
`SFTStepRunner` is the train_dna_qwen.py step (:179-213 `_step` -> outputs.loss, AdamW :393-397).

Parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).  The path has exactly two
exchange steps, both on tiny or single-bucket payloads:
  * ONE all-gather of a packed [B_local, F + 1] fp32 record per rollout (rewards per function + completion length —
    the reference issues `accelerator.gather` once plus four `gather_for_metrics`, :679-716);
  * the all-reduce of the flat gradient arena (≈148 MB fp32 for Qwen3-1.7B r=32), cut into a few layer-aligned
    buckets that are issued from inside the backward as soon as their layers are done (overlap with the rest of the
    backward); the per-rank scalars of compute_loss's metrics (loss, KL, clip ratio, :803-812) ride in a spare slot of
    the first bucket instead of three more collectives.
No model sharding: NT-500M + Qwen3-1.7B + KV cache + activations use < 60 GB of the 288 GB HBM.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist

from . import grpo, ops

METRIC_SLOT = "__dp_metrics__"


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
    num_iterations: int = 1
    gradient_accumulation_steps: int = 1
    learning_rate: float = 1e-5
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 1.0
    eos_token_id: Optional[int] = None
    pad_token_id: Optional[int] = None
    seed: int = 42
    # execution knobs (no effect on the arithmetic): hipGraph replay of the token loop, decode attention over one shared
    # copy of each prompt's K/V, number of gradient buckets overlapped with the backward
    rollout_graph: Optional[bool] = None
    rollout_shared_prefix: bool = True
    # opt-in (BASELINE config 5, "fp8 weights"): the token loop streams e4m3 images of the merged weights, one fp32 scale per output row
    # (half the bytes; W8A16: bf16 activations, fp32 accumulation).  The rollouts are then sampled from the QUANTISED policy — log-probs,
    # the reference pass and the gradients stay bf16 — i.e. the behaviour policy differs from the trained one by quantisation noise, as
    # with any fp8 rollout engine; never on in the bf16 headline.
    rollout_fp8: bool = False
    # opt-in, with rollout_fp8 (round 6): the prompt pass of the rollout runs its projections on the fp8 MFMA path over the same quantised
    # weights (generation.prefill: W8A8, per-token activation scales — on by default with rollout_fp8, BRA_FP8_PREFILL=0 restores the bf16
    # prompt pass); ref_fp8 also runs the no-grad REFERENCE pass that way over e4m3 images of the base weights (the KL term then measures
    # the distance to the quantised reference policy).  The policy pass and every gradient stay bf16.
    ref_fp8: bool = False
    grad_buckets: int = 4
    # transport dtype of the gradient all-reduce: "fp32" (default: the flat arena itself, 148 MB) or "bf16" (each bucket is rounded to a
    # bf16 image, summed over the ranks in bf16 and widened back: half the xGMI bytes, one rounding per rank and per sum step — DDP's
    # bf16 compression hook; the metric slot rides in fp32 either way)
    grad_allreduce_dtype: str = "fp32"
    share_dna_encoding: bool = True      # frozen-encoder rows computed once per step and shared by its three passes
    # the policy pass (forward AND backward) may run the prompt of a group of consecutive copies once
    # (grpo.per_token_logps_shared_policy): identical results at lora_dropout = 0; under LoRA dropout the shared prompt rows would carry
    # ONE mask stream for all copies of a prompt, where PEFT draws one per copy (grpo_trainer.py:777-779 runs every row's full prompt).
    # None (default) = the reference's sampling scheme decides: shared when the adapters have no dropout, the full-row pass with
    # independent masks per copy when they do.  True opts into the shared pass under dropout too (unbiased, lower variance — DESIGN.md
    # section 6 — but not the reference's scheme; bench.py opts in explicitly and reports both).
    share_policy_prompt: Optional[bool] = None
    # the reference-policy pass (no grad, adapters off) on a second HIP stream, concurrent with the policy forward: at one prompt x 8
    # rollouts both are chains of kernels that fill about half of the chip (grids of 128 - 144 workgroups on 256 CUs)
    overlap_ref_pass: bool = True
    # the two row segments of the shared-prompt policy pass (prompt chain, completion chain) on two HIP streams, one event per layer
    overlap_policy_chains: bool = True
    overlap_ref_chains: bool = False         # prompt / completion chains of the reference pass on two streams as well (measured: see NOTES)
    overlap_rollout_weights: bool = True     # merge + pack the rollout's weight set on a side stream beside the frozen DNA encoder


def token_stat_rewards(completion_ids: torch.Tensor, completion_mask: torch.Tensor) -> torch.Tensor:
    """Stand-in reward functions for synthetic runs without a tokenizer: two cheap statistics of the sampled ids ->
    [B, 2] fp32 (the reference's are CPU regexes over decoded text, reason.py:193-230; see `text_reward_fn`)."""
    m = completion_mask.float()
    n = m.sum(1).clamp(min=1)
    r0 = ((completion_ids % 7 == 0).float() * m).sum(1) / n * 2.0
    r1 = ((completion_ids % 2 == 0).float() * m).sum(1) / n * 0.5
    return torch.stack([r0, r1], dim=1)


from .rewards import reward_hop as text_reward_fn      # noqa: E402  (the reference's reward hop; ONE implementation, rewards.py)


# =============================================================================================== data-parallel plumbing
class _DataParallelStep:
    """Gradient reduction + optimiser shared by the GRPO and SFT steps: layer-aligned buckets of the flat gradient arena are
    all-reduced asynchronously from inside the backward (engine.layer_done_hook), the tail bucket (first layers + the
    projection, whose gradients arrive last) after it; AdamW + global-norm clip is one fused launch."""

    def __init__(self, model, n_buckets: int = 4, grad_dtype: str = "fp32"):
        self.model = model
        if grad_dtype not in ("fp32", "bf16"):
            raise ValueError("grad_allreduce_dtype must be 'fp32' or 'bf16'")
        self._grad_bf16 = grad_dtype == "bf16"
        self._bf_images: List = []
        self._bf_cache: Dict[tuple, torch.Tensor] = {}      # bf16 transport images, one per arena range, allocated once
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # `dp`: the collectives of the step are issued.  BRA_DP_SINGLE_RANK=1 issues them in a ONE-rank group too (each is then the
        # identity): the hardware rehearsal of the RCCL path on a 1-GPU box (tools/rccl_single_rank.sh), never set in production
        self.dp = dist.is_initialized() and (self.world > 1 or os.environ.get("BRA_DP_SINGLE_RANK") == "1")
        self.timers: Dict[str, float] = {}
        self._handles: List = []
        # per-bucket stamps of ONE backward (trace_buckets = True, read with bucket_report()): when each gradient bucket was handed to
        # the collective relative to the start of the backward, and how long the step then WAITED for it after the backward — the first
        # multi-GPU run shows overlap per bucket instead of one total
        self.trace_buckets = False
        self._trace: List[dict] = []
        self._trace_t0 = None
        self._cuts: Dict[int, tuple] = {}
        self._n_buckets = max(1, int(n_buckets))
        arena = model.arena
        if self.dp and METRIC_SLOT not in arena._shapes:
            arena.add(METRIC_SLOT, 1, 64)            # mask stays 0: excluded from the norm and from the update
            arena.commit()
            arena.pack()
        self._plan_buckets()

    def _plan_buckets(self):
        """bucket k = the arena range of a run of consecutive decoder layers, issued when the LOWEST layer of the run has
        finished its backward (layers finish from the top down); everything below the last cut goes after the backward"""
        a = self.model.arena
        eng = self.model.text_model.ensure_packed()
        L = eng.L
        starts = []
        for li in range(L):
            offs = [a._offsets[k] for k in a._offsets if k.startswith(f"text.layers.{li}.")]
            starts.append(min(offs) if offs else None)
        self._cuts = {}
        if not self.dp or any(s is None for s in starts) or sorted(starts) != starts:
            return
        nb = min(self._n_buckets, L)
        hi = a.numel
        for k in range(nb - 1):
            lo_layer = L - (k + 1) * L // nb
            if lo_layer <= 0:
                break
            lo = starts[lo_layer]
            if lo < hi:
                self._cuts[lo_layer] = (lo, hi)
                hi = lo
        self._tail_hi = hi

    def _issue(self, lo: int, hi: int):
        """asynchronous sum of the gradient range [lo, hi) over the ranks, in the configured transport dtype"""
        g = self.model.arena.grads[lo:hi]
        tr = None
        if self.trace_buckets and g.device.type == "cuda":
            tr = {"lo": int(lo), "hi": int(hi), "bytes": int((hi - lo) * (2 if self._grad_bf16 else 4)), "first_handle": len(self._handles),
                  "issue": torch.cuda.Event(enable_timing=True)}
            tr["issue"].record()                     # on the stream the hook runs on: everything the bucket depends on is ordered before it
            self._trace.append(tr)
        if not self._grad_bf16:
            self._handles.append(dist.all_reduce(g, async_op=True))
            return
        a = self.model.arena
        # the per-rank loss / KL / clip-ratio scalars (METRIC_SLOT) are averaged in fp32 and never pass through a bf16 image: the
        # bucket that holds the slot is cast, reduced and widened as the two ranges around it (no read of the slot by the cast
        # while the fp32 all-reduce writes it, no restore afterwards).  The images are allocated once per range.
        m_lo = a._offsets.get(METRIC_SLOT)
        if m_lo is not None and lo <= m_lo < hi:
            self._handles.append(dist.all_reduce(a.grads[m_lo:m_lo + 64], async_op=True))
            ranges = [(lo, m_lo), (m_lo + 64, hi)]
        else:
            ranges = [(lo, hi)]
        for r_lo, r_hi in ranges:
            if r_hi <= r_lo:
                continue
            img = self._bf_cache.get((r_lo, r_hi))
            if img is None:
                img = self._bf_cache[(r_lo, r_hi)] = torch.empty(r_hi - r_lo, dtype=torch.bfloat16, device=g.device)
            ops.cast_grad(a.grads[r_lo:r_hi], img)
            self._handles.append(dist.all_reduce(img, async_op=True))
            self._bf_images.append((r_lo, r_hi, img))

    def _layer_done(self, li: int):
        cut = self._cuts.get(li)
        if cut is not None:
            self._issue(cut[0], cut[1])

    def begin_backward(self):
        self._handles = []
        if self.trace_buckets and torch.cuda.is_available() and self.dp:
            self._trace = []
            self._trace_t0 = torch.cuda.Event(enable_timing=True)
            self._trace_t0.record()
        eng = self.model.text_model.engine
        eng.layer_done_hook = self._layer_done if (self.dp and self._cuts) else None

    def reduce_gradients(self) -> float:
        """finish the data-parallel sum; returns the factor that turns the sum into DDP's mean"""
        if not self.dp:
            return 1.0
        eng = self.model.text_model.engine
        eng.layer_done_hook = None
        g = self.model.arena.grads
        hi = self._tail_hi if self._cuts else g.numel()
        self._issue(0, hi)
        tracing = self.trace_buckets and self._trace and self._trace_t0 is not None
        if tracing:
            self._trace_join = torch.cuda.Event(enable_timing=True)
            self._trace_join.record()                # the backward's last launch: waits from here on are exposed
            firsts = {t["first_handle"]: t for t in self._trace}
            nxt = sorted(firsts) + [len(self._handles)]
        for i, h in enumerate(self._handles):
            h.wait()
            if tracing and (i + 1) in nxt:           # the last handle of a bucket has been joined into the compute stream
                t = firsts[nxt[nxt.index(i + 1) - 1]]
                t["done"] = torch.cuda.Event(enable_timing=True)
                t["done"].record()
        self._handles = []
        for lo, hi_, img in self._bf_images:                    # widen the summed bf16 images back into the arena
            ops.cast_grad(img, g[lo:hi_])
        self._bf_images = []
        return 1.0 / self.world

    def bucket_report(self) -> List[dict]:
        """stamps of the last traced backward, in ms: `issued_at` = hand-off of the bucket to the collective after the start of the
        backward; `joined_at` = when the compute stream had waited for it; `backward_end` = the backward's own last launch.  A bucket
        whose joined_at equals backward_end (within the event resolution) was fully hidden behind the backward."""
        if not self._trace or self._trace_t0 is None:
            return []
        torch.cuda.synchronize()
        end = self._trace_t0.elapsed_time(self._trace_join)
        return [{"range": [t["lo"], t["hi"]], "bytes": t["bytes"], "issued_at_ms": round(self._trace_t0.elapsed_time(t["issue"]), 3),
                 "joined_at_ms": round(self._trace_t0.elapsed_time(t["done"]), 3) if "done" in t else None,
                 "backward_end_ms": round(end, 3)} for t in self._trace]

    def _marks(self, timing: bool, dev):
        marks: List = []

        def mark(name):
            if timing and dev.type == "cuda":
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append((name, e))
        return marks, mark

    def _close_marks(self, marks):
        if marks:
            torch.cuda.synchronize()
            for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
                self.timers[n1] = self.timers.get(n1, 0.0) + e0.elapsed_time(e1)


# =============================================================================================== GRPO
class GRPOStepRunner(_DataParallelStep):
    metric_names = ["completion_length", "reward", "reward_std", "loss", "kl", "clip_ratio"]

    def __init__(self, model, cfg: GRPOConfig, reward_fn: Callable = token_stat_rewards):
        super().__init__(model, cfg.grad_buckets, getattr(cfg, "grad_allreduce_dtype", "fp32"))
        self.cfg, self.reward_fn = cfg, reward_fn
        self.global_step = 0                # optimiser steps (HF: self.state.global_step)
        self._step = 0                      # forward/backward passes, incl. those inside an accumulation cycle (:395)
        self.step_idx = 0                   # rollouts generated (seeds the sampler)
        self._buffered_inputs: List[Optional[Dict]] = [None] * max(1, cfg.gradient_accumulation_steps)
        if hasattr(model.text_model, "set_dropout_seed"):          # every rank draws its own LoRA dropout masks, as separate
            model.text_model.set_dropout_seed(cfg.seed * 1000003 + self.rank)   # processes with their own RNG streams do
        self.rollout_profile: Dict[str, float] = {}      # filled by step(timing=True): phases inside generate(), ms
        self.loop_events: Optional[list] = None          # a list: generate() appends (start, end, token steps) HIP events of its token loop
        # optimiser step index -> learning rate (HF Trainer.create_scheduler, set by DNALLMGRPOTrainer); None = constant cfg.learning_rate
        self.lr_schedule: Optional[Callable[[int], float]] = None
        self.last_lr = cfg.learning_rate

    def _side_stream(self, dev, which: int = 0):
        if dev.type != "cuda":
            return None
        if getattr(self, "_side", None) is None:
            self._side = {}
        if which not in self._side:
            self._side[which] = torch.cuda.Stream(device=dev)
        return self._side[which]

    # ---- _generate_and_score_completions (:535-749) ---------------------------------------------------------------
    def generate_and_score(self, batch: Dict, timing: bool = False, mark=lambda n: None, defer_ref_join: bool = False) -> Dict:
        """(the fp8 rollout switch of the config is in force for THIS call only: a `model.rollout_fp8` the user set by hand, or left
        unset, is what later `generate()` / evaluation calls see again)"""
        tm = self.model.text_model
        had, prev = hasattr(tm, "rollout_fp8"), getattr(tm, "rollout_fp8", False)
        tm.rollout_fp8 = bool(prev) or bool(self.cfg.rollout_fp8)        # read by generation.rollout_weights / the decode states
        try:
            return self._generate_and_score(batch, timing, mark, defer_ref_join)
        finally:
            if had:
                tm.rollout_fp8 = prev
            else:
                try:
                    delattr(tm, "rollout_fp8")
                except AttributeError:
                    pass

    def _generate_and_score(self, batch: Dict, timing: bool = False, mark=lambda n: None, defer_ref_join: bool = False) -> Dict:
        """`defer_ref_join` (step() only): the reference pass may still be running on its side stream when this returns — the
        returned dict then carries the stream under "ref_join" and `compute_loss` joins it just before the loss reads
        `ref_per_token_logps`.  Every other caller gets the join here: whatever it reads from the dict is complete on the
        current stream."""
        m, c = self.model, self.cfg
        dev = batch["input_ids"].device
        mm = {"dna_tokenized": batch["dna_tokenized"], "batch_idx_map": batch["batch_idx_map"], "dna_alias": batch.get("dna_alias")}
        wside = None
        if (c.overlap_rollout_weights and dev.type == "cuda" and not timing and c.rollout_shared_prefix
                and batch.get("prompt_alias") is not None):
            # the rollout's weight set (LoRA merged into the base weights, gate / up interleaved, fragment-packed: ~17 GB of HBM
            # traffic, ~4 ms) depends on the parameters only: built on a side stream while the main stream runs the frozen encoder
            # (an under-filled GEMM chain); generation.rollout_weights caches it by parameter version, so generate() finds it
            from . import generation as _gen
            wside = self._side_stream(dev, 2)
            wside.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(wside):
                _gen.rollout_weights(m.text_model, rows=int(batch["input_ids"].shape[0]))
        if c.share_dna_encoding and batch["dna_tokenized"] is not None and batch["batch_idx_map"]:
            # the frozen encoder (no_grad, dna_llm.py:121) once per distinct sequence and STEP: its rows feed the rollout, the
            # reference pass and the policy pass alike (the reference evaluates it three times to the same values); the
            # trainable projection on top is applied — and differentiated — in every pass
            mm["dna_enc"] = m.encode_dna(batch["dna_tokenized"], batch.get("dna_alias"))
        prompt_ids, prompt_mask = batch["input_ids"], batch["attention_mask"]
        B = prompt_ids.shape[0]
        if wside is not None:
            torch.cuda.current_stream(dev).wait_stream(wside)
        sched = batch.get("eos_schedule")
        # rollout (unwrapped_model.generate, :579-596): only completion ids come back
        completion_ids = m.generate(input_ids=prompt_ids, attention_mask=prompt_mask, **mm,
                                    max_new_tokens=c.max_completion_length, do_sample=True, temperature=c.temperature,
                                    top_k=c.top_k, top_p=c.top_p, eos_token_id=c.eos_token_id, pad_token_id=c.pad_token_id,
                                    seed=c.seed + 1000003 * self.step_idx + self.rank, return_full_length=sched is None,
                                    prompt_alias=batch.get("prompt_alias"), use_graph=c.rollout_graph,
                                    shared_prefix_decode=c.rollout_shared_prefix, eos_schedule=sched,
                                    profile=self.rollout_profile if timing else None, loop_events=self.loop_events)
        self.step_idx += 1
        mark("rollout")
        if c.eos_token_id is not None:
            cmask = grpo.completion_mask(completion_ids, c.eos_token_id)
        else:
            cmask = torch.ones(completion_ids.shape, dtype=torch.int32, device=dev)
        old_lp = None
        if c.num_iterations > 1:            # :620-626 — the sampling policy's log-probs, adapters ON, no grad
            with torch.no_grad():
                old_lp = grpo.per_token_logps(m, prompt_ids, prompt_mask, completion_ids, cmask, **mm).detach()
        # reference policy = the same network with adapters disabled (:636-640)
        ref_lp = None
        ref_join = None
        if c.beta != 0.0:
            def ref_pass():
                eng_ = m.text_model.ensure_packed()
                w8 = None
                if c.ref_fp8 and eng_.fp8_supported():          # (other widths keep the bf16 reference pass: the flag never fails a run)
                    sig = getattr(m.text_model, "_packed_sig", None)
                    cached = getattr(eng_, "_ref_fp8", None)
                    if cached is None or cached[0] != sig:
                        cached = eng_._ref_fp8 = (sig, eng_.fp8_weight_images())      # base weights: static, built once
                    w8 = cached[1]
                with torch.no_grad(), m.text_model.disable_adapter(), eng_.use_fp8(w8):
                    if batch.get("prompt_alias") is not None:
                        return grpo.per_token_logps_shared_prefix(m, prompt_ids, prompt_mask, completion_ids, cmask,
                                                                  batch["prompt_alias"], side=ref_side2, **mm)
                    return grpo.per_token_logps(m, prompt_ids, prompt_mask, completion_ids, cmask, **mm)
            side = self._side_stream(dev) if (c.overlap_ref_pass and not timing) else None
            ref_side2 = self._side_stream(dev, 3) if (side is not None and c.overlap_ref_chains) else None
            if side is not None:
                # issued on the side stream behind everything issued so far; the caller joins (`ref_join`) before the loss reads it
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    ref_lp = ref_pass()
                ref_join = side
            else:
                ref_lp = ref_pass()
        mark("ref_logps")
        # rewards (+ completion length) -> ONE all-gather -> group statistics -> local slice (:651-699, 703-716)
        rewards = self.reward_fn(completion_ids, cmask).float()
        packed = torch.cat([rewards, cmask.sum(1, keepdim=True).float()], dim=1).contiguous()
        if self.dp:
            gathered = [torch.empty_like(packed) for _ in range(self.world)]
            dist.all_gather(gathered, packed)
            packed_all = torch.cat(gathered, dim=0)
        else:
            packed_all = packed
        F = rewards.shape[1]
        all_rewards = packed_all[:, :F].contiguous()
        adv, gmean, gstd = grpo.group_advantages(all_rewards, c.num_generations, self.rank, B)
        roll_metrics = torch.stack([packed_all[:, F].mean(), all_rewards.sum(1).mean(), gstd.mean()])
        mark("rewards")
        if ref_join is not None and not defer_ref_join:
            torch.cuda.current_stream(dev).wait_stream(ref_join)
            ref_lp.record_stream(torch.cuda.current_stream(dev))
            ref_join = None
        return {"prompt_ids": prompt_ids, "prompt_mask": prompt_mask, "completion_ids": completion_ids, "completion_mask": cmask,
                "old_per_token_logps": old_lp, "ref_per_token_logps": ref_lp, "advantages": adv, "multimodal_inputs": mm,
                "prompt_alias": batch.get("prompt_alias"), "ref_join": ref_join,
                "rewards_per_func": all_rewards.mean(0), "roll_metrics": roll_metrics}

    def shares_policy_prompt(self) -> bool:
        """`cfg.share_policy_prompt`, with None resolved by the adapters' dropout (see GRPOConfig)"""
        want = self.cfg.share_policy_prompt
        if want is None:
            return float(getattr(self.model.text_model, "lora_dropout_p", 0.0) or 0.0) == 0.0
        return bool(want)

    # ---- compute_loss (:751-814) ----------------------------------------------------------------------------------
    def compute_loss(self, inputs: Dict):
        m, c = self.model, self.cfg
        lp = None
        if self.shares_policy_prompt() and inputs.get("prompt_alias") is not None:
            side2 = self._side_stream(inputs["prompt_ids"].device, 1) if c.overlap_policy_chains else None
            lp = grpo.per_token_logps_shared_policy(m, inputs["prompt_ids"], inputs["prompt_mask"], inputs["completion_ids"],
                                                    inputs["completion_mask"], inputs["prompt_alias"], side=side2,
                                                    **inputs["multimodal_inputs"])
        if lp is None:
            lp = grpo.per_token_logps(m, inputs["prompt_ids"], inputs["prompt_mask"], inputs["completion_ids"],
                                      inputs["completion_mask"], **inputs["multimodal_inputs"])
        eps_hi = c.epsilon_high if c.epsilon_high is not None else c.epsilon
        old = inputs["old_per_token_logps"] if c.num_iterations > 1 else None          # :786
        if inputs.get("ref_join") is not None:                       # the reference pass ran beside the policy forward: join
            torch.cuda.current_stream(lp.device).wait_stream(inputs["ref_join"])
            inputs["ref_join"] = None
        return grpo.grpo_loss(lp, old, inputs["ref_per_token_logps"], inputs["advantages"], inputs["completion_mask"],
                              c.epsilon, eps_hi, c.beta)

    def step(self, batch: Dict, timing: bool = False) -> Dict[str, torch.Tensor]:
        """one HF `training_step`: one forward/backward over one micro-batch; the optimiser runs when the accumulation cycle
        closes.  `batch` is ignored (the buffered rollout is reused) on the passes the reference reuses it (:757-762)."""
        m, c = self.model, self.cfg
        dev = batch["input_ids"].device
        marks, mark = self._marks(timing, dev)
        if timing:
            self.rollout_profile.clear()
            self.timers.clear()
        ga = max(1, c.gradient_accumulation_steps)
        slot = self._step % ga
        mark("start")
        if self.global_step % c.num_iterations == 0:
            inputs = self.generate_and_score(batch, timing, mark, defer_ref_join=True)
            self._buffered_inputs[slot] = inputs
        else:
            inputs = self._buffered_inputs[slot]
        self._step += 1
        if slot == 0:
            m.arena.zero_grad()
        if slot == ga - 1:
            self.begin_backward()
        loss, stats = self.compute_loss(inputs)
        mark("policy_fwd")
        if self.dp:                         # loss / KL / clip ratio of this rank ride in the gradient bucket's spare slot
            m.arena.grad(METRIC_SLOT).view(-1)[:3].add_(stats / ga)
        (loss / ga if ga > 1 else loss).backward()
        mark("policy_bwd")
        out = {"loss_t": loss.detach(), "stats_t": stats, "reward_mean_t": inputs["roll_metrics"][1]}
        if slot == ga - 1:
            scale = self.reduce_gradients()
            if self.dp:
                local3 = m.arena.grad(METRIC_SLOT).view(-1)[:3] * scale
            else:
                local3 = stats
            lr = c.learning_rate if self.lr_schedule is None else float(self.lr_schedule(self.global_step))
            self.last_lr = lr
            m.arena.adamw_step(lr, (c.adam_beta1, c.adam_beta2), c.adam_epsilon, c.weight_decay,
                               max_grad_norm=c.max_grad_norm, grad_scale=scale)
            self.global_step += 1
            out["metrics_t"] = torch.cat([inputs["roll_metrics"], local3.detach().clone()])
            out["rewards_per_func_t"] = inputs["rewards_per_func"]
        mark("optimizer")
        if timing:
            self._close_marks(marks)
        return out


# =============================================================================================== SFT
def cosine_schedule_with_warmup(base_lr: float, total_steps: int, warmup_ratio: float = 0.1) -> Callable[[int], float]:
    """train_dna_qwen.py:393-411: `get_cosine_schedule_with_warmup(optimizer, int(0.1 * total_steps), total_steps)` stepped once per
    optimiser step — the rate of optimiser step `i` (0-based) is base_lr x lambda(i), lambda as in transformers/optimization.py
    (`_get_cosine_schedule_with_warmup_lr_lambda`, half a cosine period): linear from 0 over the warm-up steps, then
    0.5 (1 + cos(pi progress)) down to 0 at `total_steps`.  tests/test_sft_runner.py holds it equal to the installed scheduler."""
    import math
    total = max(1, int(total_steps))
    warm = int(warmup_ratio * total)

    def rate(i: int) -> float:
        if i < warm:
            return base_lr * float(i) / float(max(1, warm))
        prog = float(i - warm) / float(max(1, total - warm))
        return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
    return rate


class SFTStepRunner(_DataParallelStep):
    """train_dna_qwen.py:179-213 + :393-411 + the trainer settings of :985-1005: forward (loss + full-row logits, `outputs.logits` being
    part of the model contract), backward, DDP-mean of the gradients, clip at 1.0, AdamW.  One call = one micro-batch; the optimiser
    runs when the accumulation cycle closes (`accumulate_grad_batches`: the loss of a micro-batch is divided by the cycle length,
    the data-parallel sum is taken once, from inside the last backward of the cycle).  `lr_schedule` (optimiser step -> rate;
    `cosine_schedule_with_warmup` is the reference's) replaces the constant `learning_rate` when set."""

    def __init__(self, model, learning_rate: float = 1e-4, weight_decay: float = 0.01, max_grad_norm: float = 1.0,
                 n_buckets: int = 4, return_logits: bool = True, gradient_accumulation_steps: int = 1,
                 lr_schedule: Optional[Callable[[int], float]] = None):
        super().__init__(model, n_buckets)
        self.lr, self.wd, self.max_grad_norm, self.return_logits = learning_rate, weight_decay, max_grad_norm, return_logits
        self.ga = max(1, int(gradient_accumulation_steps))
        self.lr_schedule = lr_schedule
        self.global_step = 0                # optimiser steps taken
        self._micro = 0                     # micro-batches seen
        self.last_lr = learning_rate
        if hasattr(model.text_model, "set_dropout_seed"):
            model.text_model.set_dropout_seed(23 * 1000003 + self.rank)

    def step(self, batch: Dict, timing: bool = False) -> Dict[str, torch.Tensor]:
        m = self.model
        dev = batch["input_ids"].device
        marks, mark = self._marks(timing, dev)
        if timing:
            self.timers.clear()
        mark("start")
        slot = self._micro % self.ga
        self._micro += 1
        if slot == 0:
            m.arena.zero_grad()
        if slot == self.ga - 1:
            self.begin_backward()
        out = m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], dna_tokenized=batch["dna_tokenized"],
                batch_idx_map=batch["batch_idx_map"], labels=batch["labels"], return_logits=self.return_logits)
        mark("forward")
        (out.loss / self.ga if self.ga > 1 else out.loss).backward()
        mark("backward")
        res = {"loss_t": out.loss.detach()}
        if slot == self.ga - 1:
            scale = self.reduce_gradients()
            lr = self.lr if self.lr_schedule is None else float(self.lr_schedule(self.global_step))
            self.last_lr = lr
            m.arena.adamw_step(lr, (0.9, 0.999), 1e-8, self.wd, max_grad_norm=self.max_grad_norm, grad_scale=scale)
            self.global_step += 1
            res["stepped"] = True
        mark("optimizer")
        if timing:
            self._close_marks(marks)
        return res
