"""`DNALLMGRPOTrainer` / `DNALLMGRPOConfig` / `RepeatRandomSampler` with the reference's constructor surface
(bioreason/trainer/grpo_trainer.py:72-119, 206-223; grpo_config.py:146-365), driving `trainer.GRPOStepRunner`.

What is kept: the constructor arguments `reason.py:566-577` passes, the config field names and defaults, the sampler, the
text path of a training step (dataset rows -> `dna_module.prepare_prompt` -> `prepare_model_inputs` (DLProcessor) -> rollout
-> batch_decode -> python reward functions -> gathered advantages -> loss), rollout buffering across
gradient-accumulation slots and `num_iterations`, the metric names of `log`, `train()`, `save_model()`.
The learning rate follows HF `Trainer.create_scheduler`: `args.lr_scheduler_type` (default linear decay to 0), `warmup_steps` /
`warmup_ratio`, over `max_steps` or epochs x optimiser steps per epoch — computed with transformers' own `get_scheduler`.
What is not re-created (SURVEY §8b / §2 out of scope): the HF `Trainer` base class (callback bus, wandb, hub, evaluation
loop, deepspeed / accelerate wrappers, model cards) and reward *models* (`reward_funcs` entries must be callables).
LoRA: `peft_config` may be a peft `LoraConfig` or any object / dict with `r`, `lora_alpha`, `lora_dropout`
(peft itself is not required); targets are every text-model linear outside the DNA modules, as `find_all_linear_names`
selects them (grpo_trainer.py:256-274).
"""
from __future__ import annotations

import os
import math
import time
from collections import defaultdict
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Any, Callable, Dict, Iterator, List, Optional, Sized, Union

import torch
import torch.distributed as dist
from transformers import TrainingArguments

from .trainer import GRPOConfig, GRPOStepRunner, text_reward_fn


# ------------------------------------------------------------------------------------------------- sampler (:72-119)
class RepeatRandomSampler(torch.utils.data.Sampler):
    """Every rank draws the same seeded permutation; chunks of `batch_size` unique indices, each index emitted
    `mini_repeat_count` times in a row, the whole chunk `repeat_count` times; an incomplete last chunk is dropped."""

    def __init__(self, data_source: Sized, mini_repeat_count: int, batch_size: int = 1, repeat_count: int = 1,
                 seed: Optional[int] = None):
        self.data_source = data_source
        self.mini_repeat_count = mini_repeat_count
        self.batch_size = batch_size
        self.repeat_count = repeat_count
        self.num_samples = len(data_source)
        self.seed = seed
        self.generator = torch.Generator()
        if seed is not None:
            self.generator.manual_seed(seed)

    def __iter__(self) -> Iterator[int]:
        perm = torch.randperm(self.num_samples, generator=self.generator).tolist()
        for lo in range(0, len(perm) - self.batch_size + 1, self.batch_size):
            chunk = perm[lo:lo + self.batch_size]
            for _ in range(self.repeat_count):
                for index in chunk:
                    for _ in range(self.mini_repeat_count):
                        yield index

    def __len__(self) -> int:
        return self.num_samples * self.mini_repeat_count * self.repeat_count


class LengthBucketedRepeatSampler(RepeatRandomSampler):
    """`RepeatRandomSampler` whose global batches hold prompts of SIMILAR expected cost (length-bucketed scheduling
    across ranks).  Every optimisation step ends in a collective (reward all-gather, gradient all-reduce: grpo_trainer.py:679-699), so a
    step lasts as long as its slowest rank; with one or two prompts per rank the ranks of a step differ by whatever the shuffle dealt them.
    Here the seeded permutation is cut into windows of `bucket_batches` batches; inside a window the indices are sorted by `costs[index]`
    (prompt + DNA length, or a running estimate of the completion length), cut into batches, and the batches of the window are emitted in
    a seeded random order.  Every epoch is still a permutation of the same indices (minus the dropped incomplete batch), every rank still
    draws the same stream, and `bucket_batches = 1` IS the reference sampler, index for index.
    `costs` may be updated between epochs (`set_costs`): the trainer feeds back observed completion lengths."""

    def __init__(self, data_source: Sized, mini_repeat_count: int, batch_size: int = 1, repeat_count: int = 1,
                 seed: Optional[int] = None, costs=None, bucket_batches: int = 1):
        super().__init__(data_source, mini_repeat_count, batch_size, repeat_count, seed)
        self.bucket_batches = max(1, int(bucket_batches))
        self.costs = None
        self.set_costs(costs)

    def set_costs(self, costs) -> None:
        if costs is not None and len(costs) != self.num_samples:
            raise ValueError(f"costs has {len(costs)} entries for {self.num_samples} samples")
        self.costs = None if costs is None else [float(c) for c in costs]

    def __iter__(self) -> Iterator[int]:
        perm = torch.randperm(self.num_samples, generator=self.generator).tolist()
        nb = len(perm) // self.batch_size
        batches = [perm[b * self.batch_size:(b + 1) * self.batch_size] for b in range(nb)]
        if self.costs is not None and self.bucket_batches > 1:
            out = []
            for w0 in range(0, nb, self.bucket_batches):
                window = [i for b in batches[w0:w0 + self.bucket_batches] for i in b]
                window.sort(key=lambda i: (self.costs[i], i))                       # (ties by index: identical on every rank)
                wb = [window[j:j + self.batch_size] for j in range(0, len(window), self.batch_size)]
                order = torch.randperm(len(wb), generator=self.generator).tolist()  # buckets leave the window in random order
                out.extend(wb[j] for j in order)
            batches = out
        for chunk in batches:
            for _ in range(self.repeat_count):
                for index in chunk:
                    for _ in range(self.mini_repeat_count):
                        yield index


def prompt_cost(example: Dict[str, Any]) -> float:
    """static cost proxy of a training example before anything is known about its completions: DNA characters (6 per NT-v2 token,
    1 per Evo2 token) + characters of the text part of the prompt"""
    n = sum(len(s_) for s_ in (example.get("dna_sequences") or []))
    prompt = example.get("prompt")
    if isinstance(prompt, str):
        n += len(prompt)
    elif prompt:
        for msg in prompt:
            c = msg.get("content")
            if isinstance(c, str):
                n += len(c)
            elif c:
                n += sum(len(it.get("text") or "") for it in c if isinstance(it, dict))
    return float(n)


# ------------------------------------------------------------------------------------------------- config (:146-365)
@dataclass
class DNALLMGRPOConfig(TrainingArguments):
    """`transformers.TrainingArguments` + the GRPO fields of the reference's config, same names and defaults
    (grpo_config.py:146-365).  vLLM / reference-model-sync fields are accepted for command-line compatibility and unused
    (the reference never reaches its vLLM branch either: the code is commented out, demo_grpo.py)."""
    model_init_kwargs: Optional[dict] = field(default=None)
    remove_unused_columns: Optional[bool] = field(default=False)
    max_prompt_length: Optional[int] = field(default=512)
    num_generations: Optional[int] = field(default=8)
    max_completion_length: Optional[int] = field(default=800)
    ds3_gather_for_generation: bool = field(default=True)
    temperature: float = field(default=0.6)
    top_p: float = field(default=0.95)
    top_k: Optional[int] = field(default=20)
    min_p: Optional[float] = field(default=None)
    repetition_penalty: float = field(default=1.0)
    cache_implementation: Optional[str] = field(default=None)
    use_vllm: Optional[bool] = field(default=False)
    vllm_device: Optional[str] = field(default="auto")
    vllm_gpu_memory_utilization: float = field(default=0.9)
    vllm_dtype: Optional[str] = field(default="auto")
    vllm_max_model_len: Optional[int] = field(default=None)
    vllm_enable_prefix_caching: Optional[bool] = field(default=True)
    vllm_guided_decoding_regex: Optional[str] = field(default=None)
    learning_rate: float = field(default=1e-6)
    beta: float = field(default=0.04)
    num_iterations: int = field(default=1)
    epsilon: float = field(default=0.2)
    epsilon_high: Optional[float] = field(default=None)
    reward_weights: Optional[List[float]] = field(default=None)
    sync_ref_model: bool = field(default=False)
    ref_model_mixup_alpha: float = field(default=0.6)
    ref_model_sync_steps: int = field(default=512)
    log_completions: bool = field(default=True)
    logging_first_step: bool = field(default=False)
    logging_steps: float = field(default=2)
    # ---- not in the reference: None (default) runs the prompt rows of a group once in the policy pass when that is IDENTICAL to the
    # reference's full-row pass (lora_dropout = 0) and falls back to the full-row pass — an independent LoRA-dropout mask per copy, exactly
    # as the reference draws them (grpo_trainer.py:777-779) — when the adapters have dropout.  True forces the shared pass (one mask
    # stream for the shared rows: unbiased, lower gradient variance, NOT the reference's sampling scheme), False the
    # full-row pass.
    share_policy_prompt: Optional[bool] = field(default=None)
    # ---- not in the reference: stream fp8 (e4m3, one scale per output row) images of the merged weights in the rollout's token loop
    # (BASELINE config 5; bioreason_amd.trainer.GRPOConfig.rollout_fp8)
    rollout_fp8: bool = field(default=False)
    # ---- not in the reference: > 1 = `LengthBucketedRepeatSampler` — the global batches of a window of this many batches hold prompts of
    # similar expected cost (prompt + DNA length, refined by observed completion lengths), so the ranks of a step finish closer together
    # at the gather barrier; 1 (default) = the reference's RepeatRandomSampler, index for index
    length_bucket_batches: int = field(default=1)


def _lora_fields(peft_config) -> Optional[Dict[str, Any]]:
    if peft_config is None:
        return None
    get = (lambda k, d: peft_config.get(k, d)) if isinstance(peft_config, dict) else (lambda k, d: getattr(peft_config, k, d))
    return {"r": int(get("r", 32)), "alpha": float(get("lora_alpha", 64)), "dropout": float(get("lora_dropout", 0.0))}


# ------------------------------------------------------------------------------------------------- trainer (:206-904)
class DNALLMGRPOTrainer:
    def __init__(self, model, reward_funcs: Union[Callable, List[Callable]], args: DNALLMGRPOConfig = None, dna_module=None,
                 train_dataset=None, eval_dataset=None, processing_class=None, reward_processing_classes=None,
                 callbacks: Optional[list] = None, optimizers=(None, None), peft_config=None,
                 freeze_dna_modules: Optional[bool] = False, attn_implementation: str = "flash_attention_2",
                 torch_dtype: str = "bfloat16", **kwargs):
        assert not isinstance(model, str), "model must NOT be a string in the current implementation"      # :243
        if args is None:
            args = DNALLMGRPOConfig(output_dir="DNALLM-GRPO", report_to="none")
        if dna_module is None:
            from .dna_modules import NucleotideDNAModule
            dna_module = NucleotideDNAModule()
        self.args, self.model, self.dna_module = args, model, dna_module
        self.train_dataset, self.eval_dataset = train_dataset, eval_dataset
        self.callbacks = list(callbacks or [])
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # LoRA on every text-model linear outside the DNA modules (:253-274)
        lf = _lora_fields(peft_config)
        if lf is not None and not any("lora_" in n for n, _ in model.text_model.named_parameters()):
            model.text_model.apply_lora(r=lf["r"], alpha=lf["alpha"], dropout=lf["dropout"], arena=model.arena)
        for p in model.dna_model.parameters():                      # the encoder never trains (dna_llm.py:121)
            p.requires_grad_(False)
        # processing class (:318-337)
        if processing_class is None:
            processing_cls = dna_module.get_processing_class()
            processing_class = processing_cls(tokenizer=model.text_tokenizer, dna_tokenizer=model.dna_tokenizer)
            for component, keyword in dna_module.get_custom_processing_keywords():
                if keyword in kwargs:
                    target = getattr(processing_class, component, processing_class)
                    setattr(target, keyword, kwargs[keyword])
        tok = getattr(processing_class, "tokenizer", None) or processing_class
        pad_token_id = getattr(tok, "pad_token_id", None)
        processing_class.pad_token_id = pad_token_id
        processing_class.eos_token_id = getattr(tok, "eos_token_id", None)
        self.processing_class = processing_class
        dna_module.post_model_init(model, processing_class)
        if not isinstance(reward_funcs, list):
            reward_funcs = [reward_funcs]
        for f in reward_funcs:
            if not callable(f) or isinstance(f, (str, torch.nn.Module)):
                raise NotImplementedError("reward models are out of scope: reward_funcs must be python callables (reason.py:312-320)")
        self.reward_funcs = reward_funcs
        # generation settings (:384-391): the warpers are fixed in the reference, not read from args
        self.max_completion_length, self.num_generations = args.max_completion_length, args.num_generations
        eos = processing_class.eos_token_id
        if hasattr(dna_module, "get_eos_token_id"):
            eos = dna_module.get_eos_token_id(processing_class)
        self.generation_config = SimpleNamespace(max_new_tokens=self.max_completion_length, do_sample=True, temperature=0.6,
                                                 top_p=0.95, top_k=20, pad_token_id=pad_token_id, eos_token_id=eos)
        self.beta, self.epsilon_low = args.beta, args.epsilon
        self.epsilon_high = args.epsilon_high if args.epsilon_high is not None else args.epsilon
        self.num_iterations = args.num_iterations
        model.warnings_issued["estimate_tokens"] = True
        self._metrics = defaultdict(list)
        self.log_history: List[Dict[str, float]] = []
        self.log_completions = args.log_completions
        # the global batch must hold whole groups (:428-437)
        global_batch = args.per_device_train_batch_size * self.world
        possible = [n for n in range(2, global_batch + 1) if global_batch % n == 0]
        if self.num_generations not in possible:
            raise ValueError(f"The global train batch size ({self.world} x {args.per_device_train_batch_size}) must be evenly "
                             f"divisible by the number of generations per prompt ({self.num_generations}). Given the current train "
                             f"batch size, the valid values for the number of generations are: {possible}.")
        cfg = GRPOConfig(num_generations=self.num_generations, max_completion_length=self.max_completion_length,
                         temperature=0.6, top_p=0.95, top_k=20, beta=args.beta, epsilon=args.epsilon, epsilon_high=args.epsilon_high,
                         num_iterations=args.num_iterations, gradient_accumulation_steps=args.gradient_accumulation_steps,
                         learning_rate=args.learning_rate, weight_decay=args.weight_decay, adam_beta1=args.adam_beta1,
                         adam_beta2=args.adam_beta2, adam_epsilon=args.adam_epsilon, max_grad_norm=args.max_grad_norm,
                         eos_token_id=self.processing_class.eos_token_id, pad_token_id=pad_token_id,
                         seed=args.seed, share_policy_prompt=getattr(args, "share_policy_prompt", None),
                         rollout_fp8=bool(getattr(args, "rollout_fp8", False)))
        self.runner = GRPOStepRunner(model, cfg)
        self.state = SimpleNamespace(global_step=0, epoch=0.0, log_history=self.log_history)
        model.train()                                               # HF Trainer.training_step puts the model in train mode

    # ---- the reference's methods, same names ---------------------------------------------------------------------------
    def _get_train_sampler(self) -> RepeatRandomSampler:            # :883-897
        a = self.args
        effective = a.per_device_train_batch_size * self.world * a.gradient_accumulation_steps
        nbk = int(getattr(a, "length_bucket_batches", 1) or 1)
        if nbk > 1:
            if getattr(self, "_prompt_costs", None) is None:
                self._prompt_costs = [prompt_cost(self.train_dataset[i]) for i in range(len(self.train_dataset))]
            return LengthBucketedRepeatSampler(self.train_dataset, mini_repeat_count=self.num_generations,
                                               batch_size=effective // self.num_generations, repeat_count=self.num_iterations, seed=a.seed,
                                               costs=self._prompt_costs, bucket_batches=nbk)
        return RepeatRandomSampler(self.train_dataset, mini_repeat_count=self.num_generations,
                                   batch_size=effective // self.num_generations, repeat_count=self.num_iterations, seed=a.seed)

    def _get_per_token_logps(self, model, input_ids, attention_mask, **mm):                        # :510-520
        """[B, L-1] log-probs of input_ids[:, 1:] (the reference's full-width form; the step itself uses the fused
        completion-rows-only form, grpo.per_token_logps)"""
        from . import grpo
        B, L = input_ids.shape
        one = input_ids[:, :1]
        return grpo.per_token_logps(model, one, attention_mask[:, :1], input_ids[:, 1:], attention_mask[:, 1:].to(torch.int32), **mm)

    def _prepare_batch(self, inputs: List[Dict[str, Any]]) -> Dict[str, Any]:
        """dataset rows -> model inputs on the device (:535-571) + the reward closure over this batch's columns (:651-676)"""
        dev = self.model.device
        prompts = [x["prompt"] for x in inputs]
        prompts_text = self.dna_module.prepare_prompt(self.processing_class, inputs)
        batch_dna = []
        for x in inputs:
            d = x.get("dna_sequences")
            assert d is not None, "The key dna_sequences is not found in the input"
            batch_dna.append(list(d) if isinstance(d, (list, tuple)) else [d])
        pi = self.dna_module.prepare_model_inputs(self.processing_class, self.model, prompts_text, batch_dna, return_tensors="pt",
                                                  padding=True, padding_side="left", add_special_tokens=False)
        batch = {"input_ids": pi["input_ids"].to(dev), "attention_mask": pi["attention_mask"].to(dev),
                 "batch_idx_map": list(pi["batch_idx_map"]) if pi.get("batch_idx_map") is not None else [],
                 "dna_tokenized": None}
        if pi.get("dna_tokenized") is not None:
            batch["dna_tokenized"] = {k: v.to(dev) for k, v in pi["dna_tokenized"].items()}
        # rows that repeat a prompt (RepeatRandomSampler emits each index G times in a row) share prefill and encoder work
        first: Dict[str, int] = {}
        alias = []
        for i, t in enumerate(prompts_text):
            key = t + "\x00" + "\x00".join(batch_dna[i])
            alias.append(first.setdefault(key, i))
        if len(set(alias)) < len(alias) and batch["dna_tokenized"] is not None:
            batch["prompt_alias"] = alias
            seq_first: Dict[tuple, int] = {}
            dalias, s = [], 0
            for i, seqs in enumerate(batch_dna):
                for j in range(len(seqs)):
                    dalias.append(seq_first.setdefault((alias[i], j), s))
                    s += 1
            batch["dna_alias"] = dalias
        extra = {k: [ex[k] for ex in inputs] for k in inputs[0].keys() if k not in ("prompt", "completion")}
        self.runner.reward_fn = text_reward_fn(self.processing_class, self.reward_funcs, prompts=prompts, extra=extra)
        return batch

    def training_step(self, inputs: List[Dict[str, Any]]) -> torch.Tensor:
        out = self.runner.step(self._prepare_batch(inputs))
        if "metrics_t" in out:
            vals = out["metrics_t"].tolist()
            for k, v in zip(self.runner.metric_names, vals):
                if k != "loss":
                    self._metrics[k].append(v)
            for f, v in zip(self.reward_funcs, out["rewards_per_func_t"].tolist()):
                self._metrics[f"rewards/{f.__name__}"].append(v)
            self.state.global_step = self.runner.global_step
        return out["loss_t"]

    def log(self, logs: Dict[str, float], start_time: Optional[float] = None) -> None:             # :816-823
        metrics = {k: sum(v) / len(v) for k, v in self._metrics.items()}
        logs = {**logs, **metrics, "step": self.state.global_step}
        self.log_history.append(logs)
        if self.rank == 0:
            print(logs, flush=True)
        self._metrics.clear()

    def save_model(self, output_dir: Optional[str] = None) -> None:
        """pytorch_model.bin with the reference's key names (SaveWithPyTorchCallback, reason.py:46-81)"""
        if self.rank != 0:
            return
        output_dir = output_dir or self.args.output_dir
        os.makedirs(output_dir, exist_ok=True)
        torch.save(self.model.state_dict(), os.path.join(output_dir, "pytorch_model.bin"))

    def _lr_schedule(self, num_training_steps: int) -> Callable[[int], float]:
        """optimiser step -> learning rate exactly as HF `Trainer.create_scheduler` would set it: transformers' own
        `get_scheduler(args.lr_scheduler_type, ..., args.get_warmup_steps(n), n)` stepped on a one-parameter stand-in
        optimiser (the real update is the fused arena AdamW, which takes the rate per call)"""
        from transformers.optimization import get_scheduler
        a = self.args
        opt = torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=a.learning_rate)
        sched = get_scheduler(a.lr_scheduler_type, optimizer=opt, num_warmup_steps=a.get_warmup_steps(num_training_steps),
                              num_training_steps=num_training_steps,
                              scheduler_specific_kwargs=getattr(a, "lr_scheduler_kwargs", None) or {})
        rates: List[float] = []

        def at(step: int) -> float:
            while len(rates) <= step:                # rate in force for optimiser step len(rates), then advance as HF does
                rates.append(float(sched.get_last_lr()[0]))
                opt.step()
                sched.step()
            return rates[step]
        return at

    def train(self, resume_from_checkpoint=None):
        a = self.args
        per_rank = a.per_device_train_batch_size
        ga = max(1, a.gradient_accumulation_steps)
        max_steps = a.max_steps if a.max_steps and a.max_steps > 0 else None
        t0 = time.time()
        done = False
        epochs = int(a.num_train_epochs) if max_steps is None else 10 ** 9
        # ONE sampler for the whole run (Trainer.get_train_dataloader builds it once): its torch.Generator carries over, so every
        # epoch is a new permutation of the same seeded stream, as in the reference
        sampler = self._get_train_sampler()
        gmb_ = per_rank * self.world
        steps_per_epoch = max(1, (len(sampler) // gmb_) // ga)
        total = max_steps if max_steps is not None else max(1, math.ceil(a.num_train_epochs * steps_per_epoch))   # (HF Trainer: math.ceil)
        self.runner.lr_schedule = self._lr_schedule(total)
        for epoch in range(epochs):
            stream = list(iter(sampler))
            # accelerate's sharding of the sampler stream: consecutive global micro-batches, rank r takes slice r
            gmb = per_rank * self.world
            for lo in range(0, len(stream) - gmb + 1, gmb):
                mine = stream[lo + self.rank * per_rank: lo + (self.rank + 1) * per_rank]
                rows = [self.train_dataset[i] for i in mine]
                step_before = self.runner.global_step
                loss = self.training_step(rows)
                if self.runner.global_step != step_before:
                    gs = self.runner.global_step
                    if a.logging_steps and (gs % max(1, int(a.logging_steps)) == 0 or (a.logging_first_step and gs == 1)):
                        # HF logs `_get_learning_rate()` AFTER `lr_scheduler.step()`: the rate of the NEXT optimiser step
                        lr_next = self.runner.lr_schedule(gs) if self.runner.lr_schedule is not None else self.runner.last_lr
                        self.log({"loss": float(loss), "learning_rate": float(lr_next),
                                  "epoch": epoch + lo / max(1, len(stream))})
                    if a.save_steps and a.save_strategy != "no" and gs % int(a.save_steps) == 0:
                        for cb in self.callbacks:
                            if hasattr(cb, "on_save"):
                                cb.on_save(a, self.state, SimpleNamespace(), model=self.model)
                    if max_steps is not None and gs >= max_steps:
                        done = True
                        break
            if done:
                break
        return SimpleNamespace(global_step=self.runner.global_step, training_loss=None,
                               metrics={"train_runtime": time.time() - t0})
