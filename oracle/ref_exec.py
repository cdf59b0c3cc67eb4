"""CPU ORACLE support (test infrastructure) — runs the REFERENCE'S OWN statements of the GRPO arithmetic.

`bioreason/trainer/grpo_trainer.py` cannot be imported here (trl / peft / deepspeed are absent), but its arithmetic is
plain torch: this module parses the file, takes the AST nodes of exactly the cited lines and executes them unmodified
against a stub `self`.  Nothing is restated — `oracle/grpo_math.py` (the restatement the GPU tests use) is pinned by
comparing it with what these functions return (tests/test_oracle_pinned.py) and by the fixtures
`oracle/make_grpo_golden.py` writes from them (tests/golden/grpo_ref.pt, which travel to the GPU box).

  per_token_logps     grpo_trainer.py:510-520   DNALLMGRPOTrainer._get_per_token_logps (whole method)
  completion_mask     grpo_trainer.py:605-609   statements inside _generate_and_score_completions
  advantages          grpo_trainer.py:679-699   statements inside _generate_and_score_completions (gather .. local slice)
  compute_loss        grpo_trainer.py:751-814   DNALLMGRPOTrainer.compute_loss (whole method; buffering included)

Only tests/ and oracle/ scripts import this file; it needs /root/reference (build container only).
"""
from __future__ import annotations

import ast
import contextlib
import io
import os
import types
from collections import defaultdict

import torch

REF_TRAINER = "/root/reference/bioreason/trainer/grpo_trainer.py"


def available() -> bool:
    return os.path.exists(REF_TRAINER)


def _trainer_class() -> ast.ClassDef:
    tree = ast.parse(open(REF_TRAINER).read())
    return next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "DNALLMGRPOTrainer")


def _method(name: str):
    """the reference method `name`, compiled from its own source lines, as a plain function f(self, ...)"""
    node = next(n for n in _trainer_class().body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[node], type_ignores=[]), REF_TRAINER, "exec"), ns)
    return ns[name], (node.lineno, node.end_lineno)


def _statements(method: str, first: int, last: int):
    """top-level statements of `method` whose source lines lie inside [first, last], compiled as they stand"""
    node = next(n for n in _trainer_class().body if isinstance(n, ast.FunctionDef) and n.name == method)
    body = [s for s in node.body if s.lineno >= first and s.end_lineno <= last]
    assert body and body[0].lineno >= first, (method, first, last)
    return compile(ast.Module(body=body, type_ignores=[]), REF_TRAINER, "exec"), (body[0].lineno, body[-1].end_lineno)


def per_token_logps(model, input_ids, attention_mask, **multimodal):
    fn, span = _method("_get_per_token_logps")
    assert span == (510, 520), span
    return fn(types.SimpleNamespace(), model, input_ids, attention_mask, **multimodal)


def completion_mask(completion_ids: torch.Tensor, eos_token_id: int) -> torch.Tensor:
    code, span = _statements("_generate_and_score_completions", 598, 609)
    assert span == (605, 609), span
    self = types.SimpleNamespace(processing_class=types.SimpleNamespace(eos_token_id=eos_token_id))
    ns = {"torch": torch, "self": self, "completion_ids": completion_ids, "device": completion_ids.device}
    exec(code, ns)
    return ns["completion_mask"]


def advantages(rewards_per_func_local: torch.Tensor, num_generations: int, gather=None, process_index: int = 0, n_local=None):
    """`gather` plays accelerator.gather (default: identity = one process).  -> (advantages of the local slice, the locals)"""
    code, span = _statements("_generate_and_score_completions", 678, 699)
    assert span == (679, 699), span
    acc = types.SimpleNamespace(gather=gather or (lambda x: x), process_index=process_index)
    self = types.SimpleNamespace(accelerator=acc, num_generations=num_generations)
    n_local = rewards_per_func_local.shape[0] if n_local is None else n_local
    ns = {"torch": torch, "self": self, "rewards_per_func": rewards_per_func_local, "prompts": [None] * n_local}
    exec(code, ns)
    return ns["advantages"], {k: ns[k] for k in ("rewards", "mean_grouped_rewards", "std_grouped_rewards")}


class _FakeModel:
    """stands in for the policy: returns prepared logits (a leaf that requires grad, so the loss can be differentiated)"""

    def __init__(self, logits):
        self.logits = logits

    def __call__(self, input_ids=None, attention_mask=None, **kw):
        return types.SimpleNamespace(logits=self.logits)


def compute_loss(logits, prompt_ids, prompt_mask, completion_ids, completion_mask_, advantages_, ref_per_token_logps=None,
                 old_per_token_logps=None, beta=0.04, epsilon_low=0.2, epsilon_high=0.2, num_iterations=1):
    """runs the reference's compute_loss (and, through it, its _get_per_token_logps) on a fake model that returns
    `logits` [B, P + C, V].  -> (loss, metrics dict of lists as the reference appends them)"""
    fn, span = _method("compute_loss")
    assert span == (751, 814), span
    lp_fn, _ = _method("_get_per_token_logps")
    inputs = {"prompt_ids": prompt_ids, "prompt_mask": prompt_mask, "completion_ids": completion_ids,
              "completion_mask": completion_mask_, "multimodal_inputs": {}, "advantages": advantages_,
              "ref_per_token_logps": ref_per_token_logps, "old_per_token_logps": old_per_token_logps}
    self = types.SimpleNamespace(
        state=types.SimpleNamespace(global_step=0), num_iterations=num_iterations, _step=0,
        args=types.SimpleNamespace(gradient_accumulation_steps=1), _buffered_inputs=[None],
        _generate_and_score_completions=lambda inp, model: inp, beta=beta, epsilon_low=epsilon_low,
        epsilon_high=epsilon_high, _metrics=defaultdict(list),
        accelerator=types.SimpleNamespace(gather_for_metrics=lambda x: x))
    self._get_per_token_logps = lambda model, ids, am, **kw: lp_fn(self, model, ids, am, **kw)
    with contextlib.redirect_stdout(io.StringIO()):            # the method prints "index 1" .. debug lines
        loss = fn(self, _FakeModel(logits), inputs)
    return loss, dict(self._metrics)
