"""The reference's GRPO reward functions (reason.py:117-126, 193-230; registry :312-320) — SURVEY §8 row a13.
CPU string work on the decoded completions; it stays Python in the reference and here.  Same names, same arguments
(`completions` = one list per sample holding a `{"content": str}` message), same values; the reference's debug print
is behind `verbose`.

Kept as the reference has it: `correctness_reward_func` pairs the i-th response with the i-th ELEMENT of `answer[0]`
(`zip(extracted_responses, answer[0])`, reason.py:200) — with a string answer that is its i-th character."""
import re
from typing import Callable, Dict, List, Optional

import torch


def extract_xml_answer(text: str) -> str:                      # reason.py:117-121
    return text.split("</think>")[-1].strip()


def extract_hash_answer(text: str) -> Optional[str]:           # reason.py:123-126
    if "####" not in text:
        return None
    return text.split("####")[1].strip()


def correctness_reward_func(prompts, completions, answer, verbose: bool = False, **kwargs) -> List[float]:
    responses = [c[0]["content"] for c in completions]
    extracted = [extract_xml_answer(r) for r in responses]
    if verbose:
        q = prompts[0][-1]["content"]
        print("-" * 20, f"Question:\n{q}", f"\nAnswer:\n{answer[0]}", f"\nResponse:\n{responses[0]}", f"\nExtracted:\n{extracted[0]}")
    return [2.0 if a.lower() in r.lower() else 0.0 for r, a in zip(extracted, answer[0])]


def less_than_4_reward_func(completions, **kwargs) -> List[float]:
    extracted = [extract_xml_answer(c[0]["content"]) for c in completions]
    return [0.5 if len(r.split(" ")) <= 4 else 0.0 for r in extracted]


def strict_format_reward_func(completions, **kwargs) -> List[float]:
    pattern = r"^<think>\n.*?\n</think>\n.*?\n$"
    return [0.5 if re.match(pattern, c[0]["content"]) else 0.0 for c in completions]


def soft_format_reward_func(completions, **kwargs) -> List[float]:
    pattern = r"<think>.*?</think>\s*.*?"
    return [0.5 if re.match(pattern, c[0]["content"]) else 0.0 for c in completions]


def count_xml(text: str) -> float:
    count = 0.0
    if text.count("<think>\n") == 1:
        count += 0.125
    if text.count("\n</think>\n") == 1:
        count += 0.125
    return count


def xmlcount_reward_func(completions, **kwargs) -> List[float]:
    return [count_xml(c[0]["content"]) for c in completions]


reward_funcs_registry: Dict[str, Callable] = {
    "xmlcount": xmlcount_reward_func,
    "soft_format": soft_format_reward_func,
    "strict_format": strict_format_reward_func,
    "less_than_4": less_than_4_reward_func,
    "correctness": correctness_reward_func,
}


def reward_hop(processing_class, reward_funcs: List[Callable], prompts=None, extra: Optional[Dict[str, list]] = None) -> Callable:
    """THE reward hop of the reference (grpo_trainer.py:643-676), the one implementation both entry points below use:
    completion ids -> host -> `batch_decode(..., skip_special_tokens=True)` -> python reward functions -> [B, F] fp32 back on
    the device.  Completions are wrapped as one-message conversations only for conversational prompts
    (`is_conversational(inputs[0])`, :646-649: the prompt is a list of role / content messages); plain-string prompts hand the
    reward functions plain strings.  `prompts` None (synthetic runs without a prompt column): the conversational shape, which is
    what reason.py's reward functions take."""
    conversational = bool(prompts) and isinstance(prompts[0], (list, tuple)) and len(prompts[0]) > 0 \
        and isinstance(prompts[0][0], dict) and "role" in prompts[0][0] and "content" in prompts[0][0]
    if prompts is None:
        conversational = True

    def fn(completion_ids: torch.Tensor, completion_mask: torch.Tensor) -> torch.Tensor:
        texts = processing_class.batch_decode(completion_ids.cpu().tolist(), skip_special_tokens=True)
        completions = [[{"role": "assistant", "content": t}] for t in texts] if conversational else texts
        cols = [f(prompts=prompts, completions=completions, **(extra or {})) for f in reward_funcs]
        return torch.tensor(cols, dtype=torch.float32).t().contiguous().to(completion_ids.device)
    return fn


def text_reward_fn(tokenizer, names: List[str], prompts=None, answer=None) -> Callable:
    """`reward_hop` for reward functions named as in reason.py's registry (:312-320), with the `answer` dataset column
    (`GRPOStepRunner(reward_fn=...)` of synthetic runs; a list of None prompts counts as no prompt column)"""
    if prompts is not None and all(p is None for p in prompts):
        prompts = None
    return reward_hop(tokenizer, [reward_funcs_registry[n] for n in names], prompts=prompts, extra={"answer": answer})
