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


def text_reward_fn(tokenizer, names: List[str], prompts=None, answer=None) -> Callable:
    """adapter for `GRPOStepRunner(reward_fn=...)`: decodes the sampled ids (skip_special_tokens, as
    grpo_trainer.py:642-647 does) and evaluates the named reward functions -> fp32 [B, len(names)]"""
    funcs = [reward_funcs_registry[n] for n in names]

    def fn(completion_ids: torch.Tensor, completion_mask: torch.Tensor) -> torch.Tensor:
        ids = completion_ids.masked_fill(completion_mask == 0, tokenizer.pad_token_id).tolist()
        texts = tokenizer.batch_decode(ids, skip_special_tokens=True)
        completions = [[{"role": "assistant", "content": t}] for t in texts]
        cols = [f(prompts=prompts, completions=completions, answer=answer) for f in funcs]
        return torch.tensor(cols, dtype=torch.float32, device=completion_ids.device).t().contiguous()

    return fn
