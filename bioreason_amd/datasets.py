"""Input-side data formats of the hot path: what `train_dna_qwen.py:27-33` and `reason.py` feed the collate / trainer.

These are the record formats either side of `DLProcessor` (the callers of rows a1 / N1), restated so that the
reference's entry scripts resolve every `bioreason.dataset.*` name with this repository first on PYTHONPATH:

  * `truncate_dna`                                   bioreason/dataset/utils.py:6-20
  * `torch_to_hf_dataset`                            bioreason/dataset/utils.py:23-59
  * `KEGGDataset`, `split_kegg_dataset`, `create_kegg_dataloader`
                                                     bioreason/dataset/kegg.py:14-146
  * `get_format_kegg_function` (+ the two formatters)        kegg.py:149-220
  * `get_format_variant_effect_function`, `clean_variant_effect_example`, `clean_variant_effect_non_snv_example`
                                                     bioreason/dataset/variant_effect.py:14-97

All of it is string / dict work on the host (no kernels); `tests/test_datasets.py` holds every function equal to the reference's
own function object (ast-extracted, so that the absent `trl` import of those files is not needed) on seeded records, and equal
to committed golden records on the GPU box where /root/reference does not exist.

One builder makes all four chat records: a conversation is (2 DNA slots, the question — with or without the sequences spelled
out in text —, the assistant's reasoning, "Answer: ...").  The four reference formatters differ only in those three choices.
"""
from __future__ import annotations

import json
import os
import random
from typing import Any, Callable, Dict, List, Tuple

import torch
from torch.utils.data import DataLoader, Dataset

SEQUENCE_KEYS = ("reference_sequence", "variant_sequence")


# ---------------------------------------------------------------------------------------------- sequences
def truncate_dna(example: Dict[str, Any], truncate_dna_per_side: int = 1024) -> Dict[str, Any]:
    """drop `truncate_dna_per_side` bases from both ends of the two sequences, but only where more than 8 bases would be
    left (utils.py:16-18: `len > 2 * n + 8`); shorter sequences are kept whole.  Mutates and returns `example`."""
    n = truncate_dna_per_side
    for key in SEQUENCE_KEYS:
        seq = example[key]
        if len(seq) > 2 * n + 8:
            example[key] = seq[n:-n]        # (n == 0 gives seq[0:-0] == "" in the reference too)
    return example


def torch_to_hf_dataset(torch_dataset):
    """a map-style torch Dataset as a `datasets.Dataset`: dict items become columns (the FIRST item's keys decide which),
    anything else goes to one column `data` (utils.py:23-59)"""
    from datasets import Dataset as HFDataset
    n = len(torch_dataset)
    if n == 0:
        return HFDataset.from_dict({})
    items = [torch_dataset[i] for i in range(n)]
    if isinstance(items[0], dict):
        cols = {k: [it[k] for it in items] for k in items[0].keys()}
    else:
        cols = {"data": items}
    return HFDataset.from_dict(cols)


# ---------------------------------------------------------------------------------------------- chat records
def _chat_record(question: str, reasoning: str, answer_text: str, dna: List[str], answer_field: str) -> Dict[str, Any]:
    """{"prompt": [user(2 x dna slot, text), assistant(reasoning_content, "Answer: ...")], "dna_sequences", "answer"}"""
    user = [{"type": "dna", "text": None}, {"type": "dna", "text": None}, {"type": "text", "text": question.strip()}]
    return {
        "prompt": [
            {"role": "user", "content": user},
            {"role": "assistant", "reasoning_content": reasoning, "content": [{"type": "text", "text": answer_text}]},
        ],
        "dna_sequences": list(dna),
        "answer": answer_field,
    }


def _spelled_out(example: Dict[str, Any]) -> str:
    """the text-only ("llm") question: both sequences in the text, empty DNA slots (kegg.py:193, variant_effect.py:70)"""
    return (f"Reference sequence: {example['reference_sequence']}\nVariant sequence: {example['variant_sequence']}\n"
            f"Question: {example['question']}")


def format_kegg_for_dna_llm(example: Dict[str, Any]) -> Dict[str, Any]:
    """kegg.py:162-187 — reasoning = the record's reasoning, `answer` passed through unstripped"""
    return _chat_record(example["question"], example["reasoning"].strip(), f"Answer: {example['answer'].strip()}",
                        [example[k] for k in SEQUENCE_KEYS], example["answer"])


def format_kegg_for_llm(example: Dict[str, Any]) -> Dict[str, Any]:
    """kegg.py:190-220"""
    return _chat_record(_spelled_out(example), example["reasoning"].strip(), f"Answer: {example['answer'].strip()}",
                        ["", ""], example["answer"])


def format_variant_effect_for_dna_llm(example: Dict[str, Any]) -> Dict[str, Any]:
    """variant_effect.py:42-66 — no reasoning trace in VEP: the reasoning content IS the answer line; `answer` stripped"""
    line = f"Answer: {example['answer'].strip()}"
    return _chat_record(example["question"], line, line, [example[k] for k in SEQUENCE_KEYS], example["answer"].strip())


def format_variant_effect_for_llm(example: Dict[str, Any]) -> Dict[str, Any]:
    """variant_effect.py:69-97"""
    line = f"Answer: {example['answer'].strip()}"
    return _chat_record(_spelled_out(example), line, line, ["", ""], example["answer"].strip())


_FORMATTERS: Dict[str, Dict[str, Callable]] = {
    "kegg": {"llm": format_kegg_for_llm, "dna-llm": format_kegg_for_dna_llm},
    "variant_effect": {"llm": format_variant_effect_for_llm, "dna-llm": format_variant_effect_for_dna_llm},
}


def _pick(task: str, model_name: str) -> Callable:
    try:
        return _FORMATTERS[task][model_name.lower()]
    except KeyError:
        raise ValueError(f"Unsupported model name: {model_name}") from None


def get_format_kegg_function(model_name: str) -> Callable:
    """kegg.py:149-159: "llm" | "dna-llm" (case-insensitive), ValueError otherwise"""
    return _pick("kegg", model_name)


def get_format_variant_effect_function(model_name: str) -> Callable:
    """variant_effect.py:14-23"""
    return _pick("variant_effect", model_name)


def clean_variant_effect_example(example: Dict[str, Any]) -> Dict[str, Any]:
    """answer := first ';'-separated field, stripped, lower-cased (variant_effect.py:26-31)"""
    example["answer"] = example["answer"].split(";")[0].strip().lower()
    return example


_NON_SNV_DROP = str.maketrans({"[": None, "]": None, "'": None, "_": " "})


def clean_variant_effect_non_snv_example(example: Dict[str, Any]) -> Dict[str, Any]:
    """answer := the list's repr without brackets / quotes, '_' -> ' ', stripped (variant_effect.py:34-39)"""
    example["answer"] = example["answer"].translate(_NON_SNV_DROP).strip()
    return example


# ---------------------------------------------------------------------------------------------- KEGG json directory
class KEGGDataset(Dataset):
    """every `*.json` of a directory (sorted by file name) as one record {question, answer (lower, stripped), reasoning (steps
    joined by newlines), reference_sequence / variant_sequence (upper, stripped)} — kegg.py:14-78.  The file name's second
    '_'-separated field is read as the KEGG id (kegg.py:32) — and, as in the reference, not kept in the record."""

    def __init__(self, data_dir: str):
        self.data_dir = data_dir
        self.data: List[Dict[str, Any]] = []
        for name in sorted(f for f in os.listdir(data_dir) if f.endswith(".json")):
            _ = name.split("_")[1]                                 # IndexError on a malformed name, like the reference
            with open(os.path.join(data_dir, name), "r", encoding="utf-8") as fh:
                self.data.append(self._process_item(json.load(fh)))

    @staticmethod
    def _process_item(item: Dict[str, Any]) -> Dict[str, Any]:
        steps = item.get("reasoning", {}).get("reasoning_steps", [])
        return {
            "question": item.get("question", ""),
            "answer": item.get("answer", "").lower().strip(),
            "reasoning": "\n".join(steps),
            "reference_sequence": item.get("reference_sequence", "").upper().strip(),
            "variant_sequence": item.get("variant_sequence", "").upper().strip(),
        }

    def __len__(self) -> int:
        return len(self.data)

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        return self.data[idx]


def split_kegg_dataset(dataset, train_ratio: float = 0.8, val_ratio: float = 0.1, test_ratio: float = 0.1, seed: int = 42) -> Tuple:
    """seeded `random_split` into int(ratio * n) / int(ratio * n) / the rest (kegg.py:81-117; seeds torch AND random)"""
    n = len(dataset)
    n_train, n_val = int(train_ratio * n), int(val_ratio * n)
    assert train_ratio + val_ratio + test_ratio == 1.0, "Ratios must sum to 1"
    torch.manual_seed(seed)
    random.seed(seed)
    return tuple(torch.utils.data.random_split(dataset, [n_train, n_val, n - n_train - n_val]))


def create_kegg_dataloader(data_dir: str, batch_size: int = 2, shuffle: bool = True, num_workers: int = 2, pin_memory: bool = True) -> DataLoader:
    """kegg.py:120-146"""
    return DataLoader(KEGGDataset(data_dir), batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, pin_memory=pin_memory)


def dna_collate_fn(batch: List[Dict[str, Any]], dna_tokenizer: Any, label2id: Dict[str, int], max_length: int = 2048) -> Dict[str, Any]:
    """the DNA-only classifier's collate (kegg.py:336-382): the two sequences tokenised separately + integer labels.  Kept for the
    import surface of `bioreason.dataset.kegg`; the classifier itself is outside SURVEY §8."""
    kw = dict(padding=True, truncation=True, max_length=max_length, return_tensors="pt")
    ref = dna_tokenizer([it["reference_sequence"] for it in batch], **kw)
    alt = dna_tokenizer([it["variant_sequence"] for it in batch], **kw)
    return {"ref_ids": ref.input_ids, "ref_attention_mask": ref.attention_mask, "alt_ids": alt.input_ids,
            "alt_attention_mask": alt.attention_mask,
            "labels": torch.tensor([label2id[it["answer"]] for it in batch], dtype=torch.long)}
