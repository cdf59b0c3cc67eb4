"""SFT collate pieces of the reference on the host side of the path.

`assistant_label_mask` is `qwen_dna_collate_fn`'s label construction (bioreason/dataset/kegg.py:252-327): labels are the
input ids inside every assistant section — from just after an `<|im_start|>assistant\\n` marker to the next `<|im_end|>`
marker (or the end of the row) — and -100 elsewhere and on padding.  The reference scans every row position by position
in Python (O(B * L * marker_len) tensor comparisons, each a device sync when the batch lives on a GPU); this is the same
function as three shifted comparisons, one cumulative maximum and one cumulative sum over the whole batch at once.
"""
from typing import Sequence

import torch


def _marker_hits(ids: torch.Tensor, marker: Sequence[int]) -> torch.Tensor:
    """[B, L] bool: position p starts an occurrence of `marker`"""
    B, L = ids.shape
    n = len(marker)
    hit = torch.zeros((B, L), dtype=torch.bool, device=ids.device)
    if n == 0 or n > L:
        return hit
    ok = torch.ones((B, L - n + 1), dtype=torch.bool, device=ids.device)
    for j, tok in enumerate(marker):
        ok &= ids[:, j:L - n + 1 + j] == int(tok)
    hit[:, :L - n + 1] = ok
    return hit


def assistant_label_mask(input_ids: torch.Tensor, assistant_marker_ids: Sequence[int], end_marker_ids: Sequence[int],
                         pad_token_id: int) -> torch.Tensor:
    """labels [B, L] (int64) of kegg.py:252-327 for already tokenised, padded rows"""
    ids = input_ids
    B, L = ids.shape
    pos = torch.arange(L, device=ids.device)[None, :].expand(B, L)
    # section starts = position just after a start marker (kegg.py:283-290); may equal L (marker at the very end)
    s_hit = _marker_hits(ids, assistant_marker_ids)
    start_at = torch.zeros((B, L + 1), dtype=torch.bool, device=ids.device)
    n = len(assistant_marker_ids)
    if 0 < n <= L:
        start_at[:, n:] = s_hit[:, :L - n + 1]
    start_at = start_at[:, :L]
    e_hit = _marker_hits(ids, end_marker_ids)                                    # kegg.py:293-299
    # last start <= q, and the number of end markers in (last_start, q]: a section is open at q iff that count is 0
    # (kegg.py:302-313: the section of a start runs to the first end marker strictly after it, else to the row end)
    last_start = torch.where(start_at, pos, torch.full_like(pos, -1)).cummax(dim=1).values
    ends_upto = e_hit.long().cumsum(dim=1)                                       # E(q) = #ends at positions <= q
    e_at_start = torch.gather(ends_upto, 1, last_start.clamp(min=0))
    open_ = (last_start >= 0) & (ends_upto - e_at_start == 0)
    labels = torch.where(open_, ids, torch.full_like(ids, -100))
    labels = torch.where(ids == int(pad_token_id), torch.full_like(ids, -100), labels)      # kegg.py:321-322
    return labels.to(torch.long)


def qwen_dna_collate_fn(examples, processor, max_length_text: int, max_length_dna: int, return_answer_in_batch: bool = False):
    """`qwen_dna_collate_fn` of the reference (bioreason/dataset/kegg.py:223-333): chat template over each example's
    conversation -> DLProcessor (left padding) -> labels on the assistant sections only (`assistant_label_mask`) [-> answers].
    trl's `maybe_apply_chat_template` (-> `apply_chat_template`, trl/data_utils.py) is restated for the shape the reference
    feeds it, a {"prompt": messages} example: the last message decides — `continue_final_message=True` when it is the
    assistant's (the SFT / kegg shape: the text ends AT the assistant content, no closing `<|im_end|>`), and
    `add_generation_prompt=True` when it is the user's."""
    tok = processor.tokenizer
    kw = {"chat_template": processor.chat_template} if getattr(processor, "chat_template", None) is not None else {}

    def render(messages):
        last = messages[-1]["role"] if len(messages) else None
        if last == "assistant":
            return tok.apply_chat_template(messages, tokenize=False, continue_final_message=True, **kw)
        if last == "user":
            return tok.apply_chat_template(messages, tokenize=False, add_generation_prompt=True, **kw)
        return tok.apply_chat_template(messages, tokenize=False, **kw)

    prompts_text = [ex["prompt"] if isinstance(ex["prompt"], str) else render(ex["prompt"]) for ex in examples]
    batch = processor(text=prompts_text, batch_dna_sequences=[ex["dna_sequences"] for ex in examples], return_tensors="pt",
                      padding=True, padding_side="left", add_special_tokens=False, max_length_text=max_length_text,
                      max_length_dna=max_length_dna)
    start_ids = tok.encode("<|im_start|>assistant\n", add_special_tokens=False)
    end_ids = tok.encode("<|im_end|>", add_special_tokens=False)
    batch["labels"] = assistant_label_mask(batch["input_ids"], start_ids, end_ids, tok.pad_token_id)
    if return_answer_in_batch:
        batch["answer"] = [ex["answer"].strip() for ex in examples]
    return batch
