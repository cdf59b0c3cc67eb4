"""Synthetic DNA + prompt batches of the shapes BASELINE.json quotes: no datasets or tokenizers
exist offline, so token ids are drawn directly in the form DLProcessor would emit
(bioreason/models/dl/processing_dl.py:134-221): per sample `n_dna` DNA sequences of Sd NT tokens (<cls> first, no
padding) and a text of `text_len` ordinary ids with  <|dna_start|> <|dna_pad|> x Sd <|dna_end|>  per sequence
inserted after the first 8 tokens, i.e. P = n_dna * (Sd + 2) + text_len (= 2180 for Sd = 1024, text 128)."""
from __future__ import annotations

from typing import Dict, List

import torch


def synth_prompt_batch(B: int, n_unique: int, Sd: int = 1024, text_len: int = 128, n_dna: int = 2, dna_token_id: int = 151670,
                       vocab_text: int = 151643, vocab_dna: int = 4107, device="cpu", seed: int = 42) -> Dict:
    """B samples made of `n_unique` distinct prompts, each repeated B // n_unique times consecutively (the layout
    RepeatRandomSampler produces for GRPO, grpo_trainer.py:107-116).  `dna_alias[s]` = first sequence identical to s."""
    assert B % n_unique == 0
    rep = B // n_unique
    g = torch.Generator().manual_seed(seed)
    u_dna = torch.randint(6, vocab_dna, (n_unique * n_dna, Sd), generator=g)
    u_dna[:, 0] = 3
    u_txt = torch.randint(0, vocab_text, (n_unique, text_len), generator=g)
    rows, dna_rows, alias, bmap = [], [], [], []
    for b in range(B):
        u = b // rep
        head, tail = u_txt[u, :8].tolist(), u_txt[u, 8:].tolist()
        mid: List[int] = []
        for j in range(n_dna):
            mid += [dna_token_id - 1] + [dna_token_id] * Sd + [dna_token_id + 1]
            dna_rows.append(u_dna[u * n_dna + j])
            alias.append((u * rep) * n_dna + j)           # index of the first copy of this sequence
            bmap.append(b)
        rows.append(head + mid + tail)
    ids = torch.tensor(rows, dtype=torch.long)
    dna = torch.stack(dna_rows)
    return {"input_ids": ids.to(device), "attention_mask": torch.ones_like(ids).to(device),
            "dna_tokenized": {"input_ids": dna.to(device), "attention_mask": torch.ones_like(dna).to(device)},
            "batch_idx_map": bmap, "dna_alias": alias, "prompt_alias": [(b // rep) * rep for b in range(B)]}


class SyntheticTokenizer:
    """Stand-in for the Qwen3 tokenizer in benchmarks (no tokenizer files exist offline): a fixed id -> text-piece table,
    so that the reward hop of the reference — completion ids to the host, `batch_decode(..., skip_special_tokens=True)`,
    python / regex reward functions over the decoded text, rewards back to the device (grpo_trainer.py:642-676) — runs
    for real inside the timed step.  Pieces are drawn from the strings the reward regexes look for (reason.py:193-245)
    and ordinary words, so every reward function does work on every completion."""
    PIECES = ["<think>", "</think>", "<answer>", "</answer>", "\n", " ", "the", " gene", " variant", " pathway", " therefore",
              " 1", " 2", " 3", " disease", " protein", ".", ",", " is", " of", " in", " A", " C", " G", " T", " KEGG"]

    def __init__(self, vocab_size: int = 151936, pad_token_id: int = 151643, eos_token_id: int = 151645):
        self.vocab_size, self.pad_token_id, self.eos_token_id = vocab_size, pad_token_id, eos_token_id
        self.special = {pad_token_id, eos_token_id}
        n = len(self.PIECES)
        self._table = [self.PIECES[(i * 2654435761) % n] for i in range(vocab_size)]

    def batch_decode(self, ids, skip_special_tokens: bool = True):
        if isinstance(ids, torch.Tensor):
            ids = ids.tolist()
        tab, sp = self._table, self.special
        return ["".join(tab[i] for i in row if not (skip_special_tokens and i in sp)) for row in ids]
