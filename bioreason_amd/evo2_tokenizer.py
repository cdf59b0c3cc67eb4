"""Evo2's byte-level DNA tokenizer behind the HuggingFace tokenizer interface (first slice).

Reference: `bioreason/models/evo2_tokenizer.py:16-218` (`Evo2Tokenizer`, `register_evo2_tokenizer`), constructed at
`bioreason/models/dna_llm.py:87` around `Evo2(...).tokenizer` — vortex's `CharLevelTokenizer(512)`.  `evo2` / `vortex` are
neither under /root/reference nor installed (unpinned, absent), so `CharLevelTokenizer` below restates the
published algorithm — one token per UTF-8 byte, `eod = eos = 0`, `pad = 1`, decoding clamps an id into [32, vocab] before
`chr` — and is the ONLY part of this file whose parity is unpinned; the wrapper is held equal to the reference's own class
(fed the same inner tokenizer) by `tests/test_evo2_tokenizer.py` and by the committed records `tests/golden/evo2_tokenizer.json`.

What callers rely on (`DLProcessor.tokenize_dna_sequences`, processing_dl.py:87-132; `DNALLMModel.process_dna_embeddings`,
dna_llm.py:123-146,168):
  * `__call__(text, padding=, truncation=, max_length=, return_tensors=)` -> BatchEncoding{input_ids, attention_mask};
  * **padding is on the LEFT** to the longest row of the batch (`max_length` only truncates; evo2_tokenizer.py:127-146), with
    `pad_token_id = 1`, so `attention_mask.sum()` rows counted from position 0 (dna_llm.py:168) are NOT the valid rows of a
    padded sequence — the reference's quirk, reproduced by the scatter plan (`bra_dna_scatter_plan` takes the first
    `sum(mask)` rows of every sequence whatever the mask's layout; `tests/test_evo2_glue.py`);
  * `pad_token_id == 1`, `eos_token_id == 0`, `vocab_size == 512`, `decode` / `batch_decode` through the inner detokenizer.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from transformers import AutoTokenizer
from transformers.tokenization_utils import PreTrainedTokenizer
from transformers.tokenization_utils_base import BatchEncoding


class CharLevelTokenizer:
    """vortex `CharLevelTokenizer` (evo2's `model.tokenizer`; restated, parity unpinned — see the module docstring)."""

    def __init__(self, vocab_size: int = 512):
        self.name = "CharLevelTokenizer"
        self._vocab_size = int(vocab_size)
        self.eod_id = self.eos_id = 0
        self.pad_id = 1

    @property
    def vocab_size(self) -> int:
        return self._vocab_size

    def clamp(self, n: int) -> int:
        return max(32, min(int(n), self._vocab_size))

    def decode_token(self, token: int) -> str:
        return chr(self.clamp(token))

    def tokenize(self, text: str) -> List[np.uint8]:
        return list(np.frombuffer(text.encode("utf-8"), dtype=np.uint8))

    def tokenize_batch(self, text_batch: Sequence[str]) -> List[List[np.uint8]]:
        return [self.tokenize(t) for t in text_batch]

    def detokenize(self, token_ids) -> str:
        return "".join(self.decode_token(t) for t in token_ids)

    def detokenize_batch(self, token_ids) -> List[str]:
        return [self.detokenize(row) for row in token_ids]


class Evo2Tokenizer(PreTrainedTokenizer):
    """HF-shaped wrapper: tokens are single characters whose id is their code point; batches are LEFT-padded."""

    vocab_files_names: Dict[str, str] = {}
    model_input_names = ["input_ids", "attention_mask"]

    def __init__(self, evo2_tokenizer=None, bos_token="<s>", eos_token="</s>", pad_token="<pad>", unk_token="<unk>", **kwargs):
        self.evo2_tokenizer = evo2_tokenizer if evo2_tokenizer is not None else CharLevelTokenizer(512)
        self._pad_token, self._eos_token, self._bos_token, self._unk_token = pad_token, eos_token, bos_token, unk_token
        super().__init__(bos_token=bos_token, eos_token=eos_token, pad_token=pad_token, unk_token=unk_token, **kwargs)
        # the ids the encoder was trained with win over whatever the base class assigned to the four strings (evo2_tokenizer.py:59-60)
        self.pad_token_id = self.evo2_tokenizer.pad_id
        self.eos_token_id = self.evo2_tokenizer.eos_id

    # ---- vocabulary: the identity between characters and code points ---------------------------------------------------
    @property
    def vocab_size(self) -> int:
        return self.evo2_tokenizer.vocab_size

    def get_vocab(self) -> Dict[str, int]:
        return {chr(i): i for i in range(self.vocab_size)}

    def _tokenize(self, text: str) -> List[str]:
        return [chr(int(b)) for b in self.evo2_tokenizer.tokenize(text)]

    def _convert_token_to_id(self, token: str) -> int:
        return ord(token)

    def _convert_id_to_token(self, index: int) -> str:
        return chr(index)

    def convert_tokens_to_string(self, tokens: List[str]) -> str:
        return "".join(tokens)

    def save_vocabulary(self, save_directory: str, filename_prefix: Optional[str] = None) -> Tuple[str]:
        return ()

    # ---- batches -----------------------------------------------------------------------------------------------------------
    def __call__(self, text: Union[str, List[str]], text_pair=None, padding: Union[bool, str] = False,
                 truncation: Union[bool, str] = False, max_length: Optional[int] = None, return_tensors: Optional[str] = None,
                 return_token_type_ids: Optional[bool] = None, return_attention_mask: Optional[bool] = True, **kwargs) -> BatchEncoding:
        seqs = [text] if isinstance(text, str) else list(text)
        rows = [np.asarray(self.evo2_tokenizer.tokenize(s), dtype=np.int64) for s in seqs]
        if truncation and max_length:
            rows = [r[:max_length] for r in rows]
        lens = [int(r.shape[0]) for r in rows]
        if padding:
            width = max(lens)                                   # ValueError on an empty batch, as in the reference
            ids = np.full((len(rows), width), int(self.pad_token_id), dtype=np.int64)
            mask = np.zeros((len(rows), width), dtype=np.int64)
            for i, (r, n) in enumerate(zip(rows, lens)):        # left padding: the sequence sits at the END of its row
                ids[i, width - n:] = r
                mask[i, width - n:] = 1
            ids_l, mask_l = ids.tolist(), mask.tolist()
        else:
            ids_l, mask_l = [r.tolist() for r in rows], [[1] * n for n in lens]
        data = {"input_ids": ids_l}
        if return_attention_mask:
            data["attention_mask"] = mask_l
        if return_tensors == "pt":
            data = {k: torch.tensor(v) for k, v in data.items()}
        return BatchEncoding(data=data, tensor_type=return_tensors, prepend_batch_axis=False, encoding=None)

    def batch_decode(self, sequences, skip_special_tokens: bool = False, **kwargs) -> List[str]:
        if isinstance(sequences, torch.Tensor):
            sequences = sequences.tolist()
        return self.evo2_tokenizer.detokenize_batch(sequences)

    def decode(self, token_ids, skip_special_tokens: bool = False, **kwargs) -> str:
        if isinstance(token_ids, torch.Tensor):
            token_ids = token_ids.tolist()
        nested = isinstance(token_ids, list) and len(token_ids) > 0 and isinstance(token_ids[0], (list, torch.Tensor))
        if nested:                                             # a batch of one: its first row
            return self.batch_decode(token_ids, skip_special_tokens, **kwargs)[0]
        return self.evo2_tokenizer.detokenize(token_ids)


def register_evo2_tokenizer() -> None:
    """`AutoTokenizer.register("evo2", Evo2Tokenizer)` (evo2_tokenizer.py:204-214; called at import time by reason.py /
    train_dna_qwen.py:38).  Registering twice is an error in `transformers`; the second call is a no-op here."""
    try:
        AutoTokenizer.register("evo2", Evo2Tokenizer)
    except ValueError:
        pass
    print("Evo2Tokenizer registered with AutoTokenizer")
