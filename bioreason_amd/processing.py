"""`DLProcessor` — the reference's joint text + DNA processor (bioreason/models/dl/processing_dl.py:36-300), row a1 of
SURVEY §8: flattens the per-sample DNA lists into one padded DNA batch with `batch_idx_map`, expands every
`<|dna_pad|>` in the prompts to one placeholder per real DNA token of its sequence, and tokenises the prompts.

Host-side string / integer work (it stays Python in the reference too); kept here so that the outputs feeding
`DNALLMModel.forward` — `input_ids`, `attention_mask`, `dna_tokenized`, `batch_idx_map` — are produced the reference's
way: `tests/test_processing.py` runs the reference class itself on the same inputs.  Differences in mechanism only: the
placeholder expansion is one split/join per prompt instead of a `str.replace` per occurrence (same text), and the class
does not derive from `ProcessorMixin` (whose kwargs merging changed in transformers 5)."""
from typing import Any, Dict, List, Optional, Union

DNA_TOKEN = "<|dna_pad|>"
DNA_PAD_ID = 1          # processing_dl.py:188 counts `input_ids != 1`: the NT-v2 / ESM pad id


class DLProcessor:
    attributes = ["tokenizer", "dna_tokenizer"]

    def __init__(self, tokenizer=None, dna_tokenizer=None, chat_template=None, **kwargs):
        self.tokenizer = tokenizer
        self.dna_tokenizer = dna_tokenizer
        self.dna_token = getattr(tokenizer, "dna_token", DNA_TOKEN)                       # processing_dl.py:71-75
        if chat_template is None and hasattr(tokenizer, "chat_template"):
            chat_template = tokenizer.chat_template
        self.chat_template = chat_template
        if tokenizer is not None and getattr(tokenizer, "pad_token", None) is None:      # processing_dl.py:83-84
            tokenizer.pad_token = tokenizer.eos_token

    # ---------------------------------------------------------------------------------------------------------------
    def tokenize_dna_sequences(self, batch_dna_sequences: List[List[str]], max_length: int = 2048, return_tensors: str = "pt",
                               device: str = "cuda") -> Dict[str, Any]:
        """processing_dl.py:87-132: all sequences of the batch in sample order, `batch_idx_map[s]` = their sample"""
        all_sequences, batch_idx_map = [], []
        for b, seqs in enumerate(batch_dna_sequences):
            for s in seqs:
                all_sequences.append(s)
                batch_idx_map.append(b)
        if not all_sequences:
            return {"dna_tokenized": None, "batch_idx_map": []}
        tok = self.dna_tokenizer(all_sequences, padding=True, truncation=True, max_length=max_length,
                                 return_tensors=return_tensors, return_attention_mask=True)
        return {"dna_tokenized": tok, "batch_idx_map": batch_idx_map}

    def _expand_placeholders(self, text: List[str], dna_ids) -> List[str]:
        """processing_dl.py:183-193: the i-th `<|dna_pad|>` of the batch (prompts in order, left to right) becomes as many
        copies as its sequence has non-pad tokens"""
        out, index = [], 0
        for t in text:
            parts = t.split(self.dna_token)
            pieces = [parts[0]]
            for tail in parts[1:]:
                n = int((dna_ids[index] != DNA_PAD_ID).sum().item())
                pieces.append(self.dna_token * n)
                pieces.append(tail)
                index += 1
            out.append("".join(pieces))
        return out

    def __call__(self, batch_dna_sequences: Optional[List[List[str]]] = None, text: Optional[Union[str, List[str]]] = None,
                 max_length_text: int = 512, max_length_dna: int = 2048, return_tensors: str = "pt", device: str = "cuda",
                 **kwargs):
        from transformers.feature_extraction_utils import BatchFeature
        if not isinstance(text, list):
            text = [text]
        text = list(text)
        dna_inputs = {}
        if batch_dna_sequences is not None:
            res = self.tokenize_dna_sequences(batch_dna_sequences, max_length=max_length_dna, return_tensors=return_tensors,
                                              device=device)
            if res["dna_tokenized"] is not None:
                text = self._expand_placeholders(text, res["dna_tokenized"]["input_ids"])
            dna_inputs = {"dna_tokenized": res["dna_tokenized"], "batch_idx_map": res["batch_idx_map"]}
        text_kwargs = {k: v for k, v in kwargs.items() if k != "padding"}                  # processing_dl.py:205-209
        text_inputs = self.tokenizer(text, max_length=max_length_text + 2 * max_length_dna, return_tensors=return_tensors,
                                     padding=True, truncation=True, **text_kwargs)
        return BatchFeature(data={**text_inputs, **dna_inputs})

    # ---------------------------------------------------------------------------------------------------------------
    def batch_decode(self, *args, **kwargs) -> List[str]:
        return self.tokenizer.batch_decode(*args, **kwargs)

    def decode(self, *args, **kwargs) -> str:
        return self.tokenizer.decode(*args, **kwargs)

    def post_process_dna_to_text(self, generated_outputs, skip_special_tokens: bool = True, **kwargs) -> List[str]:
        return self.tokenizer.batch_decode(generated_outputs, skip_special_tokens=skip_special_tokens, **kwargs)

    @property
    def model_input_names(self) -> List[str]:
        return list(dict.fromkeys(list(self.tokenizer.model_input_names) + ["dna_tokenized", "batch_idx_map"]))
