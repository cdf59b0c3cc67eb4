"""`DLProcessor` against the reference class itself (imported from /root/reference when present; two
shims for transformers 5 — the removed `CommonKwargs` name and `_merge_kwargs`, whose 4.x behaviour of routing tokenizer
kwargs to `text_kwargs` is restored) and against committed known answers otherwise.  Toy tokenizers built offline."""
import os
import sys
import typing

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bioreason_amd.processing import DLProcessor      # noqa: E402

TEXTS = ["hello <|dna_start|> <|dna_pad|> <|dna_end|> what <|dna_start|> <|dna_pad|> <|dna_end|>",
         "world <|dna_start|> <|dna_pad|> <|dna_end|> ?",
         "what is this ?"]
DNA = [["ACGT", "AC"], ["GGGTTTAA"], []]


@pytest.fixture(scope="module")
def toks(tmp_path_factory):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import EsmTokenizer, GPT2TokenizerFast
    vocab = {"<|endoftext|>": 0, "<|dna_pad|>": 1, "<|dna_start|>": 2, "<|dna_end|>": 3, "hello": 4, "world": 5, "what": 6,
             "is": 7, "this": 8, "?": 9, "[UNK]": 10}
    tk = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    text_tok = GPT2TokenizerFast(tokenizer_object=tk, eos_token="<|endoftext|>", unk_token="[UNK]")
    text_tok.add_special_tokens({"additional_special_tokens": ["<|dna_start|>", "<|dna_pad|>", "<|dna_end|>"]})
    vf = tmp_path_factory.mktemp("esm") / "vocab.txt"
    vf.write_text("\n".join(["<cls>", "<pad>", "<eos>", "<unk>", "A", "C", "G", "T", "N", "<mask>"]))
    return text_tok, EsmTokenizer(str(vf))


def _call(p):
    return p(batch_dna_sequences=[list(s) for s in DNA], text=list(TEXTS), max_length_text=64, max_length_dna=6,
             return_tensors="pt", padding_side="left", add_special_tokens=False)


def test_known_answers(toks):
    text_tok, dna_tok = toks
    out = _call(DLProcessor(tokenizer=text_tok, dna_tokenizer=dna_tok))
    assert out["batch_idx_map"] == [0, 0, 1]
    d = out["dna_tokenized"]["input_ids"]
    assert d.shape == (3, 6)                                   # truncated to max_length_dna = 6, right-padded with id 1
    n_real = (d != 1).sum(1).tolist()
    ids, mask = out["input_ids"], out["attention_mask"]
    pad_id = text_tok.convert_tokens_to_ids("<|dna_pad|>")
    per_row = (ids == pad_id).sum(1).tolist()
    assert per_row == [n_real[0] + n_real[1], n_real[2], 0]    # one placeholder per real DNA token, in batch order
    assert mask[:, -1].all() and (mask[2, 0] == 0)            # left padding
    assert ids.shape[1] == max(int(m.sum()) for m in mask)
    # no DNA at all: the reference returns dna_tokenized None and an empty map (processing_dl.py:117-118)
    none = DLProcessor(tokenizer=text_tok, dna_tokenizer=dna_tok)(batch_dna_sequences=[[], []], text=["hello", "world ?"],
                                                                 return_tensors="pt")
    assert none["dna_tokenized"] is None and none["batch_idx_map"] == []
    # a bare string prompt is wrapped in a list (processing_dl.py:170-171)
    one = DLProcessor(tokenizer=text_tok, dna_tokenizer=dna_tok)(text="hello world", return_tensors="pt")
    assert one["input_ids"].shape[0] == 1 and "dna_tokenized" not in one


@pytest.mark.skipif(not os.path.isdir("/root/reference/bioreason"), reason="reference sources not on this machine")
def test_equals_reference_class(toks):
    text_tok, dna_tok = toks
    from oracle.make_golden import import_from_reference      # isolated: leaves sys.modules / sys.path as found, checks the file's origin
    RefProcessor = import_from_reference("bioreason.models.dl.processing_dl", "DLProcessor")
    assert RefProcessor is not DLProcessor
    ref = RefProcessor(tokenizer=text_tok, dna_tokenizer=dna_tok)
    ref._merge_kwargs = lambda cls, tokenizer_init_kwargs=None, **kw: {"text_kwargs": dict(kw)}      # transformers 4.x routing
    want, got = _call(ref), _call(DLProcessor(tokenizer=text_tok, dna_tokenizer=dna_tok))
    assert set(want.keys()) == set(got.keys())
    assert torch.equal(want["input_ids"], got["input_ids"]) and torch.equal(want["attention_mask"], got["attention_mask"])
    assert want["batch_idx_map"] == got["batch_idx_map"]
    for k in ("input_ids", "attention_mask"):
        assert torch.equal(want["dna_tokenized"][k], got["dna_tokenized"][k])
    assert ref.model_input_names == DLProcessor(tokenizer=text_tok, dna_tokenizer=dna_tok).model_input_names


def test_dna_module_answers_and_input_preparation(toks):
    """nucleotide_module.py:16-262: the adapter's answers, and prepare_model_inputs == a direct processor call"""
    from bioreason_amd.dna_modules import DNABaseModule, NucleotideDNAModule
    from bioreason_amd.dna_llm import DNALLMModel
    text_tok, dna_tok = toks
    mod = NucleotideDNAModule()
    assert isinstance(mod, DNABaseModule) and mod.get_dnallm_key() == "qwen" and mod.is_embeds_input()
    assert mod.get_model_class("DNALLM-qwen", {}) is DNALLMModel and mod.get_processing_class() is DLProcessor
    with pytest.raises(ValueError):
        mod.get_model_class("other", {})
    assert mod.get_custom_multimodal_keywords() == ["dna_tokenized", "batch_idx_map"] and mod.get_dnallm_modules_keywords() == ["dna"]
    assert mod.get_custom_processing_keywords() == [("dna_tokenizer", "max_length")] and mod.get_non_generate_params() == []
    proc = DLProcessor(tokenizer=text_tok, dna_tokenizer=dna_tok)

    class M:
        max_length_text, max_length_dna = 64, 6
    got = mod.prepare_model_inputs(proc, M(), list(TEXTS), [list(s) for s in DNA])
    want = _call(proc)
    assert torch.equal(got["input_ids"], want["input_ids"]) and got["batch_idx_map"] == want["batch_idx_map"]
    assert mod.prepare_prompt(proc, [{"prompt": "hello world"}]) == ["hello world"]
    ok = "<think>x</think> <answer>{\"box\": [1, 2, 3, 4]}</answer>"
    assert NucleotideDNAModule.format_reward_rec([[{"content": ok}], [{"content": "<think>x</think>"}]]) == [1.0, 0.0]
    assert NucleotideDNAModule.select_reward_func("format", "rec") is NucleotideDNAModule.format_reward_rec
