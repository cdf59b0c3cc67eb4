"""kegg.py:252-327 label construction: the vectorised form against the position-by-position restatement, on rows with
planted markers (adjacent, nested, unterminated, at the row ends, none at all) and left padding."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bioreason_amd.collate import assistant_label_mask      # noqa: E402
from oracle.collate_oracle import assistant_labels_loop     # noqa: E402

A, E, PAD = [7, 8, 9], [5], 0


def _rows(seed, B=6, L=48):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(10, 60, (B, L), generator=g)
    for b in range(B):
        for _ in range(int(torch.randint(0, 4, (1,), generator=g))):
            p = int(torch.randint(0, L - 3, (1,), generator=g))
            ids[b, p:p + 3] = torch.tensor(A)
        for _ in range(int(torch.randint(0, 4, (1,), generator=g))):
            ids[b, int(torch.randint(0, L, (1,), generator=g))] = E[0]
        npad = int(torch.randint(0, 6, (1,), generator=g))
        ids[b, :npad] = PAD                                   # padding_side="left" (kegg.py:246)
    return ids


@pytest.mark.parametrize("seed", range(12))
def test_label_mask_equals_reference_loop(seed):
    ids = _rows(seed)
    assert torch.equal(assistant_label_mask(ids, A, E, PAD), assistant_labels_loop(ids, A, E, PAD))


def test_label_mask_edge_rows():
    rows = [
        [7, 8, 9, 11, 12, 5, 13, 14],          # one closed section
        [11, 7, 8, 9, 12, 13, 14, 15],         # unterminated: runs to the row end
        [11, 12, 13, 14, 15, 7, 8, 9],         # marker at the very end: empty section
        [7, 8, 9, 5, 11, 7, 8, 9],             # end marker right after the start: empty; second start unterminated & empty
        [5, 5, 7, 8, 9, 11, 5, 12],            # end markers before any start are ignored
        [7, 8, 9, 7, 8, 9, 11, 5],             # two starts, one end: both sections close at it
        [0, 0, 7, 8, 9, 11, 12, 5],            # left padding
        [11, 12, 13, 14, 15, 16, 17, 18],      # no marker at all
    ]
    ids = torch.tensor(rows)
    got = assistant_label_mask(ids, A, E, PAD)
    assert torch.equal(got, assistant_labels_loop(ids, A, E, PAD))
    assert got[0].tolist() == [-100, -100, -100, 11, 12, -100, -100, -100]
    assert got[1].tolist() == [-100, -100, -100, -100, 12, 13, 14, 15]
    assert (got[2] == -100).all() and (got[7] == -100).all()
    # multi-token end marker and a marker longer than the row
    ids2 = torch.tensor([[7, 8, 9, 11, 5, 6, 12, 13]])
    assert torch.equal(assistant_label_mask(ids2, A, [5, 6], PAD), assistant_labels_loop(ids2, A, [5, 6], PAD))
    assert (assistant_label_mask(ids2[:, :2], A, E, PAD) == -100).all()


REF_KEGG = "/root/reference/bioreason/dataset/kegg.py"


def _reference_collate():
    """`qwen_dna_collate_fn` ast-extracted from the reference (kegg.py imports trl, which is absent): its label loop,
    kegg.py:252-327, runs unmodified on rows handed over by a stub processor"""
    import ast
    from typing import Dict, List
    tree = ast.parse(open(REF_KEGG).read())
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "qwen_dna_collate_fn")
    ns = {"torch": torch, "List": List, "Dict": Dict, "DLProcessor": object,
          "maybe_apply_chat_template": lambda ex, proc: {"prompt": ex["prompt"]}}
    exec(compile(ast.Module(body=[node], type_ignores=[]), REF_KEGG, "exec"), ns)
    return ns["qwen_dna_collate_fn"]


class _StubTokenizer:
    pad_token_id = PAD

    def encode(self, text, add_special_tokens=False):
        return {"<|im_start|>assistant\n": list(A), "<|im_end|>": list(E)}[text]


class _StubProcessor:
    def __init__(self, ids):
        self.ids, self.tokenizer = ids, _StubTokenizer()

    def __call__(self, **kw):
        return {"input_ids": self.ids.clone()}


@pytest.mark.skipif(not os.path.exists(REF_KEGG), reason="reference checkout not present")
@pytest.mark.parametrize("seed", range(8))
def test_label_mask_equals_reference_collate_function(seed):
    """pins collate.assistant_label_mask (and the loop restatement the other tests use) to kegg.py:252-327 itself"""
    ref_fn = _reference_collate()
    ids = _rows(100 + seed, B=5, L=40)
    examples = [{"prompt": "x", "dna_sequences": [], "answer": " a "} for _ in range(ids.shape[0])]
    out = ref_fn(examples, _StubProcessor(ids), 64, 16, return_answer_in_batch=True)
    assert torch.equal(out["labels"], assistant_label_mask(ids, A, E, PAD))
    assert torch.equal(out["labels"], assistant_labels_loop(ids, A, E, PAD))
    assert out["answer"] == ["a"] * ids.shape[0]


def test_qwen_dna_collate_fn_end_to_end(tmp_path):
    """chat template -> DLProcessor -> labels: only the assistant's text carries labels, DNA placeholders are expanded"""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import EsmTokenizer, GPT2TokenizerFast
    from bioreason.dataset.kegg import qwen_dna_collate_fn
    from bioreason_amd.chat_template import CHAT_TEMPLATE
    from bioreason_amd.processing import DLProcessor
    words = ["<|endoftext|>", "[UNK]", "<|im_start|>", "<|im_end|>", "user", "assistant", "Q", "?", "Answer:", "ALS", "<think>", "</think>", "why"]
    tk = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="[UNK]"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok = GPT2TokenizerFast(tokenizer_object=tk, eos_token="<|endoftext|>", unk_token="[UNK]")
    tok.add_special_tokens({"additional_special_tokens": ["<|dna_start|>", "<|dna_pad|>", "<|dna_end|>", "<|im_start|>", "<|im_end|>"]})
    tok.chat_template = CHAT_TEMPLATE
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(["<cls>", "<pad>", "<eos>", "<unk>", "A", "C", "G", "T", "N", "<mask>"]))
    proc = DLProcessor(tokenizer=tok, dna_tokenizer=EsmTokenizer(str(vf)))
    ex = [{"prompt": [{"role": "user", "content": [{"type": "dna"}, {"type": "text", "text": " Q ? "}]},
                      {"role": "assistant", "reasoning_content": " why ", "content": [{"type": "text", "text": " Answer: ALS "}]}],
           "dna_sequences": ["ACGT"], "answer": "ALS "},
          {"prompt": [{"role": "user", "content": [{"type": "text", "text": " Q "}]},
                      {"role": "assistant", "content": [{"type": "text", "text": " ALS "}]}],
           "dna_sequences": [], "answer": "ALS"}]
    out = qwen_dna_collate_fn(ex, proc, 64, 8, return_answer_in_batch=True)
    ids, labels = out["input_ids"], out["labels"]
    assert out["answer"] == ["ALS", "ALS"] and out["batch_idx_map"] == [0]
    n_real = int((out["dna_tokenized"]["input_ids"][0] != 1).sum())
    assert (ids[0] == tok.convert_tokens_to_ids("<|dna_pad|>")).sum().item() == n_real
    lab_tokens = [tok.convert_ids_to_tokens(int(t)) for t in labels[0][labels[0] != -100]]
    assert "ALS" in lab_tokens and "Answer:" in lab_tokens and "Q" not in lab_tokens and "<|im_end|>" not in lab_tokens
    assert (labels[ids == tok.pad_token_id] == -100).all()


def test_collate_renders_like_trl_maybe_apply_chat_template(tmp_path):
    """trl's apply_chat_template on a {"prompt": messages} example: last role assistant -> continue_final_message (the text
    ends at the assistant content, no closing <|im_end|>), last role user -> add_generation_prompt"""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import EsmTokenizer, GPT2TokenizerFast
    from bioreason_amd.chat_template import CHAT_TEMPLATE
    from bioreason_amd.collate import qwen_dna_collate_fn
    from bioreason_amd.processing import DLProcessor
    words = ["<|endoftext|>", "[UNK]", "<|im_start|>", "<|im_end|>", "user", "assistant", "Q", "ALS", "<think>", "</think>"]
    tk = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="[UNK]"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok = GPT2TokenizerFast(tokenizer_object=tk, eos_token="<|endoftext|>", unk_token="[UNK]")
    tok.add_special_tokens({"additional_special_tokens": ["<|dna_start|>", "<|dna_pad|>", "<|dna_end|>", "<|im_start|>", "<|im_end|>"]})
    tok.chat_template = CHAT_TEMPLATE
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(["<cls>", "<pad>", "<eos>", "<unk>", "A", "C", "G", "T", "N", "<mask>"]))
    proc = DLProcessor(tokenizer=tok, dna_tokenizer=EsmTokenizer(str(vf)))
    sft = [{"role": "user", "content": [{"type": "text", "text": " Q "}]}, {"role": "assistant", "content": [{"type": "text", "text": " ALS "}]}]
    usr = sft[:1]
    out = qwen_dna_collate_fn([{"prompt": sft, "dna_sequences": []}, {"prompt": usr, "dna_sequences": []}], proc, 64, 8)
    im_end, im_start, als = (tok.convert_tokens_to_ids(t) for t in ("<|im_end|>", "<|im_start|>", "ALS"))
    row0 = out["input_ids"][0][out["attention_mask"][0].bool()].tolist()
    row1 = out["input_ids"][1][out["attention_mask"][1].bool()].tolist()
    assert row0[-1] == als and row0.count(im_end) == 1                 # ends AT the assistant content: only the user turn is closed
    want0 = tok.apply_chat_template(sft, tokenize=False, continue_final_message=True)
    want1 = tok.apply_chat_template(usr, tokenize=False, add_generation_prompt=True)
    assert row0 == tok(want0, add_special_tokens=False)["input_ids"] and row1 == tok(want1, add_special_tokens=False)["input_ids"]
    assert row1.count(im_start) == 2 and tok.convert_tokens_to_ids("assistant") in row1[-3:]      # generation prompt appended
    lab = out["labels"][0]
    assert lab[lab != -100].tolist()[-1] == als
