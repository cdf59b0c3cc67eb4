"""SURVEY §8f N3, first slice: `Evo2Tokenizer` (bioreason/models/evo2_tokenizer.py:16-218) — equal to the reference's own class fed
the same inner tokenizer, call by call, and to what that class returned in the build container (tests/golden/evo2_tokenizer.json)."""
import json
import os
import random
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bioreason_amd.evo2_tokenizer import CharLevelTokenizer, Evo2Tokenizer, register_evo2_tokenizer      # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "evo2_tokenizer.json")))


def _plain(enc):
    return {k: (v.tolist() if isinstance(v, torch.Tensor) else v) for k, v in enc.items()}


def test_char_level_tokenizer_published_constants():
    t = CharLevelTokenizer(512)
    assert (t.vocab_size, t.pad_id, t.eos_id, t.eod_id) == (512, 1, 0, 0)
    assert [int(x) for x in t.tokenize("ACGTN")] == [65, 67, 71, 84, 78]
    assert t.detokenize([65, 1, 600]) == "A" + chr(32) + chr(512) and t.detokenize_batch([[71], []]) == ["G", ""]
    assert [int(x) for x in t.tokenize("é")] == [0xC3, 0xA9]                  # bytes, not code points


def test_wrapper_equals_golden_calls():
    tok = Evo2Tokenizer(CharLevelTokenizer(512))
    for c in GOLD["calls"]:
        assert _plain(tok(c["text"], **c["kw"])) == c["out"], (c["text"], c["kw"])
    assert [tok.decode(d) for d in GOLD["decode_in"]] == GOLD["decode"]
    assert tok.batch_decode(GOLD["decode_in"]) == GOLD["batch_decode"]
    assert tok.batch_decode(torch.tensor([[65, 67]])) == ["AC"] and tok.decode(torch.tensor([71, 84])) == "GT"
    assert tok.decode([[71, 71], [65]]) == GOLD["decode_nested"]
    assert {"pad": tok.pad_token_id, "eos": tok.eos_token_id, "vocab_size": tok.vocab_size, "len_vocab": len(tok.get_vocab())} == GOLD["ids"]
    assert tok.tokenize("ACGTn") == GOLD["tokenize"] and tok.convert_tokens_to_ids(["A", "C", "n"]) == GOLD["convert"]
    assert tok.convert_tokens_to_string(["A", "C"]) == GOLD["to_string"] and list(tok.model_input_names) == GOLD["model_input_names"]
    assert tok.save_vocabulary("/tmp") == ()


def test_left_padding_and_tensor_types():
    tok = Evo2Tokenizer()
    enc = tok(["ACGT", "AC"], padding=True, truncation=True, max_length=2048, return_tensors="pt")
    assert enc["input_ids"].dtype == torch.int64 and enc["input_ids"].tolist() == [[65, 67, 71, 84], [1, 1, 65, 67]]
    assert enc["attention_mask"].tolist() == [[1, 1, 1, 1], [0, 0, 1, 1]]
    assert enc.input_ids is enc["input_ids"]                                    # BatchEncoding attribute access (processing_dl.py)
    with pytest.raises(ValueError):
        tok([], padding=True)                                                   # max() of an empty batch, like the reference


def test_register_is_idempotent(capsys):
    register_evo2_tokenizer()
    register_evo2_tokenizer()
    assert "registered" in capsys.readouterr().out


@pytest.mark.skipif(not os.path.exists("/root/reference/bioreason/models/evo2_tokenizer.py"), reason="reference checkout not present")
def test_equal_the_reference_class_on_random_batches():
    from oracle import ref_dataformats as RD
    ref = RD.evo2_tokenizer_module().Evo2Tokenizer(CharLevelTokenizer(512))
    mine = Evo2Tokenizer(CharLevelTokenizer(512))
    rng = random.Random(5)
    for trial in range(60):
        n = rng.randrange(1, 6)
        batch = ["".join(rng.choice("ACGTNacgtn-") for _ in range(rng.randrange(0, 70))) for _ in range(n)]
        text = batch[0] if trial % 7 == 0 else batch
        kw = {"padding": rng.choice([False, True, "longest"]), "truncation": rng.choice([False, True]),
              "max_length": rng.choice([None, 1, 5, 32]), "return_tensors": rng.choice([None, "pt"]) }
        if kw["return_tensors"] == "pt" and not kw["padding"] and not isinstance(text, str) and len({len(b) if not (kw["truncation"] and kw["max_length"]) else min(len(b), kw["max_length"]) for b in batch}) > 1:
            kw["return_tensors"] = None                                         # ragged rows cannot become one tensor in either class
        assert _plain(mine(text, **kw)) == _plain(ref(text, **kw)), (text, kw)
    ids = [[rng.randrange(0, 700) for _ in range(rng.randrange(0, 12))] for _ in range(8)]
    assert mine.batch_decode(ids) == ref.batch_decode(ids) and [mine.decode(i) for i in ids] == [ref.decode(i) for i in ids]
    assert (mine.pad_token_id, mine.eos_token_id, mine.vocab_size, mine.padding_side) == (ref.pad_token_id, ref.eos_token_id, ref.vocab_size, ref.padding_side)
    assert mine.get_vocab() == ref.get_vocab() and mine.tokenize("ACGu") == ref.tokenize("ACGu")
