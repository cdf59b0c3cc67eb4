"""Input-side record formats (`bioreason_amd/datasets.py`; the `bioreason.dataset.*` names of train_dna_qwen.py:27-33): equal to the
reference's own function objects on seeded records (ast-extracted — the files import the absent `trl`), and equal to the records
those functions returned in the build container (tests/golden/dataformats.json) where /root/reference does not exist."""
import copy
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bioreason_amd import datasets as D      # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "dataformats.json")))
RECS = GOLD["records"]
HAVE_REF = os.path.exists("/root/reference/bioreason/dataset/kegg.py")


def test_truncate_dna_equals_golden():
    for per_side, rows in GOLD["truncate"].items():
        for r, (ref_seq, var_seq) in zip(RECS, rows):
            out = D.truncate_dna(copy.deepcopy(r), int(per_side))
            assert (out["reference_sequence"], out["variant_sequence"]) == (ref_seq, var_seq)
    r = {"reference_sequence": "A" * 2057, "variant_sequence": "C" * 2056}
    out = D.truncate_dna(r)                                    # default 1024 per side; 2056 is NOT longer than 2 * 1024 + 8
    assert out is r and len(r["reference_sequence"]) == 9 and len(r["variant_sequence"]) == 2056


@pytest.mark.parametrize("task,getter", [("kegg", D.get_format_kegg_function), ("vep", D.get_format_variant_effect_function)])
def test_formatters_equal_golden(task, getter):
    for name, rows in GOLD[task].items():
        fn = getter(name)
        got = [fn(copy.deepcopy(r)) for r in RECS[:len(rows)]]
        assert got == rows
    with pytest.raises(ValueError):
        getter("dna_llm")
    rec = getter("dna-llm")(copy.deepcopy(RECS[3]))
    assert [c["type"] for c in rec["prompt"][0]["content"]] == ["dna", "dna", "text"] and rec["prompt"][1]["role"] == "assistant"
    assert getter("llm")(copy.deepcopy(RECS[3]))["dna_sequences"] == ["", ""]


def test_clean_functions_equal_golden():
    assert [D.clean_variant_effect_example(copy.deepcopy(r))["answer"] for r in RECS] == GOLD["clean"]
    assert [D.clean_variant_effect_non_snv_example(copy.deepcopy(r))["answer"] for r in RECS] == GOLD["clean_non_snv"]
    assert D.clean_variant_effect_non_snv_example({"answer": "['stop_gained', 'x_y']"})["answer"] == "stop gained, x y"


def _write_kegg_dir(tmp_path, n=11):
    for i in range(n):
        item = {"question": f"q{i}", "answer": f"  Disease {i} ", "reasoning": {"reasoning_steps": [f"s{i}a", f"s{i}b"]},
                "reference_sequence": " acgt ", "variant_sequence": "acgtn"}
        if i == 3:
            item = {"question": "only"}                        # missing fields take the reference's defaults
        with open(tmp_path / f"KEGG_{100 + i}_x.json", "w") as fh:
            json.dump(item, fh)
    (tmp_path / "notes.txt").write_text("ignored")
    return str(tmp_path)


def test_kegg_dataset_split_and_hf_conversion(tmp_path):
    d = _write_kegg_dir(tmp_path)
    ds = D.KEGGDataset(d)
    assert len(ds) == 11
    assert ds[0] == {"question": "q0", "answer": "disease 0", "reasoning": "s0a\ns0b", "reference_sequence": "ACGT", "variant_sequence": "ACGTN"}
    assert ds[3] == {"question": "only", "answer": "", "reasoning": "", "reference_sequence": "", "variant_sequence": ""}
    tr, va, te = D.split_kegg_dataset(ds, seed=7)
    assert (len(tr), len(va), len(te)) == (8, 1, 2)
    tr2, _, _ = D.split_kegg_dataset(ds, seed=7)
    assert list(tr.indices) == list(tr2.indices)
    hf = D.torch_to_hf_dataset(tr)
    assert hf.num_rows == 8 and set(hf.column_names) == set(ds[0]) and hf[0] == tr[0]
    assert D.torch_to_hf_dataset([]).num_rows == 0
    assert D.torch_to_hf_dataset([1, 2, 3])["data"] == [1, 2, 3]
    dl = D.create_kegg_dataloader(d, batch_size=4, shuffle=False, num_workers=0, pin_memory=False)
    assert next(iter(dl))["question"] == ["q0", "q1", "q2", "only"]


def test_dna_collate_fn_shapes():
    from bioreason_amd.evo2_tokenizer import Evo2Tokenizer
    tok = Evo2Tokenizer()
    batch = [{"reference_sequence": "ACGT", "variant_sequence": "AC", "answer": "a"}, {"reference_sequence": "A", "variant_sequence": "ACGTN", "answer": "b"}]
    out = D.dna_collate_fn(batch, tok, {"a": 0, "b": 1}, max_length=4)
    assert out["ref_ids"].shape == (2, 4) and out["alt_ids"].shape == (2, 4) and out["labels"].tolist() == [0, 1]


@pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present")
def test_equal_the_reference_function_objects(tmp_path):
    from oracle import ref_dataformats as RD
    ut = RD.namespace("utils", ["truncate_dna", "torch_to_hf_dataset"])
    kg = RD.namespace("kegg", ["KEGGDataset", "split_kegg_dataset", "get_format_kegg_function", "format_kegg_for_dna_llm", "format_kegg_for_llm"])
    ve = RD.namespace("variant_effect", ["get_format_variant_effect_function", "clean_variant_effect_example",
                                         "clean_variant_effect_non_snv_example", "format_variant_effect_for_dna_llm",
                                         "format_variant_effect_for_llm"])
    for r in RECS:
        for n in (0, 1, 4, 12, 1024):
            assert D.truncate_dna(copy.deepcopy(r), n) == ut["truncate_dna"](copy.deepcopy(r), n)
        for name in ("llm", "dna-llm", "Dna-LLM"):
            assert D.get_format_kegg_function(name)(copy.deepcopy(r)) == kg["get_format_kegg_function"](name)(copy.deepcopy(r))
            assert D.get_format_variant_effect_function(name)(copy.deepcopy(r)) == ve["get_format_variant_effect_function"](name)(copy.deepcopy(r))
        assert D.clean_variant_effect_example(copy.deepcopy(r)) == ve["clean_variant_effect_example"](copy.deepcopy(r))
        assert D.clean_variant_effect_non_snv_example(copy.deepcopy(r)) == ve["clean_variant_effect_non_snv_example"](copy.deepcopy(r))
    d = _write_kegg_dir(tmp_path)
    mine, ref = D.KEGGDataset(d), kg["KEGGDataset"](d)
    assert [mine[i] for i in range(len(mine))] == [ref[i] for i in range(len(ref))]
    sm, sr = D.split_kegg_dataset(mine, seed=3), kg["split_kegg_dataset"](ref, seed=3)
    assert [list(a.indices) for a in sm] == [list(b.indices) for b in sr]
    assert D.torch_to_hf_dataset(sm[0]).to_dict() == ut["torch_to_hf_dataset"](sr[0]).to_dict()
