"""CPU ORACLE support (test infrastructure): golden records of the reference's data-format functions and Evo2 tokenizer wrapper.

Runs the reference's own function objects (oracle/ref_dataformats.py) on seeded records and writes what they return to
tests/golden/dataformats.json and tests/golden/evo2_tokenizer.json, so that the same comparisons run where /root/reference does
not exist.      python oracle/make_dataformat_golden.py        (build container only)
"""
import copy
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_dataformats as RD      # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def records(seed: int = 0, n: int = 12):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        L = [0, 7, 8, 9, 24, 25, 32, 33, 64, 2056, 2057, 2100][i % 12]
        seq = lambda m: "".join(rng.choice("ACGTN") for _ in range(m))       # noqa: E731
        ans = rng.choice([" Pathogenic ; extra", "benign", "['missense_variant', 'stop_gained']", "  Long QT_syndrome  ", "x;y;z"])
        out.append({"question": rng.choice(["  What is the effect?  ", "Q\n", "", "Which pathway?"]) + str(i),
                    "answer": ans, "reasoning": rng.choice(["  step a\nstep b ", "", "because"]),
                    "reference_sequence": seq(L), "variant_sequence": seq(max(0, L - rng.randrange(0, 3)))})
    return out


def main():
    ut = RD.namespace("utils", ["truncate_dna"])
    kg = RD.namespace("kegg", ["get_format_kegg_function", "format_kegg_for_dna_llm", "format_kegg_for_llm"])
    ve = RD.namespace("variant_effect", ["get_format_variant_effect_function", "clean_variant_effect_example",
                                         "clean_variant_effect_non_snv_example", "format_variant_effect_for_dna_llm",
                                         "format_variant_effect_for_llm"])
    recs = records()
    gold = {"records": recs, "truncate": {}, "kegg": {}, "vep": {}, "clean": [], "clean_non_snv": []}
    for per_side in (0, 4, 12, 1024):
        t = [ut["truncate_dna"](copy.deepcopy(r), per_side) for r in recs]
        gold["truncate"][str(per_side)] = [[r["reference_sequence"], r["variant_sequence"]] for r in t]
    for name in ("llm", "dna-llm", "DNA-LLM"):
        gold["kegg"][name] = [kg["get_format_kegg_function"](name)(copy.deepcopy(r)) for r in recs[:9]]     # (the short records)
        gold["vep"][name] = [ve["get_format_variant_effect_function"](name)(copy.deepcopy(r)) for r in recs[:9]]
    gold["clean"] = [ve["clean_variant_effect_example"](copy.deepcopy(r))["answer"] for r in recs]
    gold["clean_non_snv"] = [ve["clean_variant_effect_non_snv_example"](copy.deepcopy(r))["answer"] for r in recs]
    json.dump(gold, open(os.path.join(GOLD, "dataformats.json"), "w"), indent=0)

    # ---- tokenizer wrapper: the reference class around THIS repository's restated CharLevelTokenizer
    from bioreason_amd.evo2_tokenizer import CharLevelTokenizer
    tok = RD.evo2_tokenizer_module().Evo2Tokenizer(CharLevelTokenizer(512))
    rng = random.Random(1)
    batches = [["ACGT", "AC"], ["A"], ["", "ACG"], ["ACGTNNNNACGT" * 3, "TTT", "G" * 40], "ACGTN",
               ["".join(rng.choice("ACGTNacgt") for _ in range(rng.randrange(1, 90))) for _ in range(6)]]
    calls = []
    for b in batches:
        for kw in ({}, {"padding": True}, {"padding": True, "truncation": True, "max_length": 16},
                   {"truncation": True, "max_length": 3}, {"padding": "longest", "max_length": 64}, {"padding": True, "return_attention_mask": False}):
            enc = tok(b, **kw)
            calls.append({"text": b, "kw": kw, "out": {k: v for k, v in enc.items()}})
    dec = [[65, 67, 71, 84], [1, 1, 65], [0, 300, 511, 512, 600, 10], []]
    gold_t = {"calls": calls, "decode_in": dec, "decode": [tok.decode(d) for d in dec], "batch_decode": tok.batch_decode(dec),
              "decode_nested": tok.decode([[71, 71], [65]]),
              "ids": {"pad": tok.pad_token_id, "eos": tok.eos_token_id, "vocab_size": tok.vocab_size, "len_vocab": len(tok.get_vocab())},
              "tokenize": tok.tokenize("ACGTn"), "convert": tok.convert_tokens_to_ids(["A", "C", "n"]),
              "to_string": tok.convert_tokens_to_string(["A", "C"]), "model_input_names": list(tok.model_input_names)}
    json.dump(gold_t, open(os.path.join(GOLD, "evo2_tokenizer.json"), "w"), indent=0)
    print("written")


if __name__ == "__main__":
    main()
