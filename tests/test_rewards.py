"""the reward functions against known answers and, when /root/reference is present, against the
reference's own function bodies (extracted from reason.py with `ast` — the module itself imports trl / peft, which are
absent, so it cannot be imported as a whole)."""
import ast
import os
import random
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bioreason_amd import rewards as R      # noqa: E402

NAMES = ["extract_xml_answer", "extract_hash_answer", "correctness_reward_func", "less_than_4_reward_func",
         "strict_format_reward_func", "soft_format_reward_func", "count_xml", "xmlcount_reward_func"]


def _msg(t):
    return [{"role": "assistant", "content": t}]


def test_known_answers():
    good = "<think>\nbecause\n</think>\nBRCA1 variant\n"
    assert R.strict_format_reward_func([_msg(good)]) == [0.5]
    assert R.strict_format_reward_func([_msg("<think>x</think> y")]) == [0.0]
    assert R.soft_format_reward_func([_msg("<think>x</think> y"), _msg("no tags")]) == [0.5, 0.0]
    assert R.xmlcount_reward_func([_msg(good), _msg("<think>\n<think>\n</think>"), _msg("")]) == [0.25, 0.0, 0.0]
    assert R.less_than_4_reward_func([_msg(good), _msg("<think>\n</think>\none two three four five")]) == [0.5, 0.0]
    assert R.extract_xml_answer("a</think> b </think>  c ") == "c" and R.extract_hash_answer("x #### 42 ") == "42"
    assert R.extract_hash_answer("no marker") is None
    # the reference zips the responses with the ELEMENTS of answer[0] (reason.py:200)
    comps = [_msg("<think>\n</think>\nthe Answer is cancer"), _msg("nothing")]
    assert R.correctness_reward_func(prompts=[[{"content": "q"}]], completions=comps, answer=[["cancer", "cancer"]]) == [2.0, 0.0]
    assert R.correctness_reward_func(prompts=[[{"content": "q"}]], completions=comps, answer=["cn"]) == [2.0, 2.0]
    assert set(R.reward_funcs_registry) == {"xmlcount", "soft_format", "strict_format", "less_than_4", "correctness"}


@pytest.mark.skipif(not os.path.exists("/root/reference/reason.py"), reason="reference sources not on this machine")
def test_equal_reference_bodies(capsys):
    src = open("/root/reference/reason.py").read()
    tree = ast.parse(src)
    ns = {"re": re}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in NAMES:
            exec(compile(ast.Module(body=[node], type_ignores=[]), "reason.py", "exec"), ns)
    assert all(n in ns for n in NAMES)
    rng = random.Random(0)
    frag = ["<think>", "</think>", "\n", " ", "answer", "Cancer", "BRCA1", "is", "the", "a", "####", "x y z w v"]
    texts = ["".join(rng.choice(frag) for _ in range(rng.randint(0, 9))) for _ in range(300)]
    texts += ["<think>\nok\n</think>\nfine\n", "<think>\n\n</think>\n\n", "<think>a</think>b"]
    comps = [_msg(t) for t in texts]
    for n in ("less_than_4_reward_func", "strict_format_reward_func", "soft_format_reward_func", "xmlcount_reward_func"):
        assert ns[n](completions=comps) == getattr(R, n)(completions=comps), n
    for t in texts:
        assert ns["extract_xml_answer"](t) == R.extract_xml_answer(t) and ns["extract_hash_answer"](t) == R.extract_hash_answer(t)
    ans = [["cancer", "brca1", "the"] * 101]
    want = ns["correctness_reward_func"](prompts=[[{"content": "q"}]], completions=comps, answer=ans)
    capsys.readouterr()                       # the reference prints its debug block
    assert want == R.correctness_reward_func(prompts=[[{"content": "q"}]], completions=comps, answer=ans)


def test_text_reward_adapter():
    import torch

    class Tok:
        pad_token_id = 0

        def batch_decode(self, ids, skip_special_tokens=True):
            words = {1: "<think>\n", 2: "x", 3: "\n</think>\n", 4: "cancer", 5: "\n"}
            return ["".join(words.get(i, "") for i in row) for row in ids]

    # (conversational prompts — role / content messages — as TRL's is_conversational tests them, grpo_trainer.py:646-649; the ids
    #  behind a row's EOS are pad tokens, as generate leaves them: the reference decodes completion_ids unmasked, :644)
    fn = R.text_reward_fn(Tok(), ["xmlcount", "strict_format", "correctness"], prompts=[[{"role": "user", "content": "q"}]] * 2,
                          answer=[["cancer", "cancer"]])
    ids = torch.tensor([[1, 2, 3, 4, 5], [2, 2, 0, 0, 0]])
    mask = torch.tensor([[1, 1, 1, 1, 1], [1, 1, 0, 0, 0]])
    out = fn(ids, mask)
    assert out.shape == (2, 3) and out.dtype == torch.float32
    assert out.tolist() == [[0.25, 0.5, 2.0], [0.0, 0.0, 0.0]]
