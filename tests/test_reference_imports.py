"""SURVEY §8(b), the drop-in boundary as the reference's scripts see it: every `from bioreason... import name` of `reason.py:35-39`,
`train_dna_qwen.py:27-36` and of the package modules they pull in (tests/golden/reference_imports.json, written by
oracle/make_import_golden.py from the reference's own AST) must resolve with THIS repository first on PYTHONPATH — in a FRESH
interpreter, so the result cannot depend on what another test module put on sys.path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_imports.json")

CHILD = r"""
import importlib, json, sys
entries = json.load(open(sys.argv[1]))
bad = []
for e in entries:
    try:
        mod = importlib.import_module(e["module"])
    except Exception as ex:
        bad.append("%s:%d import %s -> %s: %s" % (e["file"], e["line"], e["module"], type(ex).__name__, ex))
        continue
    for name in e["names"]:
        if not hasattr(mod, name):
            bad.append("%s:%d from %s import %s -> missing" % (e["file"], e["line"], e["module"], name))
import bioreason, bioreason_amd.dna_llm, bioreason_amd.processing, bioreason_amd.dna_modules, bioreason_amd.grpo_trainer
import bioreason_amd.evo2_tokenizer, bioreason_amd.datasets, bioreason_amd.collate
from bioreason.models.dna_llm import DNALLMModel
from bioreason.models import DNALLMModel as M2, Evo2Tokenizer as T2
from bioreason.dna_modules import NucleotideDNAModule, DNABaseModule
from bioreason.dna_modules.dna_module import DNABaseModule as B2
from bioreason.models.dl.processing_dl import DLProcessor
from bioreason.trainer import DNALLMGRPOTrainer, DNALLMGRPOConfig
from bioreason.models.dl.chat_template_dl import CHAT_TEMPLATE
from bioreason.models.evo2_tokenizer import Evo2Tokenizer, register_evo2_tokenizer
from bioreason.dataset.kegg import get_format_kegg_function, qwen_dna_collate_fn
from bioreason.dataset.utils import truncate_dna
same = {
    "DNALLMModel": DNALLMModel is bioreason_amd.dna_llm.DNALLMModel and M2 is DNALLMModel,
    "DLProcessor": DLProcessor is bioreason_amd.processing.DLProcessor,
    "NucleotideDNAModule": NucleotideDNAModule is bioreason_amd.dna_modules.NucleotideDNAModule and DNABaseModule is B2,
    "DNALLMGRPOTrainer": DNALLMGRPOTrainer is bioreason_amd.grpo_trainer.DNALLMGRPOTrainer,
    "DNALLMGRPOConfig": DNALLMGRPOConfig is bioreason_amd.grpo_trainer.DNALLMGRPOConfig,
    "Evo2Tokenizer": Evo2Tokenizer is bioreason_amd.evo2_tokenizer.Evo2Tokenizer and T2 is Evo2Tokenizer,
    "qwen_dna_collate_fn": qwen_dna_collate_fn is bioreason_amd.collate.qwen_dna_collate_fn,
    "truncate_dna": truncate_dna is bioreason_amd.datasets.truncate_dna,
    "chat_template": "<|dna_pad|>" in CHAT_TEMPLATE,
    "package_is_this_repo": bioreason.__file__.startswith(sys.argv[2]),
}
print(json.dumps({"bad": bad, "same": same, "n": len(entries)}))
"""


def _run_child(extra_path=()):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, *extra_path])          # INTEGRATION.md's recipe: this repository FIRST
    env["HIP_VISIBLE_DEVICES"] = env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, "-c", CHILD, GOLDEN, ROOT], capture_output=True, text=True, timeout=300, env=env, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_every_import_of_the_reference_scripts_resolves_here():
    res = _run_child()
    assert res["n"] >= 30
    assert res["bad"] == [], "\n".join(res["bad"])
    assert all(res["same"].values()), res["same"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/bioreason"), reason="reference checkout not present")
def test_resolution_does_not_change_with_the_reference_checkout_behind_us_on_the_path():
    """the reference's own `bioreason` package further down the path must not leak in (it would fail on `import trl` anyway)"""
    res = _run_child(extra_path=("/root/reference",))
    assert res["bad"] == [] and all(res["same"].values()), res


@pytest.mark.skipif(not os.path.exists("/root/reference/reason.py"), reason="reference checkout not present")
def test_golden_import_list_is_the_reference_scripts_own():
    sys.path.insert(0, ROOT)
    from oracle import make_import_golden as G
    assert G.collect() == json.load(open(GOLDEN))
    files = {e["file"] for e in G.collect()}
    assert {"reason.py", "train_dna_qwen.py"} <= files
