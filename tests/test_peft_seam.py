"""the PEFT seam of the two entry points.  `reason.py:362-392` (`_prep_for_training`) and `train_dna_qwen.py:102-177`
(`_get_target_modules`, `_prep_for_training`) are taken from the reference's own source text (ast; those files import trl / lightning,
which are absent) and executed UNMODIFIED against the HIP model, with `compat/peft` providing the three names they import from
`peft`.  Then one SFT step must change the adapters the script attached — a wrapper the engine ignores would leave them untouched."""
import ast
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REF_REASON = "/root/reference/reason.py"
REF_TRAIN = "/root/reference/train_dna_qwen.py"
HAVE_REF = os.path.exists(REF_REASON) and os.path.exists(REF_TRAIN)
needs_ref = pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present (its source text is never copied into this repository)")


def _shim():
    """`import peft` as a script would see it with <repo>/compat first on the path"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("peft_shim_under_test", os.path.join(ROOT, "compat", "peft", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _sources():
    """the four functions' source text, read from the reference checkout (build container only)"""
    import textwrap
    out = {}
    src = open(REF_REASON).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in ("_get_target_modules", "_prep_for_training"):
            out["reason." + node.name] = ast.get_source_segment(src, node)
    src = open(REF_TRAIN).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef):
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name in ("_get_target_modules", "_prep_for_training"):
                    out["train." + sub.name] = textwrap.dedent(ast.get_source_segment(src, sub, padded=True))
    return out


def _tiny(backend):
    """the tiny_b model of the parity tests, without adapters"""
    from test_model_parity import GOLD as G, build
    fix = torch.load(os.path.join(G, "tiny_b.pt"), weights_only=False)
    m = build(fix, backend, False)
    m._fix = fix
    return m


def one_step(m, backend):
    from bioreason_amd.trainer import SFTStepRunner
    from test_model_parity import to_dev
    m.train()
    SFTStepRunner(m, learning_rate=1e-2, weight_decay=0.0).step(to_dev(m._fix["batch"], backend))


def _check_and_step(m, backend):
    from bioreason_amd.modeling import Qwen3ForCausalLM
    assert isinstance(m.text_model, Qwen3ForCausalLM) and m.text_model.active_adapter == "default"
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    lora = [n for n in names if "lora_" in n]
    assert lora and all(n.startswith(("text_model.", "dna_projection.")) for n in names)
    assert any(".q_proj.lora_A.default.weight" in n for n in lora) and any(".down_proj.lora_B.default.weight" in n for n in lora)
    assert not any(p.requires_grad for p in m.dna_model.parameters())
    assert all(p.requires_grad for p in m.dna_projection.parameters())
    assert m.text_model.base_model.model is m.text_model and m.text_model.base_model.config is m.text_model.config      # reason.py:71-72
    # one SFT step moves the adapters (B starts at zero: its gradient is the first thing that can move)
    b_before = {n: p.detach().clone() for n, p in m.named_parameters() if "lora_B" in n}
    one_step(m, backend)
    moved = sum(int(not torch.equal(p.detach(), b_before[n])) for n, p in m.named_parameters() if "lora_B" in n)
    assert moved == len(b_before) and moved >= 7


def test_the_calls_of_the_scripts_in_their_order(backend):
    """what `_prep_for_training` of both scripts does (reason.py:362-392, train_dna_qwen.py:136-177), restated: freeze the DNA encoder,
    LoraConfig over the last name components of every Linear but lm_head, prepare_model_for_kbit_training, get_peft_model, projection
    trainable — through the names `compat/peft` exports.  Runs where the reference checkout is absent (the GPU box)."""
    shim = _shim()
    m = _tiny(backend)
    for p in m.dna_model.parameters():
        p.requires_grad = False
    targets = []
    for name, module in m.text_model.named_modules():
        if isinstance(module, torch.nn.Linear):
            last = name.split(".")[-1]
            if last != "lm_head" and last not in targets:
                targets.append(last)
    targets += [t for t in ("q_proj", "k_proj", "v_proj", "out_proj", "query", "key", "value") if t not in targets]
    cfg = shim.LoraConfig(r=32, lora_alpha=64, lora_dropout=0.0, target_modules=targets, init_lora_weights="gaussian", bias="none",
                          task_type="CAUSAL_LM")
    m.text_model = shim.prepare_model_for_kbit_training(m.text_model)
    m.text_model = shim.get_peft_model(m.text_model, cfg)
    for p in m.dna_projection.parameters():
        p.requires_grad = True
    assert set(targets) >= {"q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"} and "lm_head" not in targets
    _check_and_step(m, backend)
    with pytest.raises(ValueError):
        shim.get_peft_model(m.text_model, cfg)                   # adapters twice: merge_and_unload() first


@needs_ref
@pytest.mark.parametrize("script", ["reason", "train"])
def test_script_prep_for_training_attaches_adapters_the_engine_runs(backend, script):
    shim = _shim()
    src = _sources()
    m = _tiny(backend)
    ns = {"LoraConfig": shim.LoraConfig, "get_peft_model": shim.get_peft_model,
          "prepare_model_for_kbit_training": shim.prepare_model_for_kbit_training, "torch": torch}
    if script == "reason":
        exec(src["reason._get_target_modules"], ns)
        exec(src["reason._prep_for_training"], ns)
        args = types.SimpleNamespace(lora_r=32, lora_alpha=64, lora_dropout=0.0)
        cfg = ns["_prep_for_training"](m, args, dna_model_finetune=False)
    else:
        exec(src["train._get_target_modules"], ns)
        exec(src["train._prep_for_training"], ns)
        self = types.SimpleNamespace(model=m, text_model=m.text_model, dna_model=m.dna_model, dna_projection=m.dna_projection,
                                     dna_model_finetune=False, dna_is_evo2=False, text_model_finetune=True, lora_rank=32, lora_alpha=64,
                                     lora_dropout=0.0)
        self._get_target_modules = lambda: ns["_get_target_modules"](self)
        cfg = ns["_prep_for_training"](self)
        assert self.text_model is m.text_model                   # the script rebinds ITS attribute to what get_peft_model returned
    assert cfg.r == 32 and "q_proj" in cfg.target_modules and "lm_head" not in cfg.target_modules
    _check_and_step(m, backend)


def test_a_foreign_peft_wrapper_is_refused(backend):
    import torch.nn as nn
    m = _tiny(backend)

    class PeftModelForCausalLM(nn.Module):                        # what real peft returns: a wrapper holding the model
        def __init__(self, inner):
            super().__init__()
            self.base_model = nn.Module()
            self.base_model.model = inner

    with pytest.raises(TypeError, match="PEFT wrapper"):
        m.text_model = PeftModelForCausalLM(m.text_model)


def test_non_hip_models_need_the_real_peft(backend):
    import torch.nn as nn
    shim = _shim()
    with pytest.raises(ImportError):
        shim.get_peft_model(nn.Linear(4, 4), shim.LoraConfig(r=4))


def test_the_scripts_import_line_resolves_with_compat_first_on_the_path():
    """`from peft import get_peft_model, LoraConfig, prepare_model_for_kbit_training` (reason.py:25, train_dna_qwen.py:16) and
    `from peft import PeftModel` (reason.py:430) in a fresh interpreter whose path starts with <repo>/compat"""
    import subprocess
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "compat"), ROOT]))
    code = ("from peft import get_peft_model, LoraConfig, prepare_model_for_kbit_training\n"
            "from peft import PeftModel\nimport peft, bioreason_amd.peft_compat as c\n"
            "assert peft.get_peft_model is c.get_peft_model and peft.prepare_model_for_kbit_training is c.prepare_model_for_kbit_training\n"
            "print('ok', LoraConfig(r=4).r)")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd="/tmp", capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok 4"), r.stderr[-2000:]
