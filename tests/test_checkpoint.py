"""Checkpoint interchange with the reference (reason.py:46-81, 422-540; train_dna_qwen.py:963-970):
the state dict this package writes loads into the oracle (whose LoRA layers carry PEFT's parameter names) and back, in
every container layout the reference's loader accepts, and `merge_and_unload` equals PEFT's merge."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from bioreason_amd import checkpoint as C                  # noqa: E402
from test_model_parity import GOLD, build, rel, to_dev     # noqa: E402
from test_oracle import rebuild                            # noqa: E402


@pytest.fixture
def emu(backend):
    """every test runs twice: on the kernel-source emulator (CPU suite) and on the HIP library (`-m gpu`) — the loaders
    write into packed device buffers, so the device path is the one that has to round-trip"""
    return backend


def _fix(name="tiny_a"):
    return torch.load(os.path.join(GOLD, f"{name}.pt"), weights_only=False)


def _logits(m, b, dev):
    d = to_dev(b, dev)
    d.pop("labels")
    return m(**d).logits.float().cpu()


def test_saved_state_dict_has_the_reference_names_and_loads_into_the_oracle(emu, tmp_path):
    fix = _fix()
    m = build(fix, emu, True)
    path = C.save_checkpoint(m, str(tmp_path / "checkpoint-10"))
    assert os.path.basename(path) == "pytorch_model.bin" and os.path.exists(tmp_path / "checkpoint-10" / "config.json")
    sd = torch.load(path, weights_only=True)
    keys = list(sd)
    assert "dna_projection.weight" in keys and "dna_projection.bias" in keys
    q = "text_model.base_model.model.model.layers.0.self_attn.q_proj."
    assert q + "base_layer.weight" in keys and q + "lora_A.default.weight" in keys and q + "lora_B.default.weight" in keys
    assert "text_model.base_model.model.model.embed_tokens.weight" in keys
    assert any(k.startswith("dna_model.esm.encoder.layer.0.") for k in keys)
    # the oracle's LoraLinear uses PEFT's names without the PeftModel wrapper: strip it and load strictly on the text side
    ora = rebuild(fix, True)
    plain = {k.replace("text_model.base_model.model.", "text_model."): v.float() for k, v in sd.items()}
    missing, unexpected = ora.load_state_dict(plain, strict=False)
    assert not unexpected
    assert all(k.startswith("dna_model.") and ("lm_head" in k or "contact_head" in k or "inv_freq" in k) for k in missing), missing
    want = ora(**{k: v for k, v in fix["batch"].items() if k != "labels"}).logits
    keep = fix["batch"]["attention_mask"].bool()
    assert rel(_logits(m, fix["batch"], emu)[keep], want.detach()[keep]) < 2.5e-2


@pytest.mark.parametrize("layout", ["plain", "state_dict", "module", "lightning"])
def test_load_sft_checkpoint_layouts(emu, tmp_path, layout):
    """a checkpoint written from one model restores another (different random adapters) exactly, whatever the container"""
    fix = _fix()
    src = build(fix, emu, True)
    sd = {k: v.cpu().clone() for k, v in C.reference_state_dict(src).items()}
    if layout == "state_dict":
        blob = {"state_dict": {"_forward_module." + k: v for k, v in sd.items()}}
    elif layout == "module":
        blob = {"module": {"module." + k: v for k, v in sd.items()}}
    elif layout == "lightning":
        blob = {"state_dict": {"model." + k: v for k, v in sd.items()}, "epoch": torch.tensor(3)}
    else:
        blob = sd
    f = str(tmp_path / "sft.ckpt")
    torch.save(blob, f)
    dst = build(fix, emu, False)                             # no adapters yet: the loader creates them (reason.py:478-481)
    with torch.no_grad():
        dst.dna_projection.weight.zero_()
    missing, unexpected = C.load_sft_checkpoint(dst, f)
    assert not unexpected and not missing, (missing[:3], unexpected[:3])
    a, b = dict(src.state_dict()), dict(dst.state_dict())
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k].cpu(), b[k].cpu()), k
    src.eval(), dst.eval()                                   # (the loader creates adapters with the reference's lora_dropout)
    assert torch.equal(_logits(src, fix["batch"], emu), _logits(dst, fix["batch"], emu))


def test_base_only_checkpoint_into_adapter_model(emu, tmp_path):
    """reason.py:523-535: a checkpoint without LoRA keys fills the base weights of a model that already has adapters"""
    fix = _fix("tiny_b")
    src = build(fix, emu, False)
    f = str(tmp_path / "base.bin")
    torch.save({k: v.cpu().clone() for k, v in src.state_dict().items()}, f)
    dst = build(fix, emu, True)
    with torch.no_grad():
        dst.text_model.model.layers[0].mlp.down_proj.base_layer.weight.zero_()
        dst.arena.pack()
    missing, unexpected = C.load_sft_checkpoint(dst, f)
    assert not unexpected and all("lora_" in k for k in missing)
    assert torch.equal(dst.text_model.model.layers[0].mlp.down_proj.base_layer.weight.cpu(),
                       src.text_model.model.layers[0].mlp.down_proj.weight.cpu())


def test_peft_adapter_directory_is_merged(emu, tmp_path):
    """reason.py:431-444: a PEFT adapter directory is loaded and merged into the base weights (merge_and_unload): the
    merged model without adapters reproduces the adapter model; the fresh adapters (B = 0) add nothing"""
    from safetensors.torch import save_file
    fix = _fix()
    src = build(fix, emu, True)
    want = _logits(src, fix["batch"], emu)
    d = tmp_path / "adapter"
    os.makedirs(d)
    ad = {}
    for k, v in src.text_model.state_dict().items():
        if "lora_" in k:
            ad["base_model.model." + k.replace(".default.weight", ".weight")] = v.cpu().clone().contiguous()
    save_file(ad, str(d / "adapter_model.safetensors"))
    with open(d / "adapter_config.json", "w") as fh:
        json.dump({"r": 32, "lora_alpha": 64, "lora_dropout": 0.05, "peft_type": "LORA",
                   "target_modules": ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]}, fh)
    dst = build(fix, emu, False)
    missing, unexpected = C.load_sft_checkpoint(dst, str(d))
    assert not unexpected
    dst.eval()
    got = _logits(dst, fix["batch"], emu)
    keep = fix["batch"]["attention_mask"].bool()
    assert rel(got[keep], want[keep]) < 1.5e-2                    # one extra bf16 rounding of every merged weight
    with dst.text_model.disable_adapter():
        assert torch.equal(_logits(dst, fix["batch"], emu)[keep], got[keep])
    # PEFT's merge: W + bf16(s * B A), in the weight dtype
    l0s, l0d = src.text_model.model.layers[0].self_attn, dst.text_model.model.layers[0].self_attn
    A, B = l0s.v_proj.lora_A["default"].weight.float().cpu(), l0s.v_proj.lora_B["default"].weight.float().cpu()
    ref_w = (l0s.v_proj.base_layer.weight.float().cpu() + ((B.to(torch.bfloat16).float() @ A.to(torch.bfloat16).float()) * 2.0).to(torch.bfloat16).float()).to(torch.bfloat16)
    assert rel(l0d.v_proj.base_layer.weight, ref_w) < 2e-3


def test_load_pretrained_pair_from_local_directories(emu, tmp_path):
    """dna_llm.py:62-84 with local directories: config.json + model.safetensors for both models"""
    from safetensors.torch import save_file
    fix = _fix("tiny_b")
    src = build(fix, emu, False)
    for sub, mod in (("text", src.text_model), ("dna", src.dna_model)):
        os.makedirs(tmp_path / sub)
        sd = {k: v.cpu().clone().contiguous() for k, v in mod.state_dict().items() if k != "lm_head.weight"}
        save_file(sd, str(tmp_path / sub / "model.safetensors"))
        cfg = mod.config
        with open(tmp_path / sub / "config.json", "w") as fh:
            json.dump({k: v for k, v in (cfg.to_dict() if hasattr(cfg, "to_dict") else vars(cfg)).items()
                       if isinstance(v, (int, float, str, bool, type(None)))}, fh)
    text, dna, toks = C.load_pretrained_pair(str(tmp_path / "text"), str(tmp_path / "dna"), None, emu)
    for a, b in ((text, src.text_model), (dna, src.dna_model)):
        sa, sb = dict(a.state_dict()), dict(b.state_dict())
        assert set(sa) == set(sb)
        for k in sa:
            assert torch.equal(sa[k].cpu(), sb[k].cpu()), k
    from bioreason_amd.dna_llm import DNALLMModel
    m = DNALLMModel(str(tmp_path / "text"), str(tmp_path / "dna"), device=emu, dna_token_id=fix["config"]["dna_token_id"])
    with torch.no_grad():
        m.dna_projection.weight.copy_(src.dna_projection.weight)
        m.dna_projection.bias.copy_(src.dna_projection.bias)
    m.arena.pack()
    assert torch.equal(_logits(m, fix["batch"], emu), _logits(src, fix["batch"], emu))
    with pytest.raises(RuntimeError, match="not a local checkpoint directory"):
        C.load_pretrained_pair("Qwen/Qwen3-1.7B", str(tmp_path / "dna"), None, emu)


def test_nt_v2_checkpoint_with_a_plain_esm_ffn_fails_loudly(emu, tmp_path):
    """VERDICT r4 #5 / SURVEY §8c: the encoder's gated FFN is recalled from the hub file — a checkpoint whose
    `intermediate.dense.weight` is not [2F, H] (plain ESM: [F, H] + biases) must be refused, not half-loaded"""
    from safetensors.torch import save_file
    fix = _fix("tiny_b")
    src = build(fix, emu, False)
    cfg = src.dna_model.config
    H, F, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    good = {k: v.cpu().clone().contiguous() for k, v in src.dna_model.state_dict().items()}
    C.check_nt_v2_key_shapes(good, H, F, L, "good")                               # the package's own layout passes
    assert tuple(good["esm.encoder.layer.0.intermediate.dense.weight"].shape) == (2 * F, H)
    bad = dict(good)
    bad["esm.encoder.layer.0.intermediate.dense.weight"] = torch.zeros(F, H)      # ungated ESM
    bad["esm.encoder.layer.0.intermediate.dense.bias"] = torch.zeros(F)
    with pytest.raises(RuntimeError, match="not an NT-v2 checkpoint"):
        C.check_nt_v2_key_shapes(bad, H, F, L, "bad")
    os.makedirs(tmp_path / "dna")
    save_file(bad, str(tmp_path / "dna" / "model.safetensors"))
    with open(tmp_path / "dna" / "config.json", "w") as fh:
        json.dump({k: v for k, v in vars(cfg).items() if isinstance(v, (int, float, str, bool, type(None)))}, fh)
    with pytest.raises(RuntimeError, match="feed-forward tensors disagree"):
        C.load_pretrained_dna(str(tmp_path / "dna"), None, emu)
    missing = {k: v for k, v in good.items() if "layer.1.output.dense.weight" not in k}
    if L > 1:
        with pytest.raises(RuntimeError, match="missing"):
            C.check_nt_v2_key_shapes(missing, H, F, L, "missing")
