"""LLM-only runs of the reference (`train_dna_qwen.py --model_type llm --max_length_text 8192 --batch_size 2`,
sh_train_dna_qwen.sh:69-86, 123-141): text-only batches of up to 8192 tokens — four times the cfg-2 / cfg-3 sequence length — through the
same `DNALLMModel.forward` / `generate` (dna_llm.py:181-306, no placeholder rows).  Qwen3-1.7B widths on 2 layers and a small vocabulary
so that the fp32 oracle stays cheap; S = 8192, B = 2 with one row left-padded by 1000 positions; logits, loss, LoRA gradients and a
greedy decode from a P = 8128 prompt (127 prompt chunks in the shared-prefix decode attention) against the oracle, each within
FACTOR x the reference's own bf16-vs-fp32 distance (the criterion of tests/test_fullsize_parity.py).
BRA_LONGTEXT_EMU=1 runs a short version (S = 384) of the same body on the kernel-source emulator."""
import os

import pytest
import torch

from bioreason_amd import configs
from test_fullsize_parity import FACTOR, _fill, rel

EMU = bool(os.environ.get("BRA_LONGTEXT_EMU"))
S, PAD, TAIL, NGEN = (384, 50, 32, 4) if EMU else (8192, 1000, 64, 8)
TC = dict(vocab_size=8192, hidden_size=2048, intermediate_size=6144, num_hidden_layers=2, num_attention_heads=16,
          num_key_value_heads=8, head_dim=128, rope_theta=1e6, max_position_embeddings=40960)
DC = dict(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2, max_position_embeddings=64)
if EMU:
    TC.update(hidden_size=256, intermediate_size=512, num_attention_heads=4, num_key_value_heads=2, head_dim=64, vocab_size=1024)
DNA_ID = TC["vocab_size"] - 5


@pytest.fixture(scope="module")
def setup():
    from bioreason_amd import _lib
    if EMU:
        from conftest import EMU_LIB, _build_emu
        _build_emu()
        _lib.use_library_for_tests(EMU_LIB)
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            pytest.skip("no GPU")
        _lib.reset_library()
        assert not _lib.get_lib().emulated
        dev = torch.device("cuda:0")
    from oracle import dna_llm_oracle as O
    from bioreason_amd.dna_llm import DNALLMModel
    from transformers.initialization import no_init_weights
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    with no_init_weights():
        text = O.make_qwen3(TC, "sdpa")
        dna = O.make_nt_v2(DC, "sdpa")
    _fill(text, 1)
    _fill(dna, 2)
    text.tie_weights()
    O.apply_lora(text, r=32, alpha=64.0, dropout=0.0)
    g = torch.Generator().manual_seed(3)
    lora_state = {}
    for n, p in text.named_parameters():
        if "lora_" in n:
            p.data = (torch.randn(p.shape, generator=g) * (0.5 / p.shape[1] ** 0.5)).to(torch.bfloat16).float()
            lora_state[n] = p.data.clone()
    ora = O.OracleDNALLM(text, dna, DNA_ID)
    ora.eval()
    ids = torch.randint(0, DNA_ID - 10, (2, S), generator=torch.Generator().manual_seed(11))
    mask = torch.ones_like(ids)
    mask[1, :PAD] = 0                               # left padding, as the processor pads text (processing_dl.py:196-204)
    ids[1, :PAD] = 0
    labels = torch.full_like(ids, -100)
    labels[:, -TAIL:] = ids[:, -TAIL:]

    def run(with_decode):
        out = {}
        ora.zero_grad(set_to_none=True)
        fw = ora(input_ids=ids, attention_mask=mask, labels=labels)
        out["loss"] = fw.loss.detach().float().clone()
        out["logits_tail"] = fw.logits[:, -TAIL:].detach().float().clone()
        fw.loss.backward()
        for li in (0, 1):
            lay = text.model.layers[li]
            for nm, mod in (("q", lay.self_attn.q_proj), ("down", lay.mlp.down_proj)):
                out[f"grad_l{li}_{nm}_A"] = mod.lora_A["default"].weight.grad.detach().float().clone()
                out[f"grad_l{li}_{nm}_B"] = mod.lora_B["default"].weight.grad.detach().float().clone()
        ora.zero_grad(set_to_none=True)
        if with_decode:
            P = S - 64
            with torch.no_grad():
                emb = text.get_input_embeddings()(ids[:, :P])
                full = text.generate(inputs_embeds=emb, attention_mask=mask[:, :P], use_cache=True, max_new_tokens=NGEN, do_sample=False,
                                     eos_token_id=None, pad_token_id=0, output_scores=True, return_dict_in_generate=True)
            out["greedy_ids"] = full.sequences.clone()
            out["greedy_scores"] = torch.stack([s.float() for s in full.scores], dim=1).clone()
        return out
    fp32 = run(True)
    keep = [(mod, mod.inv_freq.clone()) for mod in ora.modules() if isinstance(getattr(mod, "inv_freq", None), torch.Tensor)]
    ora.to(torch.bfloat16)
    for mod, buf in keep:
        mod.inv_freq = buf.clone()
    bf16 = run(False)
    m = DNALLMModel(configs.qwen3_config(**TC), configs.nt_v2_config(**DC), device=dev, dna_token_id=DNA_ID)
    base = {k.replace(".base_layer", ""): v.to(torch.bfloat16) for k, v in text.state_dict().items() if "lora_" not in k}
    missing, _ = m.text_model.load_state_dict(base, strict=False)
    assert not [k for k in missing if "lora" not in k], missing[:4]
    m.text_model.apply_lora(r=32, alpha=64.0, dropout=0.0, arena=m.arena)
    own = dict(m.text_model.named_parameters())
    for k, v in lora_state.items():
        own[k].data.copy_(v.to(dev))
    m.arena.pack()
    yield {"m": m, "dev": dev, "ids": ids, "mask": mask, "labels": labels, "fp32": fp32, "bf16": bf16}
    if EMU:
        _lib.reset_library()


def _check(name, got, s):
    want, ref = s["fp32"][name], s["bf16"][name]
    e_hip, e_ref = rel(got, want), rel(ref, want)
    assert e_hip <= FACTOR * e_ref, f"{name}: rel(hip, fp32) = {e_hip:.3e} > {FACTOR} x rel(ref_bf16, fp32) = {e_ref:.3e}"
    return e_ref


_mark = pytest.mark.skipif(False, reason="") if EMU else pytest.mark.gpu


@_mark
def test_long_text_only_forward_backward(setup):
    s = setup
    m, dev = s["m"], s["dev"]
    m.arena.zero_grad()
    m.train()
    out = m(input_ids=s["ids"].to(dev), attention_mask=s["mask"].to(dev), labels=s["labels"].to(dev))
    noise = _check("logits_tail", out.logits[:, -TAIL:], s)
    want = s["fp32"]["loss"].item()
    assert abs(out.loss.item() - want) <= FACTOR * max(abs(s["bf16"]["loss"].item() - want), noise * max(1.0, abs(want)))
    out.loss.backward()
    for li in (0, 1):
        lay = m.text_model.model.layers[li]
        for nm, mod in (("q", lay.self_attn.q_proj), ("down", lay.mlp.down_proj)):
            _check(f"grad_l{li}_{nm}_A", mod.lora_A["default"].weight.grad, s)
            _check(f"grad_l{li}_{nm}_B", mod.lora_B["default"].weight.grad, s)
    m.eval()


@_mark
@pytest.mark.parametrize("alias", [None, [0, 1]])
def test_long_prompt_greedy_decode(setup, alias):
    """P = S - 64 prompt positions per row (row 1 left-padded), NGEN greedy tokens: the per-sequence cache path (alias None) and the
    shared-prefix path with one copy per prompt (alias [0, 1]); tokens equal the fp32 oracle's except inside its own near-tie margin"""
    s = setup
    m, dev = s["m"], s["dev"]
    P = S - 64
    ids, mask = s["ids"][:, :P].to(dev), s["mask"][:, :P].to(dev)
    want, scores = s["fp32"]["greedy_ids"], s["fp32"]["greedy_scores"]
    got = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=NGEN, do_sample=False, eos_token_id=None, force_tokens=want.to(dev),
                     **({"prompt_alias": alias} if alias is not None else {})).cpu()
    assert got.shape == want.shape
    noise = rel(s["bf16"]["logits_tail"], s["fp32"]["logits_tail"])
    for b, t in (got != want).nonzero().tolist():
        a, c = int(got[b, t]), int(want[b, t])
        assert abs((scores[b, t, a] - scores[b, t, c]).item()) <= 2.0 * noise * scores[b, t].abs().max().item() + 1e-3, (b, t, a, c)
    assert int((got != want).sum()) <= max(1, NGEN // 4)


@pytest.mark.skipif(EMU, reason="this IS the emulator run")
def test_long_text_module_on_the_emulator():
    """the same module with BRA_LONGTEXT_EMU=1 (S = 384 on the kernel-source emulator) in a child process: keeps the body of the GPU
    tests above exercised by `pytest -m "not gpu"`"""
    import subprocess
    import sys
    env = dict(os.environ, BRA_LONGTEXT_EMU="1", BRA_EMU_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider"], env=env,
                       capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "3 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
