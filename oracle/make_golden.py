"""Generates tests/golden/*.pt FROM THE REFERENCE ITSELF (run in the build container only:
`python oracle/make_golden.py`; /root/reference does not exist on the GPU box).

For each tiny configuration it
  1. builds random-init HF Qwen3 + NT-v2-shaped ESM sub-models (weights rounded to bf16 so that the fp32 oracle,
     the bf16 oracle and the HIP path all start from identical numbers),
  2. instantiates the reference's UNMODIFIED `bioreason.models.dna_llm.DNALLMModel` (via __new__: its __init__
     needs the HF Hub) around those sub-models and runs forward / backward / generate on a seeded batch,
  3. checks oracle/dna_llm_oracle.OracleDNALLM (the restatement) against it BIT FOR BIT,
  4. stores weights, inputs and outputs (fp32 and bf16 runs, LoRA on/off, greedy decode, per-token log-probs,
     GRPO loss) as the golden fixture.
"""
import os
import sys
import typing

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import dna_llm_oracle as O   # noqa: E402
from oracle import grpo_math as G        # noqa: E402

CONFIGS = {
    "tiny_a": {
        "text": dict(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                     num_key_value_heads=2, head_dim=32, rope_theta=1e6, max_position_embeddings=512),
        "dna": dict(vocab_size=70, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                    max_position_embeddings=130),
        "dna_token_id": 500, "eos_token_id": 7,
        "batch": dict(seed=11, B=2, n_dna_per_sample=2, Sd=12, text_len=24, left_pad=[3, 0], dna_pad={1: 8}, label_tail=10),
        "gen_tokens": 12,
    },
    "tiny_b": {
        "text": dict(vocab_size=384, hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=2,
                     num_key_value_heads=1, head_dim=128, rope_theta=1e6, max_position_embeddings=512),
        "dna": dict(vocab_size=70, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
                    max_position_embeddings=130),
        "dna_token_id": 380, "eos_token_id": 5,
        "batch": dict(seed=23, B=3, n_dna_per_sample=1, Sd=20, text_len=30, left_pad=[0, 5, 2], dna_pad={2: 13}, label_tail=6),
        "gen_tokens": 8,
    },
}


def init_weights(model: nn.Module, seed: int):
    """activations of order one (so masks / softmax / RoPE bugs are visible), values representable in bf16"""
    g = torch.Generator().manual_seed(seed)
    for name, p in model.named_parameters():
        if p.dim() >= 2 and "embed" in name or "word_embeddings" in name:
            p.data = torch.randn(p.shape, generator=g) * 0.3
        elif p.dim() >= 2:
            p.data = torch.randn(p.shape, generator=g) * (0.8 / p.shape[1] ** 0.5)
        elif "norm" in name.lower() and name.endswith("weight"):
            p.data = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        else:
            p.data = 0.05 * torch.randn(p.shape, generator=g)
        p.data = p.data.to(torch.bfloat16).to(torch.float32)


def import_from_reference(modname: str, attr: str):
    """`attr` of the reference's module `modname`, imported from /root/reference and from nowhere else.

    The repository ships a drop-in package of the same name (`bioreason/`), so a plain `from bioreason... import` returns whichever
    of the two an earlier import left in `sys.modules` — a pin would then compare the repository with itself.  Here every
    `bioreason*` module already loaded is set aside, the reference is imported with its checkout first on `sys.path`, the object is
    checked to come from a file under REF, and then `sys.modules` / `sys.path` are put back exactly as they were (the reference's
    modules stay alive through the object's functions only), so that the caller's interpreter is left as it was found."""
    import importlib
    import inspect
    import transformers.processing_utils as pu
    if not hasattr(pu, "CommonKwargs"):           # removed in transformers 5; processing_dl.py:9 imports it
        pu.CommonKwargs = typing.TypedDict("CommonKwargs", {}, total=False)
    def ours(k): return k == "bioreason" or k.startswith("bioreason.")
    stashed = {k: sys.modules.pop(k) for k in list(sys.modules) if ours(k)}
    path_before = list(sys.path)
    sys.path.insert(0, REF)
    try:
        obj = getattr(importlib.import_module(modname), attr)
        src = os.path.realpath(inspect.getfile(obj))
        if not src.startswith(os.path.realpath(REF) + os.sep):
            raise ImportError(f"reference pin would use {src}, which is not under {REF}")
    finally:
        for k in [k for k in sys.modules if ours(k)]:
            del sys.modules[k]
        sys.modules.update(stashed)
        sys.path[:] = path_before
    return obj


def import_reference():
    return import_from_reference("bioreason.models.dna_llm", "DNALLMModel")


def build_reference(DNALLMModel, text_model, dna_model, projection, dna_token_id):
    m = DNALLMModel.__new__(DNALLMModel)
    nn.Module.__init__(m)
    m.text_model, m.dna_model, m.dna_projection = text_model, dna_model, projection
    m.dna_is_evo2, m.dna_embedding_layer = False, None
    m.text_hidden_size, m.dna_hidden_size = text_model.config.hidden_size, dna_model.config.hidden_size
    m.dna_token_id = dna_token_id
    return m


def clone_batch(b):
    return {"input_ids": b["input_ids"].clone(), "attention_mask": b["attention_mask"].clone(), "labels": b["labels"].clone(),
            "dna_tokenized": {k: v.clone() for k, v in b["dna_tokenized"].items()}, "batch_idx_map": list(b["batch_idx_map"])}


def _embeds_of(model, b):
    """the inputs_embeds the reference's generate() builds (dna_llm.py:277-295), via its own forward hooks"""
    grabbed = {}
    def hook(mod, args, kwargs):
        grabbed.setdefault("e", kwargs["inputs_embeds"].detach().clone())
        return None

    h = model.text_model.model.register_forward_pre_hook(hook, with_kwargs=True)
    with torch.no_grad():
        model(**{k: v for k, v in b.items()})
    h.remove()
    return grabbed["e"]


def run_all(model, batch, cfg, lora: bool):
    """forward (+loss, logits), backward (projection / LoRA grads), greedy generate, per-token logps, GRPO loss"""
    out = {}
    model.zero_grad(set_to_none=True)
    fw = model(**clone_batch(batch))
    out["loss"] = fw.loss.detach().float().clone()
    out["logits"] = fw.logits.detach().float().clone()
    fw.loss.backward()
    out["grad_proj_w"] = model.dna_projection.weight.grad.detach().float().clone()
    out["grad_proj_b"] = model.dna_projection.bias.grad.detach().float().clone()
    assert all(p.grad is None for p in model.dna_model.parameters())        # dna_llm.py:121
    if lora:
        l0 = model.text_model.model.layers[0]
        for nm, mod in (("q", l0.self_attn.q_proj), ("v", l0.self_attn.v_proj), ("down", l0.mlp.down_proj)):
            out[f"grad_l0_{nm}_A"] = mod.lora_A["default"].weight.grad.detach().float().clone()
            out[f"grad_l0_{nm}_B"] = mod.lora_B["default"].weight.grad.detach().float().clone()
    b = clone_batch(batch)
    b.pop("labels")
    gen = model.generate(**b, max_new_tokens=cfg["gen_tokens"], do_sample=False, eos_token_id=cfg["eos_token_id"],
                         pad_token_id=cfg["eos_token_id"])
    out["greedy_ids"] = gen.clone()
    # the per-step scores the reference took its arg-max over (for the near-tie tolerance of the decode test)
    full = model.text_model.generate(inputs_embeds=_embeds_of(model, b), attention_mask=b["attention_mask"], use_cache=True,
                                     max_new_tokens=cfg["gen_tokens"], do_sample=False, eos_token_id=cfg["eos_token_id"],
                                     pad_token_id=cfg["eos_token_id"], output_scores=True, return_dict_in_generate=True)
    assert torch.equal(full.sequences, gen)
    out["greedy_scores"] = torch.stack([s.float() for s in full.scores], dim=1).clone()
    # GRPO: prompt + completion, log-probs of the completion, loss (grpo_trainer.py:598-640, 777-814).  The completion is a
    # FIXED seeded token matrix (with an EOS inside row 0), not the greedy decode: greedy tokens of a random-init model have
    # probability ~1 (log-probs ~0, a degenerate comparison) and the bf16 run would follow its own, different, tokens
    gen = completion_of(cfg)
    out["completion"] = gen.clone()
    C = gen.shape[1]
    cmask = G.completion_mask(gen, cfg["eos_token_id"])
    pc_ids = torch.cat([b["input_ids"], gen], dim=1)
    pc_mask = torch.cat([b["attention_mask"], cmask.long()], dim=1)
    mm = {"dna_tokenized": b["dna_tokenized"], "batch_idx_map": b["batch_idx_map"]}
    P = b["input_ids"].shape[1]
    model.zero_grad(set_to_none=True)
    lp = G.per_token_logps(model, pc_ids, pc_mask, **mm)[:, P - 1:]
    out["completion_mask"] = cmask.clone()
    out["logps"] = lp.detach().float().clone()
    if lora:
        O.set_adapters(model.text_model, False)
        with torch.no_grad():
            ref_lp = G.per_token_logps(model, pc_ids, pc_mask, **mm)[:, P - 1:]
        O.set_adapters(model.text_model, True)
        out["ref_logps"] = ref_lp.float().clone()
        adv = torch.linspace(-1.0, 1.0, gen.shape[0])
        loss, kl, clip = G.grpo_loss(lp.float(), None, ref_lp.float(), adv, cmask, 0.2, 0.2, 0.04)
        loss.backward()
        out["grpo_adv"] = adv
        out["grpo_loss"] = loss.detach().clone()
        out["grpo_kl"] = kl.detach().clone()
        out["grpo_grad_proj_w"] = model.dna_projection.weight.grad.detach().float().clone()
        l0 = model.text_model.model.layers[0]
        out["grpo_grad_l0_q_A"] = l0.self_attn.q_proj.lora_A["default"].weight.grad.detach().float().clone()
        out["grpo_grad_l0_q_B"] = l0.self_attn.q_proj.lora_B["default"].weight.grad.detach().float().clone()
    return out


def completion_of(cfg):
    g = torch.Generator().manual_seed(77)
    B = cfg["batch"]["B"]
    comp = torch.randint(8, cfg["dna_token_id"] - 4, (B, cfg["gen_tokens"]), generator=g)
    comp[0, cfg["gen_tokens"] // 2] = cfg["eos_token_id"]          # row 0 ends half-way: masked tail
    return comp


def run_bf16(ref, batch, cfg, lora: bool):
    """the reference class itself with every parameter in bf16 (weights are bf16-representable, so the round trip back to
    fp32 is exact); the per-step score tensor is dropped (the fp32 one is the decode test's tie oracle)"""
    def rotary_buffers():
        return [(m, m.inv_freq) for m in ref.modules() if isinstance(getattr(m, "inv_freq", None), torch.Tensor)]

    keep = [(m, b.clone()) for m, b in rotary_buffers()]      # from_pretrained(torch_dtype=bf16) leaves inv_freq in fp32:
    ref.to(torch.bfloat16)                                    # a blanket .to() would round it (and not restore it)
    for m, b in keep:
        m.inv_freq = b.clone()
    out = run_all(ref, batch, cfg, lora=lora)
    out.pop("greedy_scores")
    ref.to(torch.float32)
    for m, b in keep:
        m.inv_freq = b.clone()
    ref.zero_grad(set_to_none=True)
    return out


def main():
    DNALLMModel = import_reference()
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    for name, cfg in CONFIGS.items():
        torch.manual_seed(0)
        text = O.make_qwen3(cfg["text"], "eager")
        dna = O.make_nt_v2(cfg["dna"], "eager")
        init_weights(text, 1)
        init_weights(dna, 2)
        text.tie_weights()
        proj = nn.Linear(cfg["dna"]["hidden_size"], cfg["text"]["hidden_size"])
        init_weights(proj, 3)
        text.eval(); dna.eval()
        batch = O.synth_batch(vocab_text=cfg["text"]["vocab_size"], vocab_dna=cfg["dna"]["vocab_size"],
                              dna_token_id=cfg["dna_token_id"], **cfg["batch"])
        ref = build_reference(DNALLMModel, text, dna, proj, cfg["dna_token_id"])
        ora = O.OracleDNALLM(text, dna, cfg["dna_token_id"])
        ora.dna_projection = proj
        fix = {"config": cfg, "batch": batch,
               "state": {"text": {k: v.clone().to(torch.bfloat16) for k, v in text.state_dict().items()},
                         "dna": {k: v.clone().to(torch.bfloat16) for k, v in dna.state_dict().items() if "lm_head" not in k and "contact_head" not in k},
                         "proj": {k: v.clone().to(torch.bfloat16) for k, v in proj.state_dict().items()}}}
        # --- no LoRA: reference vs restatement, bit for bit
        r0 = run_all(ref, batch, cfg, lora=False)
        o0 = run_all(ora, batch, cfg, lora=False)
        for k in r0:
            assert torch.equal(r0[k], o0[k]), f"{name}: restatement differs from the reference in {k}"
        # the reference's mismatch error (dna_llm.py:222-225)
        bad = clone_batch(batch)
        bad["input_ids"][0, -1] = cfg["dna_token_id"]
        for m in (ref, ora):
            try:
                m(**bad)
                raise AssertionError("mismatch not detected")
            except ValueError:
                pass
        fix["fp32"] = r0
        fix["bf16"] = run_bf16(ref, batch, cfg, lora=False)
        # --- LoRA (PEFT formula restated; B non-zero so that it matters)
        O.apply_lora(text, r=32, alpha=64.0, dropout=0.0)
        g = torch.Generator().manual_seed(5)
        lora_state = {}
        for n, p in text.named_parameters():
            if "lora_" in n:
                p.data = (torch.randn(p.shape, generator=g) * (0.5 / p.shape[1] ** 0.5)).to(torch.bfloat16).float()
                lora_state[n] = p.data.clone()
        for p in proj.parameters():
            p.requires_grad_(True)
        fix["state"]["lora"] = lora_state
        r1 = run_all(ref, batch, cfg, lora=True)
        o1 = run_all(ora, batch, cfg, lora=True)
        for k in r1:
            assert torch.equal(r1[k], o1[k]), f"{name}: restatement differs from the reference in {k} (LoRA)"
        fix["fp32_lora"] = r1
        # --- the same model in bf16 (what the reference runs on a GPU: torch_dtype=bfloat16, grpo_trainer.py:221):
        # EVERY quantity the parity tests compare, so each tolerance can be stated as a multiple of the reference's own
        # bf16-vs-fp32 distance on that quantity
        fix["bf16_lora"] = run_bf16(ref, batch, cfg, lora=True)
        path = os.path.join(ROOT, "tests", "golden", f"{name}.pt")
        if "--check-only" in sys.argv:      # tests/test_oracle.py: restatement == reference, nothing written
            print(name, "restatement equals the reference class bit for bit")
            continue
        torch.save(fix, path)
        print(name, "ok ->", path, f"{os.path.getsize(path) / 1e6:.2f} MB", "loss", float(r0["loss"]), float(r1["loss"]),
              "greedy", r1["greedy_ids"].tolist())


if __name__ == "__main__":
    main()
