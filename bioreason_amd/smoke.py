"""__graft_entry__.smoke(): one tiny forward + backward + greedy decode + GRPO step of the DNA-LLM hot path on the
given device, checked against the CPU oracle (oracle/dna_llm_oracle.py, rebuilt from the golden fixture's weights)."""
import os

import torch


def run_smoke(device) -> None:
    from . import _lib, configs
    from .dna_llm import DNALLMModel
    from .trainer import GRPOConfig, GRPOStepRunner
    lib = _lib.get_lib()
    assert not lib.emulated, "smoke must run on the HIP library"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fix = torch.load(os.path.join(root, "tests", "golden", "tiny_a.pt"), weights_only=False)
    cfg = fix["config"]
    t, d = cfg["text"], cfg["dna"]
    tc = configs.qwen3_config(**{k: t[k] for k in t})
    dc = configs.nt_v2_config(**{k: d[k] for k in d})
    m = DNALLMModel(tc, dc, device=device, dna_token_id=cfg["dna_token_id"])
    m.text_model.load_state_dict(fix["state"]["text"], strict=False)
    m.dna_model.load_state_dict(fix["state"]["dna"], strict=False)
    m.dna_projection.weight.data.copy_(fix["state"]["proj"]["weight"].float())
    m.dna_projection.bias.data.copy_(fix["state"]["proj"]["bias"].float())
    m.text_model.apply_lora(r=32, alpha=64.0, arena=m.arena)
    own = dict(m.text_model.named_parameters())
    for k, v in fix["state"]["lora"].items():
        own[k].data.copy_(v.to(device))
    m.arena.pack()
    b = fix["batch"]
    batch = {"input_ids": b["input_ids"].to(device), "attention_mask": b["attention_mask"].to(device), "labels": b["labels"].to(device),
             "dna_tokenized": {k: v.to(device) for k, v in b["dna_tokenized"].items()}, "batch_idx_map": list(b["batch_idx_map"])}
    # ---- the oracle, live, on the CPU (checker only)
    from oracle import dna_llm_oracle as O
    text = O.make_qwen3(t, "eager")
    dna = O.make_nt_v2(d, "eager")
    text.load_state_dict({k: v.float() for k, v in fix["state"]["text"].items()}, strict=False)
    dna.load_state_dict({k: v.float() for k, v in fix["state"]["dna"].items() if "inv_freq" not in k}, strict=False)   # (the bf16 fixture would round the rotary buffer)
    text.tie_weights()
    O.apply_lora(text, r=32, alpha=64.0)
    text.load_state_dict({k: v.float() for k, v in fix["state"]["lora"].items()}, strict=False)
    ora = O.OracleDNALLM(text, dna, cfg["dna_token_id"])
    ora.dna_projection.load_state_dict({k: v.float() for k, v in fix["state"]["proj"].items()})
    ora.eval()
    want = ora(**{k: v for k, v in b.items()})
    # ---- HIP path
    m.arena.zero_grad()
    out = m(**batch)
    out.loss.backward()
    keep = b["attention_mask"].bool()
    got, ref = out.logits.float().cpu()[keep], want.logits.detach()[keep]
    err = ((got - ref).norm() / ref.norm()).item()
    assert err < 3e-2, f"logits differ from the oracle: rel {err}"
    assert abs(out.loss.item() - want.loss.item()) < 3e-2 * max(1.0, abs(want.loss.item()))
    gb = {k: v for k, v in batch.items() if k != "labels"}
    gen = m.generate(**gb, max_new_tokens=cfg["gen_tokens"], do_sample=False, eos_token_id=cfg["eos_token_id"], pad_token_id=cfg["eos_token_id"])
    want_ids = fix["fp32_lora"]["greedy_ids"]
    assert gen.shape == want_ids.shape
    # greedy tokens vs the reference's, teacher-forced so that one near-tie cannot cascade: every choice must be the
    # reference's arg-max unless the reference's own margin between the two candidates is inside bf16 noise
    forced = m.generate(**gb, max_new_tokens=cfg["gen_tokens"], do_sample=False, eos_token_id=None, force_tokens=want_ids.to(device)).cpu()
    scores = fix["fp32_lora"]["greedy_scores"]
    n_tie = 0
    for bi in range(want_ids.shape[0]):
        for t in range(want_ids.shape[1]):
            ours, theirs = int(forced[bi, t]), int(want_ids[bi, t])
            if ours != theirs:
                margin = (scores[bi, t, theirs] - scores[bi, t, ours]).item()
                assert 0 <= margin < 0.02 * scores[bi, t].abs().max().item() + 0.05, f"greedy token differs from the reference: {(bi, t, ours, theirs, margin)}"
                n_tie += 1
            if theirs == cfg["eos_token_id"]:
                break
    assert n_tie <= 2, f"{n_tie} greedy positions differ from the reference"
    if n_tie == 0:
        assert torch.equal(gen.cpu(), want_ids), "free-running greedy decode differs from the reference"
    runner = GRPOStepRunner(m, GRPOConfig(num_generations=b["input_ids"].shape[0], max_completion_length=8, eos_token_id=None))
    st = runner.step(gb)
    assert torch.isfinite(st["loss_t"]).item()
    torch.cuda.synchronize()
    print(f"smoke ok: logits rel err {err:.2e}, loss {out.loss.item():.4f} (oracle {want.loss.item():.4f}), greedy {gen[0, :8].tolist()} == reference ({n_tie} near-ties), grpo loss {st['loss_t'].item():.4f}")
