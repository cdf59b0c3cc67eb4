"""Full-size probe on one MI355X: NT-500M + Qwen3-1.7B random-init, cfg-2 / cfg-3 shapes.
Prints phase timings; writes gpurun_out/e2e_probe.json."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import configs, grpo  # noqa: E402
from bioreason_amd.dna_llm import DNALLMModel  # noqa: E402
from bioreason_amd.synth import synth_prompt_batch  # noqa: E402


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def main():
    dev = torch.device("cuda:0")
    B = int(os.environ.get("PROBE_B", "8"))
    C = int(os.environ.get("PROBE_C", "256"))
    t0 = time.time()
    m = DNALLMModel(configs.qwen3_config(), configs.nt_v2_config(), device=dev)
    m.text_model.init_weights(0.02, seed=1)
    m.dna_model.init_weights(0.02, seed=2)
    m.text_model.apply_lora(r=32, alpha=64.0, dropout=0.0, arena=m.arena)
    # LoRA B non-zero so that the adapters matter
    for n, p in m.text_model.named_parameters():
        if "lora_B" in n:
            p.data.normal_(0.0, 0.01)
    m.arena.pack()
    torch.cuda.synchronize()
    print("build s", time.time() - t0, "mem GB", torch.cuda.memory_allocated() / 1e9, flush=True)
    batch = synth_prompt_batch(B=B, n_unique=1, Sd=1024, text_len=128, dna_token_id=m.dna_token_id, device=dev, seed=42)
    out = {}
    # ---- SFT-like forward/backward (cfg-2)
    labels = torch.full_like(batch["input_ids"], -100)
    labels[:, -64:] = batch["input_ids"][:, -64:]
    for it in range(3):
        m.arena.zero_grad()
        a = ev()
        o = m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], dna_tokenized=batch["dna_tokenized"],
              batch_idx_map=batch["batch_idx_map"], labels=labels, return_logits=False)
        b = ev()
        o.loss.backward()
        c = ev()
        torch.cuda.synchronize()
        out["sft_fwd_ms"], out["sft_bwd_ms"] = a.elapsed_time(b), b.elapsed_time(c)
        print("sft loss", o.loss.item(), "fwd ms", out["sft_fwd_ms"], "bwd ms", out["sft_bwd_ms"], "gradnorm", m.arena.grad_norm().item(), flush=True)
    # ---- rollout
    for it in range(2):
        a = ev()
        gen = m.generate(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], dna_tokenized=batch["dna_tokenized"],
                         batch_idx_map=batch["batch_idx_map"], dna_alias=batch["dna_alias"], max_new_tokens=C, do_sample=True,
                         temperature=0.6, top_k=20, top_p=0.95, eos_token_id=None, seed=it)
        b = ev()
        torch.cuda.synchronize()
        out["rollout_ms"] = a.elapsed_time(b)
        print("rollout ms", out["rollout_ms"], gen.shape, gen[0, :16].tolist(), flush=True)
    # ---- log-probs fwd (+bwd)
    cmask = torch.ones_like(gen, dtype=torch.int32)
    mm = {"dna_tokenized": batch["dna_tokenized"], "batch_idx_map": batch["batch_idx_map"], "dna_alias": batch["dna_alias"]}
    for it in range(2):
        m.arena.zero_grad()
        a = ev()
        with torch.no_grad(), m.text_model.disable_adapter():
            rlp = grpo.per_token_logps(m, batch["input_ids"], batch["attention_mask"], gen, cmask, **mm)
        b = ev()
        lp = grpo.per_token_logps(m, batch["input_ids"], batch["attention_mask"], gen, cmask, **mm)
        adv = torch.linspace(-1, 1, B, device=dev)
        loss, stats = grpo.grpo_loss(lp, None, rlp, adv, cmask, 0.2, 0.2, 0.04)
        c = ev()
        loss.backward()
        d = ev()
        m.arena.adamw_step(1e-5, max_grad_norm=1.0)
        e = ev()
        torch.cuda.synchronize()
        out.update(ref_lp_ms=a.elapsed_time(b), pol_fwd_ms=b.elapsed_time(c), pol_bwd_ms=c.elapsed_time(d), opt_ms=d.elapsed_time(e))
        print("grpo loss", loss.item(), stats.tolist(), {k: round(v, 2) for k, v in out.items()}, flush=True)
    out["mem_GB"] = torch.cuda.max_memory_allocated() / 1e9
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/e2e_probe.json", "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()
