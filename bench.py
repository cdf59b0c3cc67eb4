#!/usr/bin/env python
"""bench.py — throughput of the HIP DNA-LLM hot path (BASELINE.json's metric: GRPO training-step samples/sec).

    python bench.py --gpus N --steps K --warmup W            # N > 1 self-launches one process per GPU (RCCL)
    python bench.py --mode sft                               # BASELINE config 2 (train_dna_qwen.py SFT step), secondary line
    python bench.py --eos-uniform 64 256                     # SURVEY §8d straggler run (rollout lengths U[64, 256])
    python bench.py --prompts-per-gpu 2                      # secondary lines: sh_reason.sh's per_device_train_batch_size
    BENCH_DRYRUN=1 python bench.py --gpus 2                  # launcher rehearsal without GPUs: gloo + kernel-source emulator + toy dims

The default run (N = 1) also times two SECONDARY legs after the headline steps and before the CPU leg, printed as sub-objects of
the same JSON line: `sft` (BASELINE config 2, train_dna_qwen.py:179-213) and `straggler` (EOS drawn at U[64, 256]).

GRPO "step" (default, cfg-3 of SURVEY §8d) = one full GRPO step on one batch of synthetic DNA+prompt input per GPU:
NT-500M encoder + Qwen3-1.7B, 1 unique prompt x G=8 rollouts per GPU, prompt P = 2180 (2 DNA sequences x 1024 NT
tokens + 128 text tokens), 256 sampled tokens per rollout (EOS suppressed so every rollout has the full length),
reference log-probs (adapters off), policy forward/backward in train mode (LoRA r=32, lora_dropout 0.05 on all 7
projections + dna_projection), the reward hop of grpo_trainer.py:642-676 (completion ids -> host -> decode -> the five python
reward functions reason.py enables by default -> device; a synthetic id -> text table stands in for the tokenizer files),
reward all-gather + group advantages, gradient all-reduce, AdamW with grad clip 1.0.
SFT "step" (cfg-2) = forward (full-row lm_head logits + shifted CE on the last 64 positions) / backward / all-reduce /
AdamW over B=8 distinct samples of the same shape.
Inputs are resident in HBM before the timed region.  value = samples per second over all ranks.

The JSON line also carries
  roofline        — the MFMA-bound kernel family: gemm_ring_kernel (256x256 LDS-ring tiles, most of the time) + gemm_glds_kernel
                    (256x128, under-filled grids and the row-split remainders) = every projection / lm_head GEMM of the prefill,
                    log-prob and backward passes that fills the chip: algorithmic FLOPs 2*M*N*(K+K2) of exactly those API calls /
                    their duration measured with HIP events on the launch stream inside the timed steps, against 2.5 PFLOP/s dense
                    bf16 MFMA; `traffic` = HBM-side bytes per API call (same denominator as `algorithmic_bytes_per_launch`) from the
                    committed rocprofv3 PMC passes — printed only when the profile was collected from the kernel sources that are
                    running (sha256 of csrc/k_gemm.hip + bra_device.h recorded in the profile), else null;
  roofline_decode — the HBM-bound family that is the largest by TIME: the rollout's token loop (weight-streaming projections +
                    shared-prefix attention + lm_head + sampler): algorithmic bytes per token step / measured step time;
  cpu_baseline    — the oracle (the reference's glue restated around the INSTALLED HF Qwen3 / ESM modules — the code the reference
                    itself executes on a CPU) timed on the host cores for a bounded sample of the same workload, bf16 (the
                    reference's dtype) and fp32 (separate process, after the GPU line is measured).
  gpu_baseline_hf — (round 5) the SAME oracle on THIS GPU through stock PyTorch-ROCm eager kernels (bf16, sdpa): the reference's full
                    GRPO step for 1 prompt x G = 8 (HF generate, reference log-probs, policy forward / backward with full-row logits,
                    AdamW) and its cfg-2 SFT step; its own process, after the timed region; `hip_over_hf` = value / its value.
  value_reference_semantics, policy_pass — (round 5) `value` runs the shared-prompt policy pass (opted into explicitly; under LoRA dropout
                    one mask stream for the shared rows); the same step with the reference's sampling scheme (full rows, a mask per copy:
                    the library default whenever the adapters have dropout) is `value_reference_semantics` (= leg `unshared_policy`).
  rollout_fp8, prompts_per_gpu_2 — (round 5) secondary legs: the token loop over fp8 e4m3 weights (BASELINE config 5's weight format,
                    opt-in), and 16 rows per weight stream (2 prompts x G = 8 per GPU).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (MI355X_MICROARCH.md: 2.5 PF dense; 5 PF is 2:1 sparse)
PEAK_HBM_GBS = 8000.0
SD, TEXT_LEN, NDNA, G, C = 1024, 128, 2, 8, 256
LORA_DROPOUT = 0.05          # reason.py:266 / train_dna_qwen.py:1038
SFT_LABEL_TAIL = 64
PMC_PROFILE = os.path.join(ROOT, "profiles", "r6_pmc_gemm.json")
DRYRUN = os.environ.get("BENCH_DRYRUN") == "1"


# ---------------------------------------------------------------------------------------------- accounting
def flops_per_sample_reference():
    """BASELINE.md §3 (algorithmic, the reference's own accounting: every row runs the full prompt)"""
    return 34.1e12


def executed_flops(world_local_prompts: int, P: int, Cn: int, mode: str, shared_policy: bool = True):
    """FLOPs this implementation actually executes per step and GPU (shared prompts are run once): encoder once per distinct
    sequence, prefill / reference prompt once per distinct prompt, policy forward + backward for every row."""
    d_e, f_e, L_e = 1024, 4096, 29
    d, f, L, q, kv, V = 2048, 6144, 28, 2048, 1024, 151936

    def lin(tokens):
        return tokens * L * (2 * d * q + 4 * d * kv + 2 * q * d + 6 * d * f)

    def attn(S, rows=1.0):
        return rows * L * 4 * q * S * (S + 1) / 2

    enc_seq = SD * L_e * (8 * d_e * d_e + 6 * d_e * f_e) + L_e * 4 * SD * SD * d_e
    R = world_local_prompts
    if mode == "sft":
        B = 8 * R
        fwd = lin(B * P) + attn(P, B) + 2 * d * V * B * P
        bwd = lin(B * P) + 2.5 * attn(P, B) + 2 * 2 * d * V * B * SFT_LABEL_TAIL
        return NDNA * B * enc_seq + fwd + bwd
    B = R * G
    S = P + Cn
    enc = NDNA * R * enc_seq
    prefill = lin(R * P) + attn(P, R) + 2 * d * V * R
    decode = Cn * (lin(B) + 2 * d * V * B) + B * L * 4 * q * (Cn * P + Cn * (Cn + 1) / 2)
    ref = lin(R * P + B * Cn) + attn(P, R) + B * L * 4 * q * (Cn * P + Cn * (Cn + 1) / 2) + 2 * d * V * B * Cn
    if shared_policy:        # the prompt rows of a group once (forward AND backward), the completion rows per copy
        att = attn(P, R) + B * L * 4 * q * (Cn * P + Cn * (Cn + 1) / 2)
        pol_f = lin(R * P + B * Cn) + att + 2 * d * V * B * Cn
        pol_b = lin(R * P + B * Cn) + 2.5 * att + 2 * 2 * d * V * B * Cn
    else:
        pol_f = lin(B * S) + attn(S, B) + 2 * d * V * B * Cn
        pol_b = lin(B * S) + 2.5 * attn(S, B) + 2 * 2 * d * V * B * Cn
    return enc + prefill + decode + ref + pol_f + pol_b


DECODE_SOURCES = ("k_decgemm.hip", "bra_decgemm.h", "k_decattn.hip", "bra_decattn.h", "k_decode.hip", "k_grpo.hip", "bra_device.h")


def decode_source_sha():
    h = hashlib.sha256()
    for f in DECODE_SOURCES:
        with open(os.path.join(ROOT, "bioreason_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_decode_traffic():
    """HBM-side bytes per token step of the token loop's kernels from the committed PMC summary, or None when it was collected from
    different sources (same rule as pmc_traffic)"""
    try:
        with open(PMC_PROFILE) as fh:
            d = json.load(fh)
    except Exception:
        return None
    if d.get("decode_source_sha") != decode_source_sha() or "decode_traffic_bytes_per_token_step" not in d:
        return None
    return float(d["decode_traffic_bytes_per_token_step"])


def kernel_source_sha():
    h = hashlib.sha256()
    for f in ("k_gemm.hip", "bra_device.h"):
        with open(os.path.join(ROOT, "bioreason_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic():
    """(HBM bytes per launch of the dominant kernel, note) from the committed rocprofv3 PMC summary (counters cannot be read
    from inside the timed process); traffic is None unless the summary was collected from the kernel sources now running"""
    try:
        with open(PMC_PROFILE) as fh:
            d = json.load(fh)
    except Exception:
        return None, "no PMC summary committed for this round (profiles/%s)" % os.path.basename(PMC_PROFILE)
    if d.get("kernel_source_sha") != kernel_source_sha():
        return None, ("profiles/" + os.path.basename(PMC_PROFILE) + " was collected from different kernel sources (sha %s, running %s): not reported. "
                      "Last collected value, for sha %s: %.0f bytes per API call" % (
                          d.get("kernel_source_sha"), kernel_source_sha(), d.get("kernel_source_sha"),
                          float(d.get("traffic_bytes_per_call", 0.0))))
    return float(d["traffic_bytes_per_call"]), ("HBM-side bytes per API call (one call = one ring dispatch, or ring + 256x128 remainder), rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + "
                                                  "WRITE_SIZE, separate passes) on this command: profiles/" + os.path.basename(PMC_PROFILE))


# ---------------------------------------------------------------------------------------------- CPU baseline
def cpu_model_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(mode: str = "grpo"):
    """The oracle on the host cores, bounded: ONE sample of the workload at full model size, all cores (<= 64 threads), sdpa.
    GRPO, bf16 (the reference's dtype): encoder fwd (2 x 1024) + prefill P=2180 [1 warm-up + 1 timed pass] + decode steps (median
    per-step time of 12 steps timed inside one generate call, x255) + reference log-probs forward [1 pass] + policy forward/backward
    over P+C [1 pass]; then the same sample in fp32, one pass per leg (SURVEY §8d asks for both).  About 80 s of host time per dtype:
    the token loop is extrapolated from its per-step median, never run in full.  SFT: forward + backward of one sample."""
    import torch
    from oracle import dna_llm_oracle as O
    from oracle import grpo_math as GM
    ncores = min(os.cpu_count() or 1, int(os.environ.get("BENCH_CPU_THREADS", "64")))
    torch.set_num_threads(ncores)
    t_build = time.time()
    tc = dict(vocab_size=151936, hidden_size=2048, intermediate_size=6144, num_hidden_layers=28, num_attention_heads=16,
              num_key_value_heads=8, head_dim=128, rope_theta=1e6, max_position_embeddings=40960)
    dc = dict(vocab_size=4107, hidden_size=1024, intermediate_size=4096, num_hidden_layers=29, num_attention_heads=16,
              max_position_embeddings=2050)
    if os.environ.get("BENCH_CPU_TINY") == "1":           # code-path smoke test of this leg (tests/): 2-layer modules, same shapes of input
        tc.update(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=16)
        dc.update(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2)
    from transformers.initialization import no_init_weights
    with no_init_weights():
        text = O.make_qwen3(tc, "sdpa").to(torch.bfloat16)
        dna = O.make_nt_v2(dc, "sdpa").to(torch.bfloat16)
    g = torch.Generator().manual_seed(0)
    for mdl in (text, dna):
        for p in mdl.parameters():
            p.data.uniform_(-0.03, 0.03, generator=g) if p.dim() >= 2 else p.data.fill_(1.0)
    O.apply_lora(text, r=32, alpha=64.0, dropout=LORA_DROPOUT)      # reason.py:266 lora_dropout; active in the policy pass (train mode)
    for n, p in text.named_parameters():
        if "lora_" in n:
            p.data = p.data.to(torch.bfloat16)
    model = O.OracleDNALLM(text, dna, 151670).to(torch.bfloat16)
    from bioreason_amd.synth import synth_prompt_batch
    b = synth_prompt_batch(B=1, n_unique=1, Sd=SD, text_len=TEXT_LEN, n_dna=NDNA, seed=42)
    mm = {"dna_tokenized": b["dna_tokenized"], "batch_idx_map": b["batch_idx_map"]}
    build_s = time.time() - t_build
    budget_s = float(os.environ.get("BENCH_CPU_BUDGET", "150"))
    t_start = time.time()

    def med(fn, n=3):
        """1 warm-up + up to n timed passes (fewer if the leg alone would blow the wall-clock budget)"""
        fn()
        ts = []
        for _ in range(n):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
            if time.time() - t_start > budget_s:
                break
        ts.sort()
        return ts[len(ts) // 2], len(ts)

    common = {"unit": "samples/s", "cores": ncores, "cpu_model": cpu_model_name(), "kind": "port",
              "kind_note": "'port' in the sense of the bench contract = the oracle, not a re-implementation: the reference's own glue "
                           "(dna_llm.py:103-306, pinned bit-for-bit to the reference class by oracle/make_golden.py) around the INSTALLED "
                           "transformers Qwen3 / ESM modules, i.e. the code the reference itself would execute on these cores"}
    if mode == "sft":
        labels = torch.full_like(b["input_ids"], -100)
        labels[:, -SFT_LABEL_TAIL:] = b["input_ids"][:, -SFT_LABEL_TAIL:]

        def sft_step():
            for p in model.parameters():
                p.grad = None
            out = model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=labels, **mm)
            out.loss.backward()

        t_s, n_s = med(sft_step)
        return dict(common, value=1.0 / t_s,
                    sample=f"1 sample of the cfg-2 SFT step at full model size (bf16, sdpa): forward + backward {t_s:.1f}s "
                           f"(median of {n_s} after 1 warm-up); model build {build_s:.0f}s not counted")
    gen_kw = dict(do_sample=True, temperature=0.6, top_k=20, top_p=0.95, pad_token_id=0)
    comp = torch.randint(0, 151643, (1, C), generator=g)
    ids = torch.cat([b["input_ids"], comp], 1)
    mask = torch.ones_like(ids)

    def roll(n):
        return lambda: model.generate(input_ids=b["input_ids"], attention_mask=b["attention_mask"], **mm, max_new_tokens=n, **gen_kw)

    def measure(timer, NS, timer_rest=None):
        """one sample of the cfg-3 workload, leg by leg; `timer(fn, n)` -> (seconds, passes)"""
        timer_rest = timer_rest or timer
        t_r1, n1 = timer(roll(1), 1)               # encoder + prefill + first draw
        # decode steps: timed INSIDE one generate call (every call of the text model after the prefill is one token step) — the
        # difference of two whole-rollout timings is the difference of two noisy 10-second numbers and came out anywhere between
        # 0 and 1.3 s per step on a shared host
        stamps = []
        hook = text.register_forward_hook(lambda *_: stamps.append(time.time()))
        try:
            roll(1 + NS)()
        finally:
            hook.remove()
        gaps = sorted(b_ - a_ for a_, b_ in zip(stamps[:-1], stamps[1:]))       # stamps[0] = end of the prefill forward
        per_step = gaps[len(gaps) // 2] if gaps else 0.0
        t_rollout = t_r1 + per_step * (C - 1)

        kept = {}

        def ref_pass():
            O.set_adapters(text, False)
            with torch.no_grad():
                out = GM.per_token_logps(model, ids, mask, **mm)[:, -C:]
            O.set_adapters(text, True)
            kept["ref"] = out
            return out

        t_ref, n_ref = timer_rest(ref_pass, 1)
        ref_lp = kept["ref"]

        def pol_pass():
            for p in model.parameters():
                p.grad = None
            lp = GM.per_token_logps(model, ids, mask, **mm)[:, -C:]
            loss, _, _ = GM.grpo_loss(lp.float(), None, ref_lp.float(), torch.ones(1), torch.ones(1, C), 0.2, 0.2, 0.04)
            loss.backward()

        t_pol, n_pol = timer_rest(pol_pass, 1)
        total = t_rollout + t_ref + t_pol
        return total, (f"rollout {t_rollout:.1f}s (encoder + prefill {t_r1:.1f}s [{n1} runs] + {per_step * 1e3:.0f} ms/decode step, "
                       f"median of {len(gaps)} steps timed inside one generate call, x {C - 1}), ref logps {t_ref:.1f}s [{n_ref}], "
                       f"policy fwd+bwd {t_pol:.1f}s [{n_pol}]")

    def once(fn, n):
        t0 = time.time()
        fn()
        return time.time() - t0, 1

    t16 = time.time()
    total, desc = measure(med, 12, once)
    res = dict(common, value=1.0 / total, dtype="bf16",
               sample=f"1 sample of the cfg-3 workload at full model size (bf16 — the reference's dtype, grpo_trainer.py:221 — sdpa), "
                      f"prefill after 1 warm-up pass, the other legs one pass each: {desc} (measured in {time.time() - t16:.0f}s); "
                      f"model build {build_s:.0f}s not counted")
    # fp32 leg (SURVEY §8d asks for both): the same sample with the modules in fp32, ONE cold pass per leg (no warm-up, no median:
    # the leg is bounded to about a minute of host time), skipped when the bf16 leg already used up the budget
    fp32_budget = float(os.environ.get("BENCH_CPU_FP32_BUDGET", "150"))
    if time.time() - t_start < budget_s + 60 and fp32_budget > 0:
        try:
            model.float()
            t32 = time.time()

            total32, desc32 = measure(once, 4)
            res["fp32"] = {"value": 1.0 / total32, "unit": "samples/s", "cores": ncores,
                           "sample": f"the same sample, modules in fp32, one cold pass per leg: {desc32} (measured in {time.time() - t32:.0f}s)"}
        except Exception as e:                      # the bf16 number stands whatever happens here
            res["fp32"] = {"value": None, "sample": f"not measured: {type(e).__name__}: {e}"[:200]}
    else:
        res["fp32"] = {"value": None, "sample": "not measured: the bf16 leg used the host-time budget"}
    return res


def gpu_baseline_hf():
    """SURVEY §8(d) last row / VERDICT r4 #4: the SAME oracle (the reference's glue around the installed transformers Qwen3 / ESM
    modules + the restated PEFT LoRA layer) on THIS MI355X through stock PyTorch-ROCm eager kernels (bf16, sdpa) — what the reference
    itself executes on a GPU box, and the only same-node denominator for "matching or beating".  Informational: its own sub-object,
    its own process, after the timed region of the HIP path.

    GRPO (cfg-3): the reference's full step for 1 prompt x G = 8 — `generate` over [8, P] rows (HF `_sample`, T 0.6 / top-k 20 /
    top-p 0.95, 256 new tokens, EOS suppressed as in the headline), reference log-probs with the adapters off over [8, P + C]
    (`_get_per_token_logps`: full-row lm_head logits + per-row log-softmax), policy forward / backward in train mode (LoRA dropout
    0.05, a mask per row as PEFT draws them) + `compute_loss`, AdamW over the adapters and the projection (torch.optim.AdamW, foreach).
    The DNA encoder runs inside each of the three passes, as the reference runs it.  `BENCH_HF_WARMUP` (3, as the headline) warm-up steps +
    `BENCH_HF_STEPS` (2) timed.  SFT (cfg-2): forward (full-row logits + CE) / backward / AdamW over B = 8 distinct samples, 3 + 3 steps."""
    import torch
    from oracle import dna_llm_oracle as O
    from oracle import grpo_math as GM
    from bioreason_amd.synth import synth_prompt_batch
    dev = torch.device(os.environ.get("BENCH_HF_DEVICE", "cuda:0"))       # ("cpu" + BENCH_CPU_TINY=1: the code-path test of tests/)
    on_gpu = dev.type == "cuda"
    if on_gpu:
        torch.cuda.set_device(dev)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    tc = dict(vocab_size=151936, hidden_size=2048, intermediate_size=6144, num_hidden_layers=28, num_attention_heads=16,
              num_key_value_heads=8, head_dim=128, rope_theta=1e6, max_position_embeddings=40960)
    dc = dict(vocab_size=4107, hidden_size=1024, intermediate_size=4096, num_hidden_layers=29, num_attention_heads=16,
              max_position_embeddings=2050)
    tiny = os.environ.get("BENCH_CPU_TINY") == "1"
    if tiny:
        tc.update(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=16)
        dc.update(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2)
    from transformers.initialization import no_init_weights
    t0 = time.time()
    with no_init_weights():
        text = O.make_qwen3(tc, "sdpa").to(torch.bfloat16)
        dna = O.make_nt_v2(dc, "sdpa").to(torch.bfloat16)
    text, dna = text.to(dev), dna.to(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    for mdl in (text, dna):
        for p in mdl.parameters():
            p.data.uniform_(-0.03, 0.03, generator=g) if p.dim() >= 2 else p.data.fill_(1.0)
    O.apply_lora(text, r=32, alpha=64.0, dropout=LORA_DROPOUT)
    text = text.to(dev)
    for n, p in text.named_parameters():
        if "lora_" in n:
            p.data = p.data.to(torch.bfloat16)
    model = O.OracleDNALLM(text, dna, 151670).to(dev).to(torch.bfloat16)
    for p in dna.parameters():
        p.requires_grad_(False)
    trainable = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(trainable, lr=1e-5, weight_decay=0.0)
    build_s = time.time() - t0

    def to_dev(b):
        return ({"input_ids": b["input_ids"].to(dev), "attention_mask": b["attention_mask"].to(dev)},
                {"dna_tokenized": {k: v.to(dev) for k, v in b["dna_tokenized"].items()}, "batch_idx_map": list(b["batch_idx_map"])})

    def timed(fn, warm, n):
        for _ in range(warm):
            fn()
        sync()
        t = time.time()
        for _ in range(n):
            fn()
        sync()
        return (time.time() - t) / n

    out = {"unit": "samples/s", "kind": "reference glue + installed transformers %s on stock PyTorch-ROCm %s eager kernels, bf16, sdpa, "
                                        "same MI355X" % (__import__("transformers").__version__, torch.__version__),
           "device": torch.cuda.get_device_name(0) if on_gpu else "cpu", "model_build_s": round(build_s, 1)}
    # ---- GRPO, cfg-3
    b = synth_prompt_batch(B=G, n_unique=1, Sd=SD, text_len=TEXT_LEN, n_dna=NDNA, seed=42)
    io, mm = to_dev(b)
    gen_kw = dict(do_sample=True, temperature=0.6, top_k=20, top_p=0.95, pad_token_id=0, max_new_tokens=C, min_new_tokens=C)
    phases = {}

    def grpo_step():
        t_a = time.time()
        model.eval()                                            # (HF Trainer would leave train mode on; dropout-free rollouts = the headline's)
        with torch.no_grad():
            comp = model.generate(**io, **mm, **gen_kw)
        comp = comp[:, -C:]
        ids = torch.cat([io["input_ids"], comp], 1)
        mask = torch.ones_like(ids)
        sync(); t_b = time.time()
        O.set_adapters(text, False)
        with torch.no_grad():
            ref_lp = GM.per_token_logps(model, ids, mask, **mm)[:, -C:]
        O.set_adapters(text, True)
        sync(); t_c = time.time()
        model.train()
        opt.zero_grad(set_to_none=True)
        lp = GM.per_token_logps(model, ids, mask, **mm)[:, -C:]
        adv = torch.linspace(-1.0, 1.0, G, device=dev)
        loss, _, _ = GM.grpo_loss(lp.float(), None, ref_lp.float(), adv, torch.ones(G, C, device=dev), 0.2, 0.2, 0.04)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(trainable, 1.0)
        opt.step()
        sync(); t_d = time.time()
        phases.update(rollout=t_b - t_a, ref_logps=t_c - t_b, policy_fwd_bwd_opt=t_d - t_c)

    n_steps = int(os.environ.get("BENCH_HF_STEPS", "2"))
    hf_warm = int(os.environ.get("BENCH_HF_WARMUP", "3"))          # the headline's warm-up count (the caching allocator settles after the third step)
    try:
        t_step = timed(grpo_step, hf_warm, n_steps)
        out.update(value=G / t_step, ms_per_step=1000.0 * t_step, steps=n_steps, warmup=hf_warm,
                   phases_ms={k: round(1000.0 * v, 1) for k, v in phases.items()},
                   ms_per_token_step=round(1000.0 * phases["rollout"] / C, 2),
                   workload="the reference's GRPO step, cfg-3: 1 prompt x G=8 rows of P=%d, 256 sampled tokens (HF generate), reference "
                            "log-probs, policy forward/backward over [8, P+C] rows with full-row lm_head logits (grpo_trainer.py:510-520), "
                            "LoRA r=32 dropout %g, AdamW; DNA encoder inside each pass" % (b["input_ids"].shape[1], LORA_DROPOUT))
    except Exception as e:
        import traceback
        out.update(value=None, error="%s: %s | %s" % (type(e).__name__, str(e)[:300], " <- ".join(traceback.format_exc().strip().splitlines()[-6:])[:600]))
    out["hbm_peak_gib_grpo"] = round(torch.cuda.max_memory_allocated(dev) / 2.0 ** 30, 1) if on_gpu else None
    # ---- SFT, cfg-2
    try:
        bs = synth_prompt_batch(B=8, n_unique=8, Sd=SD, text_len=TEXT_LEN, n_dna=NDNA, seed=23)
        ios, mms = to_dev(bs)
        labels = torch.full_like(ios["input_ids"], -100)
        labels[:, -SFT_LABEL_TAIL:] = ios["input_ids"][:, -SFT_LABEL_TAIL:]
        model.train()

        def sft_step():
            opt.zero_grad(set_to_none=True)
            o = model(**ios, labels=labels, **mms)
            o.loss.backward()
            torch.nn.utils.clip_grad_norm_(trainable, 1.0)
            opt.step()

        if on_gpu:
            torch.cuda.empty_cache()
        t_sft = timed(sft_step, hf_warm, 3)
        out["sft"] = {"value": 8 / t_sft, "unit": "samples/s", "ms_per_step": 1000.0 * t_sft, "steps": 3, "warmup": hf_warm,
                      "workload": "cfg-2 SFT step: B=8 distinct samples, P=%d, full-row logits + shifted CE, backward, AdamW (no gradient "
                                  "checkpointing)" % bs["input_ids"].shape[1]}
    except Exception as e:
        out["sft"] = {"value": None, "error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    out["hbm_peak_gib"] = round(torch.cuda.max_memory_allocated(dev) / 2.0 ** 30, 1) if on_gpu else None
    return out


# ---------------------------------------------------------------------------------------------- GPU legs
def fp8_weight_bytes(model):
    """bytes of the fp8 weight stream per token step: one byte per parameter of the decoder linears and the lm_head + 4 bytes per output row"""
    e = model.text_model.engine
    rows = e.L * ((e.Nq + 2 * e.Nkv) + e.H + 2 * e.F + e.H) + e.V
    params = e.L * ((e.Nq + 2 * e.Nkv) * e.H + e.H * e.Nq + 2 * e.F * e.H + e.H * e.F) + e.V * e.H
    return params + 4 * rows


def decode_roofline(model, rollout_profile, Cn, n_prompts, P, loop_ev=None):
    """HBM roofline of the rollout's token loop (the largest phase of the step by time; every kernel in it is a weight / KV
    stream): algorithmic bytes of one decode step = merged bf16 projection weights + tied lm_head + the K/V rows the
    step attends to (prompt rows once per prompt, completion rows per sequence, averaged over the C steps), divided by
    the measured time per token step: HIP events on the launch stream around the token loop of every rollout of the TIMED steps
    (`loop_ev`); the host-synchronised wall time of the loop in the instrumented step is kept beside it as a cross-check."""
    e = model.text_model.engine
    w_bytes = 2 * (e.L * ((e.Nq + 2 * e.Nkv) * e.H + e.H * e.Nq + 3 * e.F * e.H) + e.V * e.H)
    if getattr(model.text_model, "rollout_fp8", False):        # (--rollout-fp8 run: the stream is the e4m3 images + row scales)
        w_bytes = fp8_weight_bytes(model)
    kv_bytes = e.L * 2 * e.Nkv * 2 * (n_prompts * P + n_prompts * G * (Cn / 2.0))
    ms = rollout_profile.get("decode_loop")
    steps = rollout_profile.get("decode_steps", Cn - 1)
    ms_instr = (ms / steps) if (ms and steps >= 2) else None
    timing = "host-synchronised wall time of the token loop in the instrumented step"
    if loop_ev is not None:
        ms, steps = loop_ev["ms"], loop_ev["steps"]
        timing = "HIP events on the launch stream around the token loop, mean over the %d rollouts of the timed steps" % loop_ev["rollouts"]
    if not ms or steps < 2:
        return None
    per_step_ms = ms / steps
    ach = (w_bytes + kv_bytes) / (per_step_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
            "traffic": None, "bytes_per_step": w_bytes + kv_bytes, "ms_per_token_step": per_step_ms,
            "share_of_step_ms": ms, "timing": timing, "ms_per_token_step_instrumented": ms_instr,
            "kernel": "token loop of the rollout: dec_gemm2_kernel (qkv / o / gate-up+SwiGLU / down / lm_head, weight streaming at "
                      "M = 8) + dec_attn_items / dec_attn_merge + sampler; per-kernel durations: profiles/*_bench_kernel_stats.csv"}


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: re-exec under torch.distributed.run, one process per GPU,
    the way the reference launches its trainers from one command (sh_reason.sh:38 `deepspeed --num_gpus=...`)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS="1" if DRYRUN else "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


class Dims:
    """workload dimensions: BASELINE.json's cfg-2 / cfg-3, or — BENCH_DRYRUN=1 — toy dimensions the kernel-source emulator
    steps through in seconds (the dry run rehearses the launcher and the distributed plumbing, it measures nothing)"""

    def __init__(self, dry: bool):
        self.dry = dry
        if dry:
            self.sd, self.text_len, self.ndna, self.g, self.c = 8, 12, 1, 2, 4
            self.text = dict(vocab_size=256, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                             num_key_value_heads=2, head_dim=32, max_position_embeddings=512)
            self.dna = dict(vocab_size=16, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2,
                            max_position_embeddings=64)
            self.dna_token_id, self.vocab_text, self.pad, self.eos = 250, 200, 201, 202
            self.sft_rows, self.label_tail = 2, 4
        else:
            self.sd, self.text_len, self.ndna, self.g, self.c = SD, TEXT_LEN, NDNA, G, C
            self.text, self.dna = {}, {}
            self.dna_token_id, self.vocab_text, self.pad, self.eos = 151670, 151643, 151643, 151645
            self.sft_rows, self.label_tail = 8, SFT_LABEL_TAIL
        self.P = self.ndna * (self.sd + 2) + self.text_len


def build_model(dims: Dims, dev, lora_dropout: float):
    import torch
    from bioreason_amd import configs
    from bioreason_amd.dna_llm import DNALLMModel
    model = DNALLMModel(configs.qwen3_config(**dims.text), configs.nt_v2_config(**dims.dna), device=dev,
                        **({"dna_token_id": dims.dna_token_id} if dims.dry else {}))
    model.text_model.init_weights(0.05 if dims.dry else 0.02, seed=1)    # random-init weights of the real architectures (same on every rank)
    model.dna_model.init_weights(0.05 if dims.dry else 0.02, seed=2)
    model.text_model.apply_lora(r=32, alpha=64.0, dropout=lora_dropout, arena=model.arena)
    model.train()                                        # HF Trainer.training_step: the policy forward / backward runs in train mode
    gen = torch.Generator().manual_seed(7)
    for n, p in model.text_model.named_parameters():     # non-zero LoRA B so the adapter path carries signal
        if "lora_B" in n:
            p.data.copy_((torch.randn(p.shape, generator=gen) * 0.01).to(dev))
    model.arena.pack()
    return model


def make_grpo_leg(model, dims: Dims, R: int, Cn: int, rank: int, dev, args, eos_uniform, nsteps: int, share_policy=None, fp8=None):
    """-> (runner, step(i, timing), samples per step): cfg-3 GRPO step on R prompts x G rollouts of this rank"""
    import torch
    from bioreason_amd.rewards import text_reward_fn
    from bioreason_amd.synth import SyntheticTokenizer, synth_prompt_batch
    from bioreason_amd.trainer import GRPOConfig, GRPOStepRunner
    B = dims.g * R
    eos_id = dims.eos if eos_uniform else None
    cfg = GRPOConfig(num_generations=dims.g, max_completion_length=Cn, eos_token_id=eos_id, pad_token_id=dims.pad if eos_id else None,
                     seed=42, rollout_graph=False if args.no_graph else None, rollout_shared_prefix=not args.no_shared_decode,
                     share_policy_prompt=(not getattr(args, "no_shared_policy", False)) if share_policy is None else share_policy,
                     overlap_ref_pass=not getattr(args, "no_overlap_ref", False),
                     overlap_policy_chains=not getattr(args, "no_overlap_chains", False),
                     overlap_rollout_weights=not getattr(args, "no_overlap_weights", False),
                     overlap_ref_chains=bool(getattr(args, "overlap_ref_chains", False)),
                     rollout_fp8=bool(getattr(args, "rollout_fp8", False)) if fp8 is None else bool(fp8),
                     ref_fp8=bool(getattr(args, "ref_fp8", False)) if fp8 is None else bool(fp8),
                     grad_allreduce_dtype="bf16" if getattr(args, "grad_bf16", False) else "fp32")
    # the reference's reward hop runs inside the timed step: ids -> host -> decode -> the five python reward functions
    # reason.py:291-296 enables by default -> device (a synthetic id -> text table stands in for the tokenizer files)
    reward_names = ["xmlcount", "soft_format", "strict_format", "less_than_4", "correctness"]
    tok = SyntheticTokenizer(model.text_model.engine.V, pad_token_id=dims.pad, eos_token_id=dims.eos)
    runner = GRPOStepRunner(model, cfg, reward_fn=text_reward_fn(tok, reward_names, prompts=[None] * B,
                                                                 # (the reference zips the rewards with the CHARACTERS of answer[0],
                                                                 # reason.py:193-199: the string must have >= B of them)
                                                                 answer=[("adenocarcinoma " * ((B + 14) // 15 + 1))[:max(B, 15)]] * B))
    batch = synth_prompt_batch(B=B, n_unique=R, Sd=dims.sd, text_len=dims.text_len, n_dna=dims.ndna, dna_token_id=model.dna_token_id,
                               vocab_text=dims.vocab_text, vocab_dna=model.dna_model.config.vocab_size, device=dev, seed=42 + rank)
    scheds = None
    if eos_uniform:
        lo, hi = eos_uniform
        gs = torch.Generator().manual_seed(1000 + rank)
        # one schedule per step, drawn up front (length L = index of the EOS token + 1)
        scheds = [(torch.randint(lo, hi + 1, (B,), generator=gs) - 1).to(torch.int32).to(dev) for _ in range(nsteps + 1)]

    def step(i, timing=False):
        if scheds is not None:
            batch["eos_schedule"] = scheds[min(i, len(scheds) - 1)]
        return runner.step(batch, timing=timing) if timing else runner.step(batch)
    return runner, step, B


def make_sft_leg(model, dims: Dims, R: int, rank: int, dev):
    """-> (runner, step(i, timing), samples per step): cfg-2 SFT step over B distinct samples (train_dna_qwen.py:179-213)"""
    import torch
    from bioreason_amd.synth import synth_prompt_batch
    from bioreason_amd.trainer import SFTStepRunner
    B = dims.sft_rows * R
    batch = synth_prompt_batch(B=B, n_unique=B, Sd=dims.sd, text_len=dims.text_len, n_dna=dims.ndna, dna_token_id=model.dna_token_id,
                               vocab_text=dims.vocab_text, vocab_dna=model.dna_model.config.vocab_size, device=dev,
                               seed=23 + rank)              # seed 23: train_dna_qwen.py:1024
    batch.pop("dna_alias"), batch.pop("prompt_alias")
    labels = torch.full_like(batch["input_ids"], -100)
    labels[:, -dims.label_tail:] = batch["input_ids"][:, -dims.label_tail:]
    batch["labels"] = labels
    runner = SFTStepRunner(model, learning_rate=1e-4, weight_decay=0.01)

    def step(i, timing=False):
        return runner.step(batch, timing=timing) if timing else runner.step(batch)
    return runner, step, B


TIMED_DIAG: dict = {}      # diagnostics of the LAST timed_steps call (read by main() right after the headline's): host load, allocator


def timed_steps(step, steps: int, warmup: int, world: int, dev, first_index: int = 0, before_timed=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides; MAX over ranks"""
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and world == 1:
        world = 2            # BRA_DP_SINGLE_RANK rehearsal: take the collective branches below in the one-rank group
    sync = (lambda: torch.cuda.synchronize()) if dev.type == "cuda" else (lambda: None)
    for i in range(warmup):
        step(first_index + i)
    sync()
    if before_timed is not None:
        before_timed()
    if world > 1:
        dist.barrier()
    sync()
    diag = TIMED_DIAG
    diag.clear()
    if dev.type == "cuda":
        ms0 = torch.cuda.memory_stats(dev)
        diag["allocator_before"] = {"segments": ms0.get("segment.all.current", 0), "device_mallocs": ms0.get("num_device_alloc", 0),
                                    "alloc_retries": ms0.get("num_alloc_retries", 0), "reserved_gib": ms0.get("reserved_bytes.all.current", 0) / 2.0 ** 30}
    try:
        diag["host_loadavg_before"] = list(os.getloadavg())
    except OSError:
        pass
    per_step = []
    t0 = time.perf_counter()
    out = None
    for i in range(steps):
        ts = time.perf_counter()
        out = step(first_index + warmup + i)
        per_step.append(time.perf_counter() - ts)          # host time inside step(): no extra synchronisation is added for it
    sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    diag["host_ms_inside_step_calls"] = [round(1000.0 * t, 1) for t in per_step]
    if dev.type == "cuda":
        ms1 = torch.cuda.memory_stats(dev)
        diag["allocator_after"] = {"segments": ms1.get("segment.all.current", 0), "device_mallocs": ms1.get("num_device_alloc", 0),
                                   "alloc_retries": ms1.get("num_alloc_retries", 0), "reserved_gib": ms1.get("reserved_bytes.all.current", 0) / 2.0 ** 30}
    try:
        diag["host_loadavg_after"] = list(os.getloadavg())
        diag["host_cpus"] = os.cpu_count()
    except OSError:
        pass
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=3,
                    help="untimed warm-up steps (default 3: the caching allocator reaches its steady state after the THIRD step — streams that "
                         "record blocks delay their reuse — and a hipMalloc inside the timed region costs 8 ms per step on a quiet host, far more "
                         "on a loaded one: timed_region_diagnostics.allocator_* in the line shows whether any happened)")
    ap.add_argument("--mode", choices=["grpo", "sft"], default="grpo")
    ap.add_argument("--no-one-stream-profile", action="store_true", help="skip the extra untimed step that profiles the GEMM family with every chain on one stream")
    ap.add_argument("--prompts-per-gpu", type=int, default=1, help="distinct prompts per GPU (x G=8 rollouts each); headline = 1")
    ap.add_argument("--eos-uniform", type=int, nargs=2, metavar=("LO", "HI"), default=None,
                    help="straggler run (SURVEY §8d): every rollout ends at a length drawn from U[LO, HI]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary `sft` / `straggler` legs of the default run")
    ap.add_argument("--secondary-steps", type=int, default=5)
    ap.add_argument("--no-qwen3-4b", action="store_true", help="skip the secondary `qwen3_4b` leg (the cfg-3 step with Qwen3-4B as the text model)")
    ap.add_argument("--no-graph", action="store_true", help="issue the rollout's decode steps eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-shared-decode", action="store_true", help="per-copy prompt K/V in the decode attention")
    ap.add_argument("--overlap-ref-chains", action="store_true", help="(experiment) the two chains of the reference pass on two streams as well")
    ap.add_argument("--no-overlap-weights", action="store_true", help="build the rollout's merged / packed weight set on the main stream (no overlap with the DNA encoder)")
    ap.add_argument("--no-overlap-chains", action="store_true", help="prompt and completion chains of the shared policy pass on one stream")
    ap.add_argument("--no-overlap-ref", action="store_true", help="reference-policy pass on the main stream instead of beside the policy forward")
    ap.add_argument("--no-shared-policy", action="store_true",
                    help="policy forward / backward over every row's full prompt (independent LoRA-dropout masks per copy, as the reference draws them)")
    ap.add_argument("--rollout-fp8", action="store_true",
                    help="BASELINE config 5's weight format in the token loop: e4m3 images of the merged weights, one fp32 scale per output "
                         "row (half the streamed bytes; W8A16).  An opt-in configuration with its own parity criterion "
                         "(tests/test_fp8_rollout.py), never the bf16 headline: the default run reports it as the secondary leg `rollout_fp8`")
    ap.add_argument("--ref-fp8", action="store_true",
                    help="with --rollout-fp8: the no-grad reference pass on the fp8 MFMA path too (GRPOConfig.ref_fp8; e4m3 images of the base weights)")
    ap.add_argument("--grad-bf16", action="store_true", help="gradient buckets travel as bf16 images (GRPOConfig.grad_allreduce_dtype); N > 1 only")
    ap.add_argument("--no-w4-gemm", action="store_true",
                    help="A/B on one box: the per-shape GEMM choice without the four-wave large-tile kernel (bra_gemm_set_variant(-2))")
    ap.add_argument("--round3-kernels", action="store_true",
                    help="A/B on one box: round 3's kernel choices (256-row LDS-DMA tiles only, block-index-fastest attention grids, "
                         "full-scan sampler + advance_counters launch); a secondary measurement, never the headline")
    ap.add_argument("--one-launch-sampler", action="store_true",
                    help="A/B on one box: the tile-maxima sampler as ONE launch (debug library knob; measured neutral, profiles/r6_l_sampler_ab.txt)")
    ap.add_argument("--completion-len", type=int, default=None)
    ap.add_argument("--lora-dropout", type=float, default=LORA_DROPOUT, help="PEFT lora_dropout of the policy pass (reference: 0.05)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="internal: run the oracle timing leg and print its JSON")
    ap.add_argument("--no-gpu-baseline-hf", action="store_true", help="skip the `gpu_baseline_hf` leg (the oracle on this GPU through stock PyTorch-ROCm)")
    ap.add_argument("--gpu-baseline-hf-only", action="store_true", help="internal: run the oracle-on-GPU timing leg and print its JSON")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.mode)), flush=True)
        return
    if args.gpu_baseline_hf_only:
        print(json.dumps(gpu_baseline_hf()), flush=True)
        return
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dims = Dims(DRYRUN)
    if DRYRUN:
        # launcher rehearsal (no GPU): the same main() on the CPU through the kernel-source emulator (tests/emu), gloo instead of
        # RCCL, toy dimensions.  Exercises self_launch, argument forwarding, the WORLD_SIZE check, both collectives of the step, the
        # MAX-reduce of the elapsed time and the rank-0 JSON line; its numbers mean nothing and the line says so.
        from bioreason_amd import _lib
        emu = os.path.join(ROOT, "tests", "emu", "libbioreason_emu.so")
        if not os.path.exists(emu):
            raise SystemExit("BENCH_DRYRUN=1 needs tests/emu/libbioreason_emu.so (make -C bioreason_amd/csrc emu)")
        os.environ.setdefault("BRA_EMU_THREADS", "2")
        torch.set_num_threads(1)
        _lib.use_library_for_tests(emu)
        dev = torch.device("cpu")
        if world > 1:
            dist.init_process_group(backend="gloo")
    else:
        if torch.cuda.device_count() < max(1, min(world, local + 1)):
            raise SystemExit(f"rank {rank}: needs GPU {local}, {torch.cuda.device_count()} visible")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if world > 1 or (os.environ.get("BRA_DP_SINGLE_RANK") == "1" and "RANK" in os.environ):
            # (the second form: one-rank RCCL group on a 1-GPU box, every collective of the step issued — tools/rccl_single_rank.sh)
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group(backend="nccl", device_id=dev)

    from bioreason_amd import ops
    if (args.no_w4_gemm or args.round3_kernels or args.one_launch_sampler) and not dims.dry:
        # A/B flags pin tile variants: those knobs exist only in the debug build (include/bioreason_hip_debug.h), never in the product library
        from bioreason_amd import _lib as _bra_lib
        _bra_lib.use_debug_library()
    if args.no_w4_gemm:
        from bioreason_amd._lib import get_lib
        get_lib().call("bra_gemm_set_variant", -2)
    if args.one_launch_sampler and not dims.dry:
        from bioreason_amd._lib import get_lib
        get_lib().call("bra_sample_set_one_launch", 1)
    if args.round3_kernels:
        from bioreason_amd._lib import get_lib
        get_lib().call("bra_gemm_set_variant", -2)
        get_lib().call("bra_gemm_set_glds_rows", 256)
        get_lib().call("bra_attn_set_block_order", 1)
        os.environ["BRA_SAMPLE_TILES"] = "0"
    model = build_model(dims, dev, args.lora_dropout)
    R = args.prompts_per_gpu
    Cn = args.completion_len if args.completion_len is not None else dims.c
    nsteps = args.warmup + args.steps + 2
    if args.mode == "sft":
        runner, step, samples_per_step = make_sft_leg(model, dims, R, rank, dev)
    else:
        runner, step, samples_per_step = make_grpo_leg(model, dims, R, Cn, rank, dev, args, args.eos_uniform, nsteps)

    def before_timed():
        # the objects built so far (model, fixtures, the 152 k-entry id -> text table) leave the collector's generations, so a full
        # collection cannot land in the middle of a timed step (candidate cause of 10-15 ms outliers between two collections)
        import gc
        gc.collect()
        gc.freeze()
        if dev.type == "cuda":
            ops.GEMM_PROFILE = ops.GemmProfile(dominant_only=True)
            if hasattr(runner, "loop_events"):
                runner.loop_events = []          # generate() appends a HIP-event pair around its token loop: the timed steps only

    elapsed, out = timed_steps(step, args.steps, args.warmup, world, dev, before_timed=before_timed)
    headline_diag = dict(TIMED_DIAG)      # (the forward / backward chains are issued from Python: a loaded host or an allocator that has to
                                          #  go to the driver shows up in the step time — NOTES.md round 5; kept in the line so that a slow run explains itself)
    prof = ops.GEMM_PROFILE.summary() if ops.GEMM_PROFILE is not None else {
        "tflops": 0.0, "flops_per_launch": 0.0, "bytes_per_launch": 0.0, "launches": 0, "avg_launch_ms": 0.0}
    ops.GEMM_PROFILE = None
    loop_ev = None
    if getattr(runner, "loop_events", None):
        torch.cuda.synchronize()
        ms_l = [a.elapsed_time(b) for a, b, _ in runner.loop_events]
        st_l = [n for _, _, n in runner.loop_events]
        if ms_l and min(st_l) >= 2:
            loop_ev = {"ms": sum(ms_l) / len(ms_l), "steps": sum(st_l) / len(st_l), "rollouts": len(ms_l)}
    if hasattr(runner, "loop_events"):
        runner.loop_events = None
    # phase breakdown (extra, untimed, instrumented steps: the first one lets the allocator see the serialised path's buffers — a fresh
    # hipMalloc inside a measured phase has been seen to cost 100 ms on a loaded host —, the second one is reported)
    step(args.warmup + args.steps, timing=True)
    step(args.warmup + args.steps, timing=True)
    # N > 1 (or the one-rank RCCL rehearsal): one more untimed, overlapped step with per-bucket stamps — when each gradient bucket was
    # handed to the collective and when the step had waited for it, against the end of the backward (trainer.bucket_report)
    dp_buckets = None
    if getattr(runner, "dp", False) and dev.type == "cuda" and hasattr(runner, "bucket_report"):
        runner.trace_buckets = True
        step(args.warmup + args.steps + 2)
        dp_buckets = runner.bucket_report()
        runner.trace_buckets = False
    # the same GEMM family with the chains of the step issued on ONE stream (one more untimed step): in the timed steps the reference
    # pass and the two chains of the policy pass run concurrently, so a HIP-event pair around a launch also spans time in which the
    # kernel shares the chip (or waits for a CU) — this figure is the per-kernel one that a rocprof trace of a serial run would give
    prof_serial = None
    cfg_ = getattr(runner, "cfg", None)
    if dev.type == "cuda" and args.mode == "grpo" and cfg_ is not None and hasattr(cfg_, "overlap_policy_chains") and not args.no_one_stream_profile:
        keep = (cfg_.overlap_policy_chains, cfg_.overlap_ref_pass)
        cfg_.overlap_policy_chains = cfg_.overlap_ref_pass = False
        try:
            ops.GEMM_PROFILE = ops.GemmProfile(dominant_only=True)
            step(args.warmup + args.steps + 1)
            prof_serial = ops.GEMM_PROFILE.summary()
        finally:
            ops.GEMM_PROFILE = None
            cfg_.overlap_policy_chains, cfg_.overlap_ref_pass = keep
    loss = float(out["loss_t"].item())
    headline_timers = dict(runner.timers)
    headline_rollout = dict(getattr(runner, "rollout_profile", {}))
    metrics_t = out.get("metrics_t")

    # ---- secondary legs of the default run: the driver only ever runs the default command, so cfg-2 (SFT) and the straggler case
    # are timed here, after the headline steps, and reported as sub-objects; the headline fields are untouched
    secondary = {}
    headline_default = (args.mode == "grpo" and R == 1 and not args.eos_uniform and Cn == dims.c)
    if headline_default and world == 1 and not args.no_secondary and args.secondary_steps > 0:
        S = args.secondary_steps
        lo_hi = (2, dims.c) if dims.dry else (64, 256)
        s_runner, s_step, s_B = make_grpo_leg(model, dims, R, Cn, rank, dev, args, lo_hi, S + 4)
        s_el, _ = timed_steps(s_step, S, 3, 1, dev)                 # 3 warm-up steps: the allocator reaches its steady state for the new shapes
        mean_len = (lo_hi[0] + lo_hi[1]) / 2.0
        secondary["straggler"] = {"value": s_B * S / s_el, "unit": "samples/s", "ms_per_step": 1000.0 * s_el / S, "steps": S, "warmup": 3,
                                  "workload": "the headline GRPO step with every rollout's EOS drawn at U[%d, %d] (SURVEY §8d straggler "
                                              "run; mean completion %.0f tokens; the step waits for its longest row)" % (lo_hi + (mean_len,))}
        del s_runner, s_step
        if not dims.dry and not args.rollout_fp8:
            # BASELINE config 5's weight format on the cfg-3 step (opt-in): the token loop over e4m3 weights; everything else unchanged
            try:
                e_runner, e_step, e_B = make_grpo_leg(model, dims, R, Cn, rank, dev, args, None, S + 3, fp8=True)
                e_runner.loop_events = []
                e_el, _ = timed_steps(e_step, S, 3, 1, dev)
                torch.cuda.synchronize()
                ev = [(a.elapsed_time(b), n) for a, b, n in (e_runner.loop_events or [])][-S:]
                e_runner.loop_events = None
                e_tok = (sum(ms for ms, _ in ev) / max(1, sum(n for _, n in ev))) if ev else None
                secondary["rollout_fp8"] = {
                    "value": e_B * S / e_el, "unit": "samples/s", "ms_per_step": 1000.0 * e_el / S, "steps": S, "warmup": 3,
                    "ms_per_token_step": e_tok,
                    "weight_bytes_per_token_step": fp8_weight_bytes(model),
                    "hbm_frac_of_peak": ((fp8_weight_bytes(model) + 0.366e9) / (e_tok * 1e-3) / 1e9 / PEAK_HBM_GBS) if e_tok else None,
                    "workload": "the headline GRPO step with fp8 (OCP e4m3) images of the merged, norm-folded weights, one fp32 scale per output "
                                "row: the token loop streams them and decodes to bf16 in registers in front of the bf16 MFMAs (W8A16, lm_head "
                                "included); the rollout's PROMPT PASS and the no-grad REFERENCE pass run their projections on the fp8 MFMA path "
                                "(v_mfma_scale_f32_16x16x128_f8f6f4, W8A8 with per-token activation scales; round 6); policy pass and gradients "
                                "in bf16.  Rollouts are sampled from the quantised policy (BASELINE config 5 'fp8 weights'); parity: "
                                "tests/test_fp8_rollout.py, tests/test_fp8_gemm.py"}
                model.text_model.rollout_fp8 = False
                del e_runner, e_step
            except Exception as e:
                model.text_model.rollout_fp8 = False
                secondary["rollout_fp8"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        if not args.no_shared_policy:
            # transparency leg: the same step with the policy pass over every row's FULL prompt (what the reference executes;
            # independent LoRA-dropout masks per copy) — the headline shares the prompt rows of a group in that pass
            u_runner, u_step, u_B = make_grpo_leg(model, dims, R, Cn, rank, dev, args, None, S + 3, share_policy=False)
            u_el, _ = timed_steps(u_step, S, 3, 1, dev)
            u_ex = executed_flops(R, dims.P, Cn, "grpo", shared_policy=False)
            secondary["unshared_policy"] = {"value": u_B * S / u_el, "unit": "samples/s", "ms_per_step": 1000.0 * u_el / S, "steps": S, "warmup": 3,
                                            "step_tflops_executed": u_ex / 1e12 / (u_el / S),
                                            "workload": "the headline GRPO step with the policy forward / backward over every row's full "
                                                        "prompt (B x (P + C) rows, an independent LoRA-dropout mask per copy, as the reference "
                                                        "draws them) instead of the shared-prompt pass"}
            del u_runner, u_step
        if not dims.dry:
            # the other lever on the token loop: more rows per weight stream — sh_reason.sh's per_device_train_batch_size
            # style of 2 prompts x G = 8 per GPU (16 rows share every streamed weight byte); a different configuration, reported beside the headline
            try:
                p_runner, p_step, p_B = make_grpo_leg(model, dims, 2, Cn, rank, dev, args, None, S + 4)
                p_el, _ = timed_steps(p_step, S, 3, 1, dev)
                secondary["prompts_per_gpu_2"] = {"value": p_B * S / p_el, "unit": "samples/s", "ms_per_step": 1000.0 * p_el / S, "steps": S, "warmup": 3,
                                                  "workload": "the headline GRPO step with 2 distinct prompts x G = 8 per GPU (16 sequences per token step)"}
                del p_runner, p_step
            except Exception as e:
                secondary["prompts_per_gpu_2"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        f_runner, f_step, f_B = make_sft_leg(model, dims, R, rank, dev)
        f_el, _ = timed_steps(f_step, S, 3, 1, dev)
        f_ex = executed_flops(R, dims.P, 0, "sft")
        secondary["sft"] = {"value": f_B * S / f_el, "unit": "samples/s", "ms_per_step": 1000.0 * f_el / S, "steps": S, "warmup": 3,
                            "step_tflops_executed": f_ex / 1e12 / (f_el / S),
                            "step_frac_of_mfma_peak_executed": f_ex / 1e12 / (f_el / S) / PEAK_BF16_TFLOPS,
                            "workload": "BASELINE config 2 (train_dna_qwen.py:179-213): B=%d distinct samples, P=%d, full-row lm_head "
                                        "logits + shifted CE on the last %d positions, backward, AdamW" % (f_B, dims.P, dims.label_tail)}
        del f_runner, f_step
        if not dims.dry and not args.no_qwen3_4b:
            # the LLM half of BASELINE configs 4-5 that needs no Evo2 oracle (README.md:84 "NT-500M + Qwen3-4B"): the cfg-3 GRPO
            # step with Qwen3-4B (36 x 2560, 32 q / 8 kv heads, F = 9728) behind the same NT-v2-500M encoder.  Built after the
            # headline model is done with its legs; 1 warm-up + `q_S` timed steps.
            import gc
            from bioreason_amd import configs as _cfgs
            q_dims = Dims(False)
            q_dims.text = dict(_cfgs.QWEN3_4B)
            try:
                q_model = build_model(q_dims, dev, args.lora_dropout)
                q_S = max(1, min(3, S))
                q_runner, q_step, q_B = make_grpo_leg(q_model, q_dims, R, Cn, rank, dev, args, None, q_S + 4)
                q_el, _ = timed_steps(q_step, q_S, 3, 1, dev)
                secondary["qwen3_4b"] = {"value": q_B * q_S / q_el, "unit": "samples/s", "ms_per_step": 1000.0 * q_el / q_S, "steps": q_S, "warmup": 3,
                                         "workload": "the headline GRPO step (1 prompt x G=8, P=%d, C=%d, LoRA r=32 dropout %g, shared-prompt policy "
                                                     "pass) with Qwen3-4B as the text model: 36 layers x 2560, 32 q-heads / 8 kv-heads (32 query rows "
                                                     "per (prompt, kv-head) in the decode attention), intermediate 9728; NT-v2-500M encoder; random-init "
                                                     "weights" % (q_dims.P, Cn, args.lora_dropout)}
                del q_runner, q_step, q_model
            except Exception as e:                    # (a secondary leg must never take the headline line down with it)
                secondary["qwen3_4b"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            gc.collect()
            torch.cuda.empty_cache()

    if rank == 0:
        samples = world * samples_per_step * args.steps
        value = samples / elapsed
        traffic, traffic_note = pmc_traffic() if (headline_default and not dims.dry) else (None, "PMC passes exist for the headline configuration only")
        ex = executed_flops(R, dims.P, Cn, args.mode, shared_policy=not args.no_shared_policy)
        tail = "; random-init weights" + ("; DRY RUN: toy dimensions on the CPU kernel emulator, numbers are meaningless" if dims.dry else "")
        if args.mode == "sft":
            metric = "SFT samples/sec (NT-500M+Qwen3-1.7B, DNA 2x1024, seq 2180, batch 8)"
            workload = ("SFT step cfg-2: NT-v2-500M encoder + Qwen3-1.7B (LoRA r=32 dropout %g all linears + dna_projection), B=%d distinct "
                        "samples per GPU, P=%d, labels on the last %d positions, full-row lm_head logits + shifted CE, backward, "
                        "AdamW" % (args.lora_dropout, samples_per_step, dims.P, dims.label_tail)) + tail
        else:
            metric = "GRPO samples/sec (NT-500M+Qwen3-1.7B, DNA 2x1024, prompt 2180, gen 256)"
            pol = (" (prompt rows of a group run once: shared K/V, gradients summed over the copies; one LoRA-dropout mask stream for "
                   "the shared rows)") if not args.no_shared_policy else " (every row's full prompt)"
            workload = ("GRPO step cfg-3: NT-v2-500M encoder + Qwen3-1.7B (LoRA r=32 dropout %g all linears + dna_projection), "
                        "%d prompt x G=%d per GPU, P=%d, C=%d sampled tokens (T=0.6, top-k 20, top-p 0.95)%s, "
                        "ref logps + policy fwd/bwd%s + AdamW"
                        % (args.lora_dropout, R, dims.g, dims.P, Cn,
                           (", EOS drawn at U[%d, %d] (straggler run)" % tuple(args.eos_uniform)) if args.eos_uniform else "", pol)) + tail
        line = {
            "metric": metric,
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic" if not dims.dry else "synthetic (dry run: toy dimensions, CPU kernel emulator, gloo)",
            "config": {"workload": workload, "global_batch": world * samples_per_step, "prompt_len": dims.P,
                       "completion_len": Cn if args.mode == "grpo" else 0, "parallelism": f"dp{world}"},
            "roofline_mfma": {"bound": "mfma", "achieved": prof["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": prof["tflops"] / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                         "flops_per_launch": prof["flops_per_launch"], "algorithmic_bytes_per_launch": prof["bytes_per_launch"],
                         "kernel": "gemm_ring_kernel<...> (256x256 LDS-ring MFMA tiles) + gemm_glds_kernel<..., MI> (LDS-DMA tiles of "
                                   "256 / 192 / 128 rows x 128 columns) + gemm_w4_kernel<..., WM, WN> (four waves with 80 x 128 ... 64 x 64 per-wave "
                                   "tiles, late round 4) — one of them per call by a cost model over tiles, rounds of 256 CUs and the "
                                   "kernels' sustained rates; row-split remainders) — every projection / "
                                   "lm_head GEMM of the prefill, ref, policy forward and backward passes that fills the chip; "
                                   "one launch = one API call (bra_gemm_bf16_nt)",
                         "launches": prof["launches"], "avg_launch_ms": prof["avg_launch_ms"],
                         "kernel_source_sha": kernel_source_sha()},
            "executed_tflop_per_step": ex / 1e12,
            "step_tflops_executed": ex / 1e12 / (elapsed / args.steps),
            "step_frac_of_mfma_peak_executed": ex / 1e12 / (elapsed / args.steps) / PEAK_BF16_TFLOPS,
            "phases_ms": {k: round(v, 2) for k, v in headline_timers.items()},
            "loss": loss,
            "timed_region_diagnostics": headline_diag,
            **({"dp_buckets": dp_buckets} if dp_buckets else {}),
            "hbm_reserved_gib": (torch.cuda.max_memory_reserved(dev) / 2.0 ** 30) if dev.type == "cuda" else None,   # peak of the caching allocator over all legs run so far (of 288)
            "parity_tolerance": "bf16-noise-relative: every compared quantity within 1.25x the reference's OWN bf16-vs-fp32 distance "
                                "(tests/test_fullsize_parity.py, tests/test_model_parity.py); north_star's 1e-3 rel is below one bf16 "
                                "rounding (4e-3) and is met by NO compared quantity: the closest are the per-token log-probs at 1.4e-3 - 1.5e-3 "
                                "(profiles/r5_m_fullsize_parity_ratios.json), the logits sit at 2.1e-2 against the reference's own 2.25e-2",
        }
        if prof_serial is not None and prof_serial["launches"]:
            line["roofline_mfma"]["one_stream"] = {
                "achieved": prof_serial["tflops"], "frac": prof_serial["tflops"] / PEAK_BF16_TFLOPS, "launches": prof_serial["launches"],
                "avg_launch_ms": prof_serial["avg_launch_ms"],
                "note": "the same launches in one extra untimed step with every chain on one stream (no concurrent kernels inside the "
                        "event pairs); `achieved` above is measured inside the timed steps, where three chains share the chip"}
        if dims.dry:
            line["dryrun"] = True
        if args.no_w4_gemm:
            line["ab_note"] = "A/B run without the four-wave GEMM kernel (--no-w4-gemm): not the shipped configuration"
        if args.round3_kernels:
            line["ab_note"] = "A/B run with round 3's kernel choices (--round3-kernels): not the shipped configuration"

        # `roofline` = the kernel family that is dominant BY TIME in the step: the rollout's token loop in a GRPO step (HBM-bound weight
        # streaming: 60-70 % of the step), the MFMA GEMM family in an SFT step; the other family keeps its own key
        line["roofline"] = line["roofline_mfma"]
        if args.mode == "grpo":
            dec = decode_roofline(model, headline_rollout, Cn, R, dims.P, loop_ev)
            if dec is not None and headline_default and not dims.dry:
                dec["traffic"] = pmc_decode_traffic()
                dec["decode_source_sha"] = decode_source_sha()
            line["roofline_decode"] = dec
            step_ms = 1000.0 * elapsed / args.steps
            gemm_ms = prof["avg_launch_ms"] * prof["launches"] / max(args.steps, 1)
            if dec is not None and dec["share_of_step_ms"] > gemm_ms:
                line["roofline"] = dict(dec, dominant_by_time="token loop %.0f ms of the %.0f ms step; MFMA GEMM family %.0f ms (roofline_mfma)"
                                        % (dec["share_of_step_ms"], step_ms, gemm_ms))
            line["rollout_phases_ms"] = {k: round(v, 2) for k, v in headline_rollout.items() if isinstance(v, float)}
            line["rollout_issue"] = {"mode": "graph" if getattr(model.text_model.engine, "_rollout_use_graph", False) else "eager",
                                     "probe_host_vs_device_ms_per_token": getattr(model.text_model.engine, "_rollout_probe_ms", None)}
            # the reference's accounting (every row re-runs its full prompt and the encoder): an upper bound on executed work
            line["step_tflops_reference_accounting"] = value / world * flops_per_sample_reference() / 1e12
            line["step_frac_of_mfma_peak_reference_accounting"] = value / world * flops_per_sample_reference() / 1e12 / PEAK_BF16_TFLOPS
            if metrics_t is not None:
                line["metrics"] = {k: float(v) for k, v in zip(runner.metric_names, metrics_t.tolist())}
        line.update(secondary)
        # ---- what the headline's policy pass is, in the line itself: `value` runs the shared-prompt policy pass, opted into
        # explicitly here (the library default follows the reference: full rows whenever the adapters have dropout); the same step with the
        # reference's sampling scheme is `value_reference_semantics` (= the `unshared_policy` leg)
        if args.mode == "grpo":
            line["policy_pass"] = ("full rows, an independent LoRA-dropout mask per copy (the reference's scheme)" if args.no_shared_policy else
                                   "shared prompt rows per group (explicit opt-in: GRPOConfig.share_policy_prompt=True); under lora_dropout > 0 one "
                                   "mask stream for the shared rows — unbiased, not the reference's sampling scheme (DESIGN.md section 6)")
            if args.no_shared_policy:
                line["value_reference_semantics"] = value
            elif "unshared_policy" in secondary:
                line["value_reference_semantics"] = secondary["unshared_policy"]["value"]
            if line.get("value_reference_semantics") and not dims.dry:
                # the same two fractions the headline carries, for the step a reader of the reference would recognise
                vr = line["value_reference_semantics"] / world
                line["reference_semantics"] = {
                    "value": line["value_reference_semantics"], "unit": "samples/s", "ms_per_step": (secondary.get("unshared_policy", {}).get("ms_per_step") if not args.no_shared_policy else line.get("ms_per_step")),
                    "step_tflops_reference_accounting": vr * flops_per_sample_reference() / 1e12,
                    "step_frac_of_mfma_peak_reference_accounting": vr * flops_per_sample_reference() / 1e12 / PEAK_BF16_TFLOPS,
                    "note": "full-row policy pass, an independent LoRA-dropout mask per copy: the reference's own sampling scheme"}
        if headline_default and world == 1 and not dims.dry and not args.no_gpu_baseline_hf:
            try:   # the oracle on this GPU (stock PyTorch-ROCm), its own process; the HIP model of this process keeps its ~60 GB of the 288
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpu-baseline-hf-only"], capture_output=True, text=True,
                                   timeout=float(os.environ.get("BENCH_HF_TIMEOUT", "420")))
                hf = json.loads(r.stdout.strip().splitlines()[-1])
                if hf.get("value"):
                    hf["hip_over_hf"] = value / hf["value"]
                    if "value_reference_semantics" in line:
                        hf["hip_reference_semantics_over_hf"] = line["value_reference_semantics"] / hf["value"]
                if hf.get("sft", {}).get("value") and "sft" in secondary:
                    hf["sft"]["hip_over_hf"] = secondary["sft"]["value"] / hf["sft"]["value"]
                line["gpu_baseline_hf"] = hf
            except Exception as e:
                line["gpu_baseline_hf"] = {"value": None, "unit": "samples/s", "error": "not measured: %s: %s" % (type(e).__name__, str(e)[:200])}
        if not args.no_cpu_baseline and world == 1 and not dims.dry:
            try:   # separate process, hard wall-clock bound: the GPU number must be reported whatever the host does
                env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--mode", args.mode],
                                   capture_output=True, text=True, timeout=float(os.environ.get("BENCH_CPU_TIMEOUT", "480")), env=env)
                line["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:
                line["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port",
                                        "sample": f"not measured: {type(e).__name__}"}
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
