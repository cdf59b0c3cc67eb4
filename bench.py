#!/usr/bin/env python
"""bench.py — GRPO training-step throughput of the HIP DNA-LLM path (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one full GRPO step on one batch of synthetic DNA+prompt input per GPU (cfg-3 of SURVEY §8d):
NT-500M encoder + Qwen3-1.7B, 1 unique prompt x G=8 rollouts per GPU, prompt P = 2180 (2 DNA sequences x 1024 NT
tokens + 128 text tokens), 256 sampled tokens per rollout (EOS suppressed so every rollout has the full length),
reference log-probs (adapters off), policy forward/backward in train mode (LoRA r=32, lora_dropout 0.05 on all 7
projections + dna_projection), reward all-gather + group advantages, gradient all-reduce, AdamW with grad clip 1.0.
Inputs are resident in HBM before the timed region.  value = samples (prompt, completion pairs) per second over all ranks.

The JSON line also carries
  roofline        — the dominant kernel, gemm_glds_kernel<*, 1> (256x128 LDS-DMA tiles; every projection / lm_head GEMM
                    of prefill, log-prob and backward passes that fills the chip): algorithmic FLOPs 2*M*N*(K+K2) of
                    exactly its launches / their duration measured with HIP events on the launch stream inside the timed
                    steps, against 2.5 PFLOP/s dense bf16 MFMA; `traffic` = HBM-side bytes per launch from the committed
                    rocprofv3 PMC passes (profiles/r1_c_pmc_gemm.json);
  decode_roofline — the HBM view of the rollout's token loop (weights + K/V bytes per token step / measured step time);
  cpu_baseline    — the oracle (reference glue + installed HF Qwen3 / ESM modules, bf16) timed on the host cores for a
                    bounded sample of the same workload (separate process, after the GPU line is measured).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (MI355X_MICROARCH.md: 2.5 PF dense; 5 PF is 2:1 sparse)
SD, TEXT_LEN, NDNA, G, C = 1024, 128, 2, 8, 256
LORA_DROPOUT = 0.05          # reason.py:266 / train_dna_qwen.py:1038


def flops_per_sample():
    """BASELINE.md §3 (algorithmic, encoder counted once per sample as in the reference's own accounting)"""
    return 34.1e12


def cpu_baseline(max_seconds: float = 60.0):
    """The oracle on the host cores, bounded: ONE sample of the cfg-3 workload at full model size —
    encoder fwd (2 x 1024), prefill P=2180, 2 of the 256 decode steps (extrapolated x128), reference log-probs
    forward and policy forward+backward over P+C.  bf16, sdpa, all cores."""
    from oracle import dna_llm_oracle as O
    ncores = min(os.cpu_count() or 1, int(os.environ.get("BENCH_CPU_THREADS", "64")))
    torch.set_num_threads(ncores)
    t_build = time.time()
    tc = dict(vocab_size=151936, hidden_size=2048, intermediate_size=6144, num_hidden_layers=28, num_attention_heads=16,
              num_key_value_heads=8, head_dim=128, rope_theta=1e6, max_position_embeddings=40960)
    dc = dict(vocab_size=4107, hidden_size=1024, intermediate_size=4096, num_hidden_layers=29, num_attention_heads=16,
              max_position_embeddings=2050)
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
    from oracle import grpo_math as GM
    t0 = time.time()
    nd = 2
    gen = model.generate(input_ids=b["input_ids"], attention_mask=b["attention_mask"], **mm, max_new_tokens=nd + 1,
                         do_sample=True, temperature=0.6, top_k=20, top_p=0.95, pad_token_id=0)
    t_roll_short = time.time() - t0
    # time one extra decode step pair to extrapolate: second call with 2*nd+1 tokens
    t0 = time.time()
    model.generate(input_ids=b["input_ids"], attention_mask=b["attention_mask"], **mm, max_new_tokens=2 * nd + 1,
                   do_sample=True, temperature=0.6, top_k=20, top_p=0.95, pad_token_id=0)
    t_roll_long = time.time() - t0
    per_step = max((t_roll_long - t_roll_short) / nd, 0.0)
    t_rollout = t_roll_short + per_step * (C - 1 - nd)
    comp = torch.randint(0, 151643, (1, C), generator=g)
    ids = torch.cat([b["input_ids"], comp], 1)
    mask = torch.ones_like(ids)
    t0 = time.time()
    O.set_adapters(text, False)
    with torch.no_grad():
        ref_lp = GM.per_token_logps(model, ids, mask, **mm)[:, -C:]
    O.set_adapters(text, True)
    t_ref = time.time() - t0
    t0 = time.time()
    lp = GM.per_token_logps(model, ids, mask, **mm)[:, -C:]
    loss, _, _ = GM.grpo_loss(lp.float(), None, ref_lp.float(), torch.ones(1), torch.ones(1, C), 0.2, 0.2, 0.04)
    loss.backward()
    t_pol = time.time() - t0
    total = t_rollout + t_ref + t_pol
    return {"value": 1.0 / total, "unit": "samples/s", "cores": ncores, "kind": "port",
            "sample": f"1 sample of the cfg-3 workload at full model size (bf16, sdpa): rollout {t_rollout:.1f}s "
                      f"(prefill + {nd} measured decode steps, extrapolated to {C}), ref logps {t_ref:.1f}s, "
                      f"policy fwd+bwd {t_pol:.1f}s; model build {build_s:.0f}s not counted"}


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (counters cannot be read from
    inside the timed process); None when the summary is absent"""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_c_pmc_gemm.json")) as fh:
            return float(json.load(fh)["traffic_bytes_per_launch"])
    except Exception:
        return None


def decode_roofline(model, rollout_profile, C):
    """HBM roofline of the rollout's token loop (the largest phase of the step; every kernel in it is a weight / KV
    stream): algorithmic bytes of one decode step = merged bf16 projection weights + tied lm_head + the K/V rows the
    step attends to (prompt rows once per prompt, completion rows per sequence, averaged over the C steps), divided by
    the measured time per step (host-synchronised wall time of the replayed hipGraph loop in the instrumented step)."""
    e = model.text_model.engine
    w_bytes = 2 * (e.L * ((e.Nq + 2 * e.Nkv) * e.H + e.H * e.Nq + 3 * e.F * e.H) + e.V * e.H)
    kv_bytes = e.L * 2 * e.Nkv * 2 * (1 * 2180 + G * (C / 2.0))
    ms = rollout_profile.get("decode_loop")
    if not ms or C < 3:
        return None
    per_step_ms = ms / (C - 1)
    ach = (w_bytes + kv_bytes) / (per_step_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
            "bytes_per_step": w_bytes + kv_bytes, "ms_per_token_step": per_step_ms,
            "kernels": "dec_gemm2_kernel<...> x4 + dec_attn_both + attn_decode_merge per layer, lm_head, sampler"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="issue the rollout's decode steps eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-shared-decode", action="store_true", help="per-copy prompt K/V in the decode attention")
    ap.add_argument("--completion-len", type=int, default=C)
    ap.add_argument("--lora-dropout", type=float, default=LORA_DROPOUT, help="PEFT lora_dropout of the policy pass (reference: 0.05)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="internal: run the oracle timing leg and print its JSON")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=dev)

    from bioreason_amd import configs, ops
    from bioreason_amd.dna_llm import DNALLMModel
    from bioreason_amd.synth import synth_prompt_batch
    from bioreason_amd.trainer import GRPOConfig, GRPOStepRunner

    model = DNALLMModel(configs.qwen3_config(), configs.nt_v2_config(), device=dev)
    model.text_model.init_weights(0.02, seed=1)          # random-init weights of the real architectures (same on every rank)
    model.dna_model.init_weights(0.02, seed=2)
    model.text_model.apply_lora(r=32, alpha=64.0, dropout=args.lora_dropout, arena=model.arena)
    model.train()                                        # HF Trainer.training_step: the policy forward / backward runs in train mode
    gen = torch.Generator().manual_seed(7)
    for n, p in model.text_model.named_parameters():     # non-zero LoRA B so the adapter path carries signal
        if "lora_B" in n:
            p.data.copy_((torch.randn(p.shape, generator=gen) * 0.01).to(dev))
    model.arena.pack()
    cfg = GRPOConfig(num_generations=G, max_completion_length=args.completion_len, eos_token_id=None, seed=42,
                     rollout_graph=False if args.no_graph else None, rollout_shared_prefix=not args.no_shared_decode)
    runner = GRPOStepRunner(model, cfg)
    batch = synth_prompt_batch(B=G, n_unique=1, Sd=SD, text_len=TEXT_LEN, n_dna=NDNA, dna_token_id=model.dna_token_id,
                               device=dev, seed=42 + rank)

    for _ in range(args.warmup):
        runner.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ops.GEMM_PROFILE = ops.GemmProfile(dominant_only=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = runner.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = ops.GEMM_PROFILE.summary()
    ops.GEMM_PROFILE = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # phase breakdown (one extra, untimed, instrumented step)
    runner.step(batch, timing=True)
    loss = float(out["loss_t"].item())

    if rank == 0:
        samples = world * G * args.steps
        value = samples / elapsed
        line = {
            "metric": "GRPO samples/sec (NT-500M+Qwen3-1.7B, DNA 2x1024, prompt 2180, gen 256)",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "GRPO step cfg-3: NT-v2-500M encoder + Qwen3-1.7B (LoRA r=32 dropout %g all linears + dna_projection), "
                                   "1 prompt x G=8 per GPU, P=2180, C=%d sampled tokens (T=0.6, top-k 20, top-p 0.95), "
                                   "ref logps + policy fwd/bwd + AdamW; random-init weights" % (args.lora_dropout, args.completion_len),
                       "global_batch": world * G, "prompt_len": 2180, "completion_len": args.completion_len, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "achieved": prof["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": prof["tflops"] / PEAK_BF16_TFLOPS, "traffic": pmc_traffic(),
                         "traffic_note": "HBM-side bytes per launch, rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + "
                                         "WRITE_SIZE), collected offline: profiles/r1_c_pmc_gemm.json",
                         "flops_per_launch": prof["flops_per_launch"], "algorithmic_bytes_per_launch": prof["bytes_per_launch"],
                         "kernel": "gemm_glds_kernel<*, 1> (256x128 LDS-DMA tiles: every projection / lm_head GEMM of the "
                                   "prefill, ref, policy forward and backward passes that fills the chip)",
                         "launches": prof["launches"], "avg_launch_ms": prof["avg_launch_ms"]},
            "decode_roofline": decode_roofline(model, runner.rollout_profile, args.completion_len),
            "rollout_issue": {"mode": "graph" if getattr(model.text_model.engine, "_rollout_use_graph", False) else "eager",
                              "probe_host_vs_device_ms_per_token": getattr(model.text_model.engine, "_rollout_probe_ms", None)},
            "step_tflops": value / world * flops_per_sample() / 1e12,
            "step_frac_of_mfma_peak": value / world * flops_per_sample() / 1e12 / PEAK_BF16_TFLOPS,
            "phases_ms": {k: round(v, 2) for k, v in runner.timers.items()},
            "loss": loss,
        }
        if not args.no_cpu_baseline and world == 1:
            try:   # separate process, hard wall-clock bound: the GPU number must be reported whatever the host does
                import subprocess
                env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True,
                                   text=True, timeout=float(os.environ.get("BENCH_CPU_TIMEOUT", "420")), env=env)
                line["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:
                line["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port",
                                        "sample": f"not measured: {type(e).__name__}"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
