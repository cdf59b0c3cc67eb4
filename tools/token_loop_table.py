"""Per-kernel table of the rollout's token loop from a `rocprofv3 --kernel-trace --stats` CSV of bench.py ("a per-kernel
table in profiles/"):   python tools/token_loop_table.py profiles/<tag>_bench_kernel_stats.csv [fp8] > profiles/<tag>_token_loop_table.md
Rows: launches per token step, mean duration, algorithmic bytes per launch, the rate they imply, and what is left after pricing the
bytes at 8 TB/s (the "fixed" part of the launch).  Qwen3-1.7B, 8 rollouts of one 2180-token prompt."""
import csv, re, sys

path = sys.argv[1]
fp8 = len(sys.argv) > 2 and sys.argv[2] == "fp8"
H, F, NQKV, V, L = 2048, 6144, 4096, 151936, 28
wb = 1 if fp8 else 2
# kernel-name pattern -> (label, algorithmic bytes per launch)
KV_PROMPT = 8 * 2180 * 128 * 2 * 2                      # K + V^T of one layer, one prompt (shared by the 8 rollouts)
KV_COMPL = 8 * 8 * 128 * 128 * 2 * 2                    # completion K / V of 8 sequences at the mean position (128 of 256)
rows = [
    (r"dec_gemm2_kernel<0, 2, 1, 0, 8, 8, 0, 1, 1, %d>" % fp8, "gate/up + SwiGLU (N = 12288, K = 2048)", 2 * F * H * wb),
    (r"dec_gemm2_kernel<1, 0, 0, 0, 8, 12, 0, 1, 1, %d>" % fp8, "down (N = 2048, K = 6144) + residual", H * F * wb),
    (r"dec_gemm2_kernel<0, 2, 0, 0, 8, 8, 0, 1, 1, %d>" % fp8, "qkv (N = 4096, K = 2048)", NQKV * H * wb),
    (r"dec_gemm2_kernel<1, 0, 0, 0, 4, 8, 0, 1, 1, %d>" % fp8, "o (N = 2048, K = 2048) + residual", H * H * wb),
    (r"dec_attn_items_kernel<128, 2>", "attention items (q/k norm + RoPE + KV append + 64-key chunks)", KV_PROMPT + KV_COMPL),
    (r"dec_attn_merge_kernel<128>", "attention merge (partials -> o rows)", 8 * 16 * 40 * 130 * 4),
    (r"dec_gemm2_kernel<0, 2, 0, 1, 8, 8, 0, 1, 1, %d>" % fp8, "lm_head (V = 151936) + tile maxima", V * H * wb + 8 * V * 4),
    (r"topk_slices_kernel<5>", "sampler: top-k over tile maxima", 8 * (V // 16) * 4),
    (r"sample_tiles_kernel", "sampler: draw + embed + rotary rows", 8 * 20 * 16 * 4 + 8 * H * 2),
]
stats = {}
for r in csv.DictReader(l for l in open(path) if not l.startswith("#")):
    stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
per_layer = {0, 1, 2, 3, 4, 5}
print(f"# Token loop, per kernel ({'fp8 e4m3 weights' if fp8 else 'bf16 weights'}) — from `{path.split('/')[-1]}`\n")
print("| kernel | launches / token step | mean µs | algorithmic MB / launch | TB/s | µs left after bytes ÷ 8 TB/s |")
print("|---|---|---|---|---|---|")
tot_us, tot_b, tot_l = 0.0, 0.0, 0
for i, (pat, label, nbytes) in enumerate(rows):
    hit = [(n, v) for n, v in stats.items() if pat in n]
    if not hit:
        print(f"| {label} | — | not in this trace | | | |")
        continue
    calls = sum(v[0] for _, v in hit)
    us = sum(v[0] * v[1] for _, v in hit) / calls / 1e3
    per_step = L if i in per_layer else 1
    tot_us += per_step * us; tot_b += per_step * nbytes; tot_l += per_step
    print(f"| {label} | {per_step} | {us:.2f} | {nbytes / 1e6:.1f} | {nbytes / us / 1e6:.2f} | {us - nbytes / 8e6:.2f} |")
print(f"| **sum** | **{tot_l}** | **{tot_us:.0f} µs of kernel time per token step** | **{tot_b / 1e9:.2f} GB** | {tot_b / tot_us / 1e6:.2f} | {tot_us - tot_b / 8e6:.0f} |")
print("\n(rocprof durations exclude the gaps between launches and run slightly long under the tracer; the bench line's `ms_per_token_step` is the "
      "HIP-event time of the whole loop on the launch stream.)")
