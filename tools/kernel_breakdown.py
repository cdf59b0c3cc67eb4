"""Group a `rocprofv3 --kernel-trace --stats` CSV of a bench.py run by kernel family and print ms per step:
    python tools/kernel_breakdown.py <kernel_stats.csv> <steps traced> [top N names]
(steps traced = warm-up + timed + instrumented steps of the traced command; the table is kernel time, not wall time: chains overlap.)"""
import csv, sys

path, steps = sys.argv[1], float(sys.argv[2])
topn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
FAM = [
    ("dec_gemm2", "decode projections / lm_head"), ("dec_attn", "decode attention"), ("topk_slices", "sampler"), ("sample_", "sampler"),
    ("gemm_ring", "GEMM ring 256x256"), ("gemm_glds", "GEMM glds"), ("gemm_w4", "GEMM w4"), ("gemm_nt", "GEMM nt (register-staged)"),
    ("attn_fwd4", "attention fwd (4-wave)"), ("attn_dq4", "attention dQ (4-wave)"), ("attn_dkv4", "attention dK/dV (4-wave)"),
    ("attn_fwd", "attention fwd (8-wave)"), ("attn_bwd", "attention bwd (8-wave)"), ("attn_", "attention helpers (combine / delta / sum)"),
    ("transpose", "head transposes"), ("lora_", "LoRA branch"), ("wgrad", "LoRA weight gradients"), ("lmhead", "lm_head fused log-prob / CE"),
    ("lse", "lm_head fused log-prob / CE"), ("rmsnorm", "norms"), ("layernorm", "norms"), ("norm", "norms / qk-norm+rope"), ("rope", "norms / qk-norm+rope"),
    ("swiglu", "SwiGLU"), ("group_", "group sum / broadcast"), ("adamw", "AdamW"), ("Cijk", "hipBLASLt / rocBLAS"),
]
rows = []
for r in csv.DictReader(l for l in open(path) if not l.startswith("#")):
    rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"])))
tot = sum(r[2] for r in rows)
fam = {}
for n, c, t, a in rows:
    lab = next((lab for key, lab in FAM if key in n), "other")
    f = fam.setdefault(lab, [0, 0.0])
    f[0] += c; f[1] += t
print(f"# kernel time by family, {path.split('/')[-1]}, {steps:g} steps traced: {tot / 1e6 / steps:.1f} ms of kernel time per step\n")
print("| family | launches / step | ms / step | share |")
print("|---|---|---|---|")
for lab, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"| {lab} | {c / steps:.0f} | {t / 1e6 / steps:.2f} | {100 * t / tot:.1f} % |")
print("\n| kernel | launches / step | mean µs | ms / step |")
print("|---|---|---|---|")
for n, c, t, a in sorted(rows, key=lambda r: -r[2])[:topn]:
    short = n.replace("void bra::", "").split("(")[0][:90]
    print(f"| `{short}` | {c / steps:.1f} | {a / 1e3:.1f} | {t / 1e6 / steps:.2f} |")
