"""the rollout's prompt pass (generation.prefill) in bf16 and on the fp8 MFMA path, HIP-event timed:   python tools/prefill_fp8_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bioreason_amd import generation
dev = torch.device("cuda:0")
dims = bench.Dims(False)
model = bench.build_model(dims, dev, 0.05)
tm = model.text_model
eng = tm.ensure_packed()
P = 2180
g = torch.Generator().manual_seed(0)
emb = (torch.randn(1, P, eng.H, generator=g) * 0.02).to(torch.bfloat16).to(dev)
mask = torch.ones(1, P, dtype=torch.long, device=dev)
pos = torch.arange(P, dtype=torch.int32, device=dev).view(1, P)
def run(n=5):
    cache = generation.KVCache(eng, 1, P, dev)
    for _ in range(2): generation.prefill(tm, emb, mask, cache, pos, decode_rows=8)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): h = generation.prefill(tm, emb, mask, cache, pos, decode_rows=8)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n, h
t16, h16 = run()
tm.rollout_fp8 = True
generation.rollout_weights(tm, rows=8); torch.cuda.synchronize()
t8, h8 = run()
os.environ["BRA_FP8_PREFILL"] = "0"
t16b, _ = run()
print(f"prefill of one {P}-row prompt: bf16 (LoRA dual-K) {t16:.2f} ms, fp8 MFMA path (merged e4m3 weights) {t8:.2f} ms, bf16 again {t16b:.2f} ms; "
      f"rel(last hidden fp8, bf16) {float((h8.float() - h16.float()).norm() / h16.float().norm()):.3e}")
