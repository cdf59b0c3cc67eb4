"""Attention kernels at the shapes the steps launch them with (16 q-heads / 8 kv-heads, hd 128, causal):
  full      B = 8, S = 2436            the full-row policy pass / SFT (1280 forward workgroups)
  prompt    B = 1, S = 2180            the prompt segment of a shared-prompt pass, the rollout's prefill (144 workgroups)
  compl     B = 8, Sq = 256, Sk = 2436 the completion segment: queries attend to [prompt | own] (128 workgroups)
ms and TFLOP/s (causal FLOPs = the visible half) of bra_attn_fwd and bra_attn_bwd (delta + 3 transposes + dQ + dK/dV); run under
rocprofv3 for per-kernel numbers.  AP_SHAPES=full,prompt,compl,enc selects (enc: the NT-v2 encoder's B = 16, S = 1024, 16 heads of 64,
bidirectional); AP_CHECK=1 compares the forward with an fp32 torch statement; AP_FWD4=0|1 pins the forward kernel through the debug
library (0 = the 8-wave kernel of rounds 1-5, 1 = the 4-wave kernel of k_attn4.hip); AP_FWD_ONLY=1 skips the backward."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops, _lib
if os.environ.get("AP_LIB"):                      # A/B against another build of the library (e.g. libbioreason_hip_base.so)
    _lib._LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), os.environ["AP_LIB"])

if os.environ.get("AP_FWD4") is not None:
    _lib.use_debug_library().call("bra_attn_set_fwd4", int(os.environ["AP_FWD4"]))
if os.environ.get("AP_BWD4") is not None:                      # bit 0: pipelined dQ kernel, bit 1: pipelined dK / dV kernels (k_attn4b.hip)
    _lib.use_debug_library().call("bra_attn_set_bwd4", int(os.environ["AP_BWD4"]))
dev = torch.device("cuda:0")
Hq0, Hkv0 = int(os.environ.get("AP_HQ", 16)), int(os.environ.get("AP_HKV", 8))
g = torch.Generator(device="cpu").manual_seed(0)
def rnd(*s): return (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(dev)
def timeit(fn, n=int(os.environ.get("AP_N", 8))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

SHAPES = {"full": (8, 2436, 2436), "prompt": (1, 2180, 2180), "compl": (8, 256, 2436), "enc": (16, 1024, 1024), "sft": (8, 2180, 2180)}
for name in os.environ.get("AP_SHAPES", "full,prompt,compl").split(","):
    B, Sq, Sk = SHAPES[name]
    causal = name != "enc"
    Hq, Hkv, hd = (16, 16, 64) if name == "enc" else (Hq0, Hkv0, 128)
    scale = hd ** -0.5
    q, k, v, do = rnd(B, Sq, Hq, hd), rnd(B, Sk, Hkv, hd), rnd(B, Sk, Hkv, hd), rnd(B, Sq, Hq, hd)
    kmask = torch.ones((B, Sk), dtype=torch.uint8, device=dev)
    vt = ops.head_transpose(v)
    off = Sk - Sq
    pairs = Sq * off + Sq * (Sq + 1) / 2 if causal else Sq * Sk     # visible (query, key) pairs per (batch, head)
    fl = 4.0 * B * Hq * pairs * hd
    o, lse = ops.attn_fwd(q, k, vt, kmask, causal, scale)
    ms = timeit(lambda: ops.attn_fwd(q, k, vt, kmask, causal, scale))
    print(f"{name:7s} fwd ms {ms:.3f}  TFLOP/s {fl / ms / 1e9:.1f}", flush=True)
    if os.environ.get("AP_FWD_ONLY") == "1":
        if os.environ.get("AP_CHECK") == "1":
            nb = 1 if B * Sq * Sk > 8 * 256 * 2436 else B          # fp32 statement on one batch row of the big shapes
            qf, kf, vf = q[:nb].float(), k[:nb].float().repeat_interleave(Hq // Hkv, 2), v[:nb].float().repeat_interleave(Hq // Hkv, 2)
            sc = torch.einsum("bqhd,bkhd->bhqk", qf, kf) * scale
            if causal:
                vis = torch.arange(Sk, device=dev)[None, :] <= (torch.arange(Sq, device=dev)[:, None] + off)
                sc = sc.masked_fill(~vis, float("-inf"))
            ref = torch.einsum("bhqk,bkhd->bqhd", sc.softmax(-1), vf)
            lref = torch.logsumexp(sc, -1)
            print(f"{name:7s} fwd rel err vs fp32 {float((o[:nb].float() - ref).norm() / ref.norm()):.3e}  lse max abs err {float((lse[:nb] - lref).abs().max()):.2e}", flush=True)
        print(f"{name:7s} checksum o {o.float().abs().sum().item():.6e} lse {lse.double().sum().item():.9e}", flush=True)
        continue
    msb = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, kmask, causal, scale))
    print(f"{name:7s} bwd (delta + 3 transposes + dq + dkv) ms {msb:.3f}  TFLOP/s (2.5x fwd flops) {2.5 * fl / msb / 1e9:.1f}", flush=True)
    if os.environ.get("AP_CHECK") == "1" and B * Sq * Sk <= 8 * 256 * 2436:
        qf, kf, vf = q.float(), k.float().repeat_interleave(Hq // Hkv, 2), v.float().repeat_interleave(Hq // Hkv, 2)
        sc = torch.einsum("bqhd,bkhd->bhqk", qf, kf) * scale
        vis = torch.arange(Sk, device=dev)[None, :] <= (torch.arange(Sq, device=dev)[:, None] + off)
        sc = sc.masked_fill(~vis, float("-inf"))
        ref = torch.einsum("bhqk,bkhd->bqhd", sc.softmax(-1), vf)
        print(f"{name:7s} fwd rel err vs fp32 {float((o.float() - ref).norm() / ref.norm()):.3e}", flush=True)
    dq, dk, dv = ops.attn_bwd(q, k, v, o, do, lse, kmask, causal, scale)
    print(f"{name:7s} checksums o {o.float().abs().sum().item():.6e} dq {dq.float().abs().sum().item():.6e} dk {dk.float().abs().sum().item():.6e} dv {dv.float().abs().sum().item():.6e}", flush=True)
