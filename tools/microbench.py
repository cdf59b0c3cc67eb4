"""Kernel micro-benchmarks at the production shapes (NT-500M + Qwen3-1.7B, cfg-2/3 of SURVEY §8d)."""
import json
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    out = {}
    T = 8 * 2436
    shapes = [("qkv", T, 4096, 2048, 128), ("o", T, 2048, 2048, 64), ("gate_up", T, 12288, 2048, 64), ("down", T, 2048, 6144, 64),
              ("enc_qkv", 16384, 3072, 1024, 0), ("enc_ffn_up", 16384, 8192, 1024, 0), ("enc_ffn_down", 16384, 1024, 4096, 0),
              ("lm_head", 2048, 151936, 2048, 0), ("decode_qkv", 8, 4096, 2048, 0), ("decode_gate_up", 8, 12288, 2048, 0),
              ("sq4096", 4096, 4096, 4096, 0), ("sq8192", 8192, 8192, 8192, 0)]
    for name, M, N, K, K2 in shapes:
        a = torch.randn(M, K, device=dev).to(BF)
        b = torch.randn(N, K, device=dev).to(BF)
        a2 = torch.randn(M, K2, device=dev).to(BF) if K2 else None
        b2 = torch.randn(N, K2, device=dev).to(BF) if K2 else None
        c = torch.empty(M, N, dtype=BF, device=dev)
        ms = timeit(lambda: ops.gemm_nt(a, b, a2=a2, b2=b2, out=c))
        ref = a.float() @ b.float().T
        if K2:
            ref += a2.float() @ b2.float().T
        err = ((c.float() - ref).norm() / ref.norm()).item()
        tf = 2.0 * M * N * (K + K2) / ms / 1e9
        ms_t = timeit(lambda: torch.matmul(a, b.T))
        out[f"gemm_{name}"] = {"M": M, "N": N, "K": K, "K2": K2, "ms": ms, "TF": tf, "rel_err": err, "torch_ms": ms_t,
                               "torch_TF": 2.0 * M * N * K / ms_t / 1e9}
        print(name, out[f"gemm_{name}"], flush=True)
        del a, b, c, ref
    # attention
    for name, B, Hq, Hkv, S, hd, causal in [("qwen_causal", 8, 16, 8, 2436, 128, True), ("esm_bidir", 16, 16, 16, 1024, 64, False)]:
        q = torch.randn(B, S, Hq, hd, device=dev).to(BF)
        k = torch.randn(B, S, Hkv, hd, device=dev).to(BF)
        v = torch.randn(B, S, Hkv, hd, device=dev).to(BF)
        vt = ops.head_transpose(v)
        scale = hd ** -0.5
        ms = timeit(lambda: ops.attn_fwd(q, k, vt, None, causal, scale))
        fl = 4.0 * B * Hq * S * S * hd * (0.5 if causal else 1.0)
        o, lse = ops.attn_fwd(q, k, vt, None, causal, scale)
        qq, kk, vv = q.transpose(1, 2), k.transpose(1, 2).repeat_interleave(Hq // Hkv, 1), v.transpose(1, 2).repeat_interleave(Hq // Hkv, 1)
        ro = torch.nn.functional.scaled_dot_product_attention(qq, kk, vv, is_causal=causal).transpose(1, 2)
        err = ((o.float() - ro.float()).norm() / ro.float().norm()).item()
        ms_t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qq, kk, vv, is_causal=causal))
        dout = torch.randn_like(o)
        ms_b = timeit(lambda: ops.attn_bwd(q, k, v, o, dout, lse, None, causal, scale), iters=5, warm=2)
        out[f"attn_{name}"] = {"fwd_ms": ms, "fwd_TF": fl / ms / 1e9, "rel_err": err, "sdpa_ms": ms_t, "sdpa_TF": fl / ms_t / 1e9,
                               "bwd_ms": ms_b, "bwd_TF": 2.5 * fl / ms_b / 1e9}
        print(name, out[f"attn_{name}"], flush=True)
    # HBM-bound
    x = torch.randn(T, 2048, device=dev).to(BF)
    w = torch.ones(2048, device=dev).to(BF)
    ms = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-6))
    out["rmsnorm"] = {"ms": ms, "GBps": 2 * x.numel() * 2 / ms / 1e6}
    gu = torch.randn(T, 12288, device=dev).to(BF)
    ms = timeit(lambda: ops.swiglu_fwd(gu))
    out["swiglu"] = {"ms": ms, "GBps": 3 * T * 6144 * 2 / ms / 1e6}
    print(out["rmsnorm"], out["swiglu"])
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/microbench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
