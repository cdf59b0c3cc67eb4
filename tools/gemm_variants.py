import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops
from bioreason_amd import _lib as _bra_lib
_bra_lib.use_debug_library()          # knobs / probes / persistent step: libbioreason_hip_debug.so (include/bioreason_hip_debug.h)
from bioreason_amd._lib import get_lib
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters
T = 8 * 2436
P1 = 2180
if os.environ.get("GV_SMALLM") == "1":
    # the GEMMs of the shared-prompt GRPO step (round 3+): prompt chain M = 2180, completion chain M = 8 x 256 = 2048; forward
    # projections and the input-gradient GEMMs (K = the forward's N); encoder rows M = 2052.  Variants: 0 = 128x128 register-staged,
    # 5 / 9 / 10 = LDS-DMA kernel at 256 / 192 / 128-row tiles, 7 = 256x256 ring, 11 - 14 = four waves, macro tile 160x256 / 128x256 / 160x128 / 128x128
    shapes = []
    for tag, M in (("p", 2180), ("c", 2048)):
        shapes += [(tag + "_qkv", M, 4096, 2048, 128), (tag + "_o", M, 2048, 2048, 64), (tag + "_gate_up", M, 12288, 2048, 64),
                   (tag + "_down", M, 2048, 6144, 64), (tag + "_d_gate_up", M, 2048, 12288, 128), (tag + "_d_down", M, 6144, 2048, 64),
                   (tag + "_d_qkv", M, 2048, 4096, 64), (tag + "_d_o", M, 2048, 2048, 64)]
    shapes += [("e_qkv", 2052, 3072, 1024, 0), ("e_o", 2052, 1024, 1024, 0), ("e_ffn_up", 2052, 8192, 1024, 0), ("e_ffn_dn", 2052, 1024, 4096, 0)]
elif os.environ.get("GV_SFT") == "1":
    # SFT (cfg-2) rows: M = 8 x 2180 = 17 440 — 68.1 tile-rows of 256: the N = 2048 projections are 552 ring tiles = 2.16 rounds
    M = 17440
    shapes = [("s_qkv", M, 4096, 2048, 128), ("s_o", M, 2048, 2048, 64), ("s_gate_up", M, 12288, 2048, 64), ("s_down", M, 2048, 6144, 64),
              ("s_d_gate_up", M, 2048, 12288, 128), ("s_d_down", M, 6144, 2048, 64), ("s_d_qkv", M, 2048, 4096, 64), ("s_d_o", M, 2048, 2048, 64)]
elif os.environ.get("GV_PREFILL") == "1":
    # one-prompt prefill / encoder shapes (M ~ 2 k): fewer than 192 256x128 tiles
    shapes = [("p_qkv", P1, 4096, 2048, 128), ("p_o", P1, 2048, 2048, 64), ("p_gate_up", P1, 12288, 2048, 64), ("p_down", P1, 2048, 6144, 64),
              ("e_qkv", 2052, 3072, 1024, 0), ("e_o", 2052, 1024, 1024, 0), ("e_ffn_up", 2052, 8192, 1024, 0), ("e_ffn_dn", 2052, 1024, 4096, 0),
              ("lora_A", T, 64, 2048, 0), ("lora_A_gu_bwd", T, 64, 12288, 0)]
else:
  shapes = [("qkv", T, 4096, 2048, 128), ("o", T, 2048, 2048, 64), ("gate_up", T, 12288, 2048, 64), ("down", T, 2048, 6144, 64),
          ("d_gate_up", T, 2048, 12288, 128), ("d_down", T, 6144, 2048, 64), ("d_qkv", T, 2048, 4096, 64),
          ("enc_qkv", 16384, 3072, 1024, 0), ("enc_ffn_dn", 16384, 1024, 4096, 0), ("lm_head", 2048, 151936, 2048, 0), ("sq8192", 8192, 8192, 8192, 0),
          ("sft_lm_head", 17440, 151936, 2048, 0)]
for name, M, N, K, K2 in shapes:
    a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(N, K, device=dev).to(BF)
    a2 = torch.randn(M, K2, device=dev).to(BF) if K2 else None; b2 = torch.randn(N, K2, device=dev).to(BF) if K2 else None
    c = torch.empty(M, N, dtype=BF, device=dev)
    # GV_RES=1: with a residual operand, as the o / down projections and every input-gradient GEMM of the model run (the tile
    # epilogue then reads M x N more elements; round 2's last change batches those loads per wave)
    rs = torch.randn(M, N, device=dev).to(BF) if os.environ.get("GV_RES") == "1" and M * N <= (1 << 29) else None
    res = {}
    if os.environ.get("GV_CHECK") == "1":            # variant 6 against variant 5 on the same operands (both fp32-accumulated)
        get_lib().call("bra_gemm_set_variant", 5); c5 = ops.gemm_nt(a, b, a2=a2, b2=b2).float()
        for vv in ((9, 10, 0, 11, 12, 13, 14) if os.environ.get("GV_SMALLM") == "1" else (7,)):
            get_lib().call("bra_gemm_set_variant", vv); c6 = ops.gemm_nt(a, b, a2=a2, b2=b2).float()
            print(name, "variant", vv, "max |v - v5| / max|v5| =", float((c6 - c5).abs().max() / c5.abs().max()), "mismatching elements", int((c6 != c5).sum()), flush=True)
        del c5, c6
    if os.environ.get("GV_SFT") == "1":
        # current choice (auto at the 75 % ring fill gate), the ring gate at 70 % (ring + row split for 2.16 rounds), pinned tiles
        for tag, v, fill in (("auto75", -1, 75), ("auto70", -1, 70), ("auto60", -1, 60), ("glds256", 5, 75), ("glds192", 9, 75), ("ring", 7, 75)):
            get_lib().call("bra_gemm_set_variant", v); get_lib().call("bra_gemm_set_ring_fill", fill)
            ms = timeit(lambda: ops.gemm_nt(a, b, a2=a2, b2=b2, out=c, res=rs))
            res[tag] = round(2.0 * M * N * (K + K2) / ms / 1e9)
        get_lib().call("bra_gemm_set_variant", -1); get_lib().call("bra_gemm_set_ring_fill", 60)
    vs = (5, 9, 10, 7, 11, 12, 13, 14, -2, -1) if os.environ.get('GV_SMALLM') == '1' else ((5, 6, 7) if os.environ.get('GV_FAST') == '1' else (0, 4, 5, 6, 7))
    for v in (() if os.environ.get("GV_SFT") == "1" else vs):
        get_lib().call("bra_gemm_set_variant", v)
        ms = timeit(lambda: ops.gemm_nt(a, b, a2=a2, b2=b2, out=c, res=rs))
        res[v] = round(2.0 * M * N * (K + K2) / ms / 1e9)
    get_lib().call("bra_gemm_set_variant", -1)
    ms = timeit(lambda: torch.matmul(a, b.T))
    print(name, "TF by variant", res, "torch", round(2.0 * M * N * K / ms / 1e9), flush=True)
