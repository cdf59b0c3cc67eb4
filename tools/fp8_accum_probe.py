"""how exact is the accumulation inside v_mfma_scale_f32_16x16x128_f8f6f4?  fp32 outputs of bra_gemm_fp8_nt against fp64 sums of the same
decoded operands, error measured against the sum of |products| (what a fixed-point alignment to the largest term would lose)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioreason_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for K in (128, 256, 2048):
    for mask, tag in ((0xbf, "|v| <= 1.875"), (0xff, "all codes")):
        M = N = 256
        a = torch.randint(0, 256, (M, K), generator=g, dtype=torch.int32); b = torch.randint(0, 256, (N, K), generator=g, dtype=torch.int32)
        a = torch.where((a & 0x7f) == 0x7f, a & 0x80, a) & mask; b = torch.where((b & 0x7f) == 0x7f, b & 0x80, b) & mask
        a8, b8 = a.to(torch.uint8).to(dev), b.to(torch.uint8).to(dev)
        one_m, one_n = torch.ones(M, device=dev), torch.ones(N, device=dev)
        c = ops.gemm_fp8_nt(a8, one_m, b8, one_n, out_f32=True).double().cpu()
        A, B = a8.cpu().view(torch.float8_e4m3fn).double(), b8.cpu().view(torch.float8_e4m3fn).double()
        want = A @ B.T
        sabs = A.abs() @ B.abs().T
        err = (c - want).abs()
        # the largest |product| per output element
        print(f"K {K:5d} {tag:14s}: max err / sum|products| {float((err / sabs).max()):.3e}   mean {float((err / sabs).mean()):.3e}   "
              f"max err / |result| {float((err / want.abs().clamp_min(1e-30)).max()):.3e}   exact elements {float((err == 0).double().mean()):.3f}")
