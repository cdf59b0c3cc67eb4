"""Which GEMM shapes carry the time of the GRPO step: one step with HIP events around EVERY bra_gemm_bf16_nt call (M >= 256).
   python tools/gemm_shapes.py      (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import bench
from bioreason_amd import ops

args = argparse.Namespace(no_graph=False, no_shared_decode=False, no_shared_policy=os.environ.get("GS_UNSHARED") == "1", lora_dropout=0.05)
dev = torch.device("cuda:0")
dims = bench.Dims(False)
model = bench.build_model(dims, dev, 0.05)
runner, step, B = bench.make_grpo_leg(model, dims, 1, dims.c, 0, dev, args, None, 4)
step(0); step(1)
torch.cuda.synchronize()
ops.GEMM_PROFILE = ops.GemmProfile(min_m=256, dominant_only=False)
step(2)
rows = ops.gemm_profile_by_shape(ops.GEMM_PROFILE)
ops.GEMM_PROFILE = None
tot = sum(r[2] for r in rows)
print(f"total {tot:.1f} ms in {sum(r[1] for r in rows)} launches")
for (M, N, K, K2, res), n, ms, tf in rows[:40]:
    t256, t128 = -(-M // 256) * -(-N // 256), -(-M // 256) * -(-N // 128)
    print(f"M {M:6d} N {N:6d} K {K:6d} K2 {K2:3d} res {int(res)}  x{n:4d}  {ms:7.2f} ms  {ms / n * 1e3:7.1f} us each  {tf:7.1f} TF/s   tiles 256x256 {t256:4d} / 256x128 {t128:4d}")
