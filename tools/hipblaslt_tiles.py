"""Which kernels the tuned library picks for the step's GEMM shapes (run under rocprofv3 --kernel-trace --stats: the kernel NAMES carry the
macro tile MT<m>x<n>x<k>, workgroup shape, prefetch / stream-K settings).  Intelligence for the next tile design, not a product path."""
import torch
dev = torch.device("cuda:0")
shapes = [("p_qkv", 2180, 4096, 2048), ("p_o", 2180, 2048, 2048), ("p_gate_up", 2180, 12288, 2048), ("p_down", 2180, 2048, 6144),
          ("c_qkv", 2048, 4096, 2048), ("c_o", 2048, 2048, 2048), ("c_gate_up", 2048, 12288, 2048), ("c_down", 2048, 2048, 6144),
          ("full_gate_up", 19488, 12288, 2048), ("full_down", 19488, 2048, 6144)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    torch.cuda.synchronize()
    torch.cuda.nvtx.range_push(name) if hasattr(torch.cuda, "nvtx") else None
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
    print(name, M, N, K, flush=True)
