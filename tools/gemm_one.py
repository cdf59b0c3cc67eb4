"""one GEMM shape x one tile variant, a few launches: the subject of a rocprofv3 --pmc pass (tools/gemm_pmc.sh)
usage: gemm_one.py <variant> <M> <N> <K> <K2> [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops
from bioreason_amd import _lib as _bra_lib
_bra_lib.use_debug_library()          # knobs / probes / persistent step: libbioreason_hip_debug.so (include/bioreason_hip_debug.h)
from bioreason_amd._lib import get_lib
v, M, N, K, K2 = (int(x) for x in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0"); BF = torch.bfloat16
a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(N, K, device=dev).to(BF)
a2 = torch.randn(M, K2, device=dev).to(BF) if K2 else None; b2 = torch.randn(N, K2, device=dev).to(BF) if K2 else None
c = torch.empty(M, N, dtype=BF, device=dev)
get_lib().call("bra_gemm_set_variant", v)
for _ in range(iters):
    ops.gemm_nt(a, b, a2=a2, b2=b2, out=c)
torch.cuda.synchronize()
