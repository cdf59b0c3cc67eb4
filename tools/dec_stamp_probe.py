"""intra-kernel wall-clock stamps of bra_dec_gemm2 (100 MHz s_memrealtime): where a ~9 us decode projection spends its time.
Two probe buffers alternate over back-to-back launches in one stream; printed for the last two launches:
  entry -> requests issued -> products ready -> barrier passed -> epilogue issued   (first and last workgroup)
  and the gap  previous launch's epilogue-issued -> next launch's entry."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops
from bioreason_amd import _lib as _bra_lib
_bra_lib.use_debug_library()          # knobs / probes / persistent step: libbioreason_hip_debug.so (include/bioreason_hip_debug.h)
from bioreason_amd._lib import get_lib, current_stream
dev = torch.device("cuda:0"); BF = torch.bfloat16
M = 8
CASES = {"qkv": (4096, 2048, 0, 1, 0), "o": (2048, 2048, 0, 0, 1), "gate_up": (12288, 2048, 1, 1, 0), "down": (2048, 6144, 0, 0, 1)}
lib = get_lib()
for name, (N, K, act, norm, res) in CASES.items():
    for tag in ("hbm", "packed"):
        Kx = K
        ncopy = max(2, int(600e6 // (N * Kx * 2)) + 1)
        Ws = [torch.randn(N, Kx, device=dev).to(BF) for _ in range(min(ncopy, 40))]
        x = torch.randn(M, Kx, device=dev).to(BF); nw = torch.ones(Kx, device=dev).to(BF)
        ss = ops.row_sumsq(x, 256) if norm else None
        r = torch.randn(M, N, device=dev).to(BF) if res else None
        out = torch.empty((M, N // 2 if act else N), dtype=BF, device=dev)
        nss_out = (N // 8 + 32) // 32 * 32
        ss_out = torch.zeros((8, nss_out), device=dev) if res else None
        probes = [torch.zeros(16, dtype=torch.int64, device=dev) for _ in range(2)]
        st = current_stream(x)
        for i in range(100):
            lib.call("bra_dec_gemm2_probe", x, Kx, ss, 256 if norm else 0, nw if norm else None, 1e-6, Ws[i % len(Ws)], Kx,
                     r, N if res else 0, out, out.shape[1], ss_out, nss_out if res else 0, M, N, Kx, act, 0, int(tag == 'packed'), probes[i & 1], st)
        torch.cuda.synchronize()
        a, b = probes[0].cpu().tolist(), probes[1].cpu().tolist()          # launch 98 -> a, launch 99 -> b
        def seg(p, o): return [round((p[o + k + 1] - p[o + k]) * 0.01, 2) for k in range(4)]
        print(f"{name:8s} {tag:4s} first WG {seg(b, 0)} total {(b[4] - b[0]) * 0.01:.2f} us | last WG {seg(b, 8)} total {(b[12] - b[8]) * 0.01:.2f} us | "
              f"entry skew last-first {(b[8] - b[0]) * 0.01:.2f} us | prev epilogue(last WG) -> entry(first WG) {(b[0] - a[12]) * 0.01:.2f} us | "
              f"launch period {(b[0] - a[0]) * 0.01:.2f} us", flush=True)
