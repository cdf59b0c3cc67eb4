"""Dependent kernel chain: ordinary launches on one stream vs two streams + device-side completion counters (k_persist.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioreason_amd._lib import get_lib
from bioreason_amd import _lib as _bra_lib
_bra_lib.use_debug_library()          # knobs / probes / persistent step: libbioreason_hip_debug.so (include/bioreason_hip_debug.h)

dev = torch.device("cuda:0")
lib = get_lib()
nwg = 256
n = 168
done = torch.zeros(n * 16, dtype=torch.int32, device=dev)
buf = torch.zeros(2 * nwg * 32, dtype=torch.int32, device=dev)
errs = torch.zeros(4, dtype=torch.int32, device=dev)
wts = torch.randint(0, 2 ** 31 - 1, (1 << 28,), dtype=torch.int32, device=dev)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for wch in (4, 8, 12):
    for chained in (0, 1, 0, 1):
        best = 1e9
        for rep in range(3):
            done.zero_(); errs.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(sa)
            sb.wait_stream(sa)
            lib.call("bra_chain_probe", done, buf, errs, wts, wts.numel() * 4, nwg, n, chained, wch, 20000, sa.cuda_stream, sb.cuda_stream)
            sa.wait_stream(sb)
            e1.record(sa)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        er = errs.tolist()
        mb = nwg * 512 * wch * 16 / 1e6
        print(f"wchunks {wch} ({mb:.1f} MB per kernel) chained {chained}: {best * 1e3 / n:.2f} us per kernel, {mb * 1e6 * n / (best * 1e-3) / 1e12:.2f} TB/s, bad words {er[0]}, timeout at {er[3]}", flush=True)
