"""In-launch grid barrier + hand-off protocol of bra_gridsync.h, measured and word-checked on the device:
   python tools/gridbar_probe.py            (GPU box)
us per iteration of {publish 128 B per workgroup, grid barrier, read all slots} on 256 workgroups x 512 threads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioreason_amd._lib import get_lib, current_stream
from bioreason_amd import _lib as _bra_lib
_bra_lib.use_debug_library()          # knobs / probes / persistent step: libbioreason_hip_debug.so (include/bioreason_hip_debug.h)

dev = torch.device("cuda:0")
lib = get_lib()
nwg = torch.cuda.get_device_properties(0).multi_processor_count
sync = torch.zeros(2048, dtype=torch.uint8, device=dev)          # >= bra_gridsync_bytes() = 1152
buf = torch.zeros(2 * nwg * 32, dtype=torch.int32, device=dev)
errs = torch.zeros(4, dtype=torch.int32, device=dev)
wts = torch.randint(0, 2 ** 31 - 1, (1 << 28,), dtype=torch.int32, device=dev)      # 1 GiB: past the 256 MiB Infinity Cache
print("CUs", nwg)
for mode, wch, name in [(0, 0, "barrier only"), (1, 0, "sc1 stores / sc1 loads"), (4, 0, "sc1, scalar-path poll"),
                        (3, 2, "sc1 + 4 MB weight stream across the barrier"), (3, 4, "sc1 + 8 MB stream"), (3, 8, "sc1 + 16 MB stream"),
                        (5, 2, "scalar poll + 4 MB stream"), (5, 4, "scalar poll + 8 MB stream"), (5, 8, "scalar poll + 16 MB stream")]:
    for iters in (200, 2000):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        lib.call("bra_gridbar_probe", sync, buf, errs, wts, wts.numel() * 4, nwg, iters, mode, wch, 20000, current_stream(buf))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        er = errs.tolist()
        tmo = int(sync.view(torch.int32)[8 * 16 + 16 + 8 * 16].item())
        extra = ""
        if mode in (3, 5):
            extra = "  stream %.2f TB/s" % (nwg * 512 * wch * 16 * iters / (ms * 1e-3) / 1e12)
        print(f"mode {mode} [{name}] iters {iters}: {ms * 1e3 / iters:.2f} us / iteration, bad words {er[0]}, barriers {er[1]}, timeout word {tmo}{extra}", flush=True)
