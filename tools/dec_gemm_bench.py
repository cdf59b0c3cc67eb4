import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops
from bioreason_amd._lib import get_lib, current_stream
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3
lib = get_lib()
M = 8
for name, N, K, act in [("qkv", 4096, 2048, 0), ("o", 2048, 2048, 0), ("gate_up", 12288, 2048, 1), ("down", 2048, 6144, 0), ("lm_head", 151936, 2048, 0)]:
    # rotate over several weight copies so that the 256 MB infinity cache does not serve the weights
    ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
    Ws = [torch.randn(N, K, device=dev).to(BF) for _ in range(min(ncopy, 24))]
    x = torch.randn(M, K, device=dev).to(BF); nw = torch.ones(K, device=dev).to(BF)
    out = torch.empty(M, N // 2 if act else N, dtype=BF, device=dev)
    st = current_stream(x)
    i = [0]
    def run(norm):
        W = Ws[i[0] % len(Ws)]; i[0] += 1
        lib.call("bra_dec_gemm", x, K, nw if norm else None, 1e-6, W, K, None, 0, out, out.shape[1], M, N, K, act, 0, st)
    t_norm = timeit(lambda: run(True)); t_plain = timeit(lambda: run(False))
    def run_sk():
        W = Ws[i[0] % len(Ws)]; i[0] += 1
        ops.gemm_nt(x, W, out=out if not act else None)
    t_sk = timeit(run_sk) if not act else float("nan")
    mb = N * K * 2 / 1e6
    print(f"{name}: {mb:.0f} MB  norm {t_norm:.1f} us ({mb/t_norm:.2f} TB/s)  plain {t_plain:.1f} us ({mb/t_plain:.2f} TB/s)  skinny {t_sk:.1f} us", flush=True)
