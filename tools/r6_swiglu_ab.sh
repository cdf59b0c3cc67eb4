#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels.py -q -x -p no:cacheprovider -m gpu -k "gemm" 2>&1 | tail -2
python - <<'PY' 2>&1 | grep -v amdgpu
import torch, sys, os
sys.path.insert(0, os.getcwd())
from bioreason_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters
for name, M, F, K in (("enc_ffn", 16384, 4096, 1024), ("p_gate_up", 2180, 6144, 2048), ("c_gate_up", 2048, 6144, 2048), ("s_gate_up", 17440, 6144, 2048)):
    x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(2 * F, K, device=dev) * 0.05).to(BF)
    t2 = timeit(lambda: ops.swiglu_fwd(ops.gemm_nt(x, w)))
    t1 = timeit(lambda: ops.gemm_swiglu(x, w))
    same = torch.equal(ops.swiglu_fwd(ops.gemm_nt(x, w)), ops.gemm_swiglu(x, w))
    print(f"{name:10s} M {M:6d} F {F:5d} K {K:5d}: gemm + swiglu {t2 * 1e3:7.1f} us, fused {t1 * 1e3:7.1f} us  x{t2 / t1:5.3f}  identical {same}")
PY
C="--steps 6 --warmup 3 --no-cpu-baseline --no-gpu-baseline-hf --no-secondary --no-qwen3-4b --no-one-stream-profile"
for rep in 1 2; do
for f in 1 0; do
  BRA_FUSE_SWIGLU=$f timeout 300 python bench.py $C 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('grpo fuse=$f', round(d['value'],3), round(d['ms_per_step'],2), d.get('phases_ms'))"
  BRA_FUSE_SWIGLU=$f timeout 300 python bench.py $C --mode sft 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sft  fuse=$f', round(d['value'],3), round(d['ms_per_step'],2), d.get('phases_ms'))"
done
done
