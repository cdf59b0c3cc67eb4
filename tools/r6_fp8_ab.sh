#!/bin/bash
# round 6: fp8 MFMA path — kernel / model tests, per-shape probe, then the fp8 leg's step with the bf16 prompt pass (round 5's form), with the
# fp8 prompt pass, and with the fp8 reference pass as well, on ONE box
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fp8_gemm.py tests/test_fp8_rollout.py -q -x -p no:cacheprovider 2>&1 | tail -3
python tools/gemm_fp8_probe.py 2>&1 | grep -v amdgpu
C="--steps 4 --warmup 3 --no-cpu-baseline --no-gpu-baseline-hf --no-secondary --no-qwen3-4b --no-one-stream-profile --rollout-fp8"
run() {
  timeout 300 python bench.py $C $2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), round(d['ms_per_step'],2), d.get('phases_ms'), d.get('rollout_phases_ms'), 'loss', d.get('loss'), 'kl', d.get('metrics',{}).get('kl'))"
}
BRA_FP8_PREFILL=0 run "fp8 loop, bf16 prompt pass      "
run "fp8 loop + fp8 prompt pass       "
run "fp8 loop + prompt + reference    " --ref-fp8
BRA_FP8_PREFILL=0 run "fp8 loop, bf16 prompt pass (2)  "
run "fp8 loop + prompt + reference (2)" --ref-fp8
