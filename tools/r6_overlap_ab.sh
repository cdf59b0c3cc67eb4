#!/bin/bash
# same-box A/B of the step's stream-overlap switches at the round-6 kernels
export TMPDIR=/tmp
C="--steps 6 --warmup 3 --no-cpu-baseline --no-gpu-baseline-hf --no-secondary --no-qwen3-4b --no-one-stream-profile"
for rep in 1 2; do
for f in "" "--overlap-ref-chains" "--no-overlap-ref" "--no-overlap-chains"; do
  timeout 300 python bench.py $C $f 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('${f:-default}', round(d['value'],3), round(d['ms_per_step'],2))"
done
done
