#!/bin/bash
# round 6: sampler kernel tests on the device, then the token step with the one-launch / two-launch tile-maxima sampler on ONE box,
# then the SFT leg (ring gate 60 %)
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -x -p no:cacheprovider -k "sampl or tile or gemm" 2>&1 | tail -3
C="--steps 4 --warmup 3 --no-cpu-baseline --no-gpu-baseline-hf --no-secondary --no-qwen3-4b --no-one-stream-profile"
for rep in 1 2; do
  for f in "" "--two-launch-sampler"; do
    timeout 300 python bench.py $C $f 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sampler', '$f' or 'one-launch', d['value'], d['ms_per_step'], d['roofline']['ms_per_token_step'])"
  done
done
timeout 300 python bench.py $C --mode sft 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sft', d['value'], d['ms_per_step'], d.get('phases_ms'))"
