#!/bin/bash
# round 6: the 4-wave attention forward (k_attn4.hip) against the 8-wave kernel on one box; kernel tests first
export TMPDIR=/tmp
OUT=gpurun_out/r6_attn_ab.txt
: > $OUT
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "attn" -x -p no:cacheprovider 2>&1 | tail -5 >> $OUT
for v in 1 0 1 0; do
  echo "== AP_FWD4=$v" >> $OUT
  AP_FWD4=$v AP_FWD_ONLY=1 AP_CHECK=1 AP_SHAPES=${AP_SHAPES:-full,sft,prompt,compl,enc} AP_N=20 timeout 300 python tools/attn_probe.py >> $OUT 2>&1
done
cat $OUT
