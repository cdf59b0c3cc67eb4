#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
C="--steps 1 --warmup 1 --no-cpu-baseline --no-gpu-baseline-hf --no-secondary --no-qwen3-4b --no-one-stream-profile"
for f in "" "--two-launch-sampler"; do
  rm -rf /tmp/prof_s
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o p --output-format csv -- python $ROOT/bench.py $C $f > /dev/null 2>&1
  echo "== ${f:-one-launch}"
  python - "$(find /tmp/prof_s -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "sample" in n or "topk" in n or "dec_gemm2_kernel<0, 2, 0, 1" in n:
        print(f'{n[:80]:80s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:8.2f}')
PY
done
