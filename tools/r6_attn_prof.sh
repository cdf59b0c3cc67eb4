#!/bin/bash
# per-kernel durations of the attention forward / backward at the full shape: rocprofv3 kernel trace of tools/attn_probe.py
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for v in 1 0; do
  rm -rf /tmp/prof_$v
  AP_BWD4=$((v*3)) AP_FWD4=$v AP_SHAPES=${AP_SHAPES:-full} AP_N=10 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o p --output-format csv -- python $ROOT/tools/attn_probe.py > /dev/null 2>&1
  echo "== pipelined kernels = $v"
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "attn" in n or "transpose" in n or "delta" in n:
        print(f'{n[:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f}')
PY
done
