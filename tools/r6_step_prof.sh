#!/bin/bash
# round 6: rocprofv3 kernel traces of three bench.py steps on one box — the SFT leg (cfg-2), the headline GRPO step (cfg-3) and the
# reference-semantics step (full-row policy pass) — each summarised by kernel family (tools/kernel_breakdown.py).
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r6_j}
COMMON="--steps 3 --warmup 3 --no-cpu-baseline --no-gpu-baseline-hf --no-secondary --no-qwen3-4b --no-one-stream-profile"
cd /tmp
for leg in sft grpo unshared; do
  case $leg in
    sft) FL="--mode sft";;
    grpo) FL="";;
    unshared) FL="--no-shared-policy";;
  esac
  rm -rf /tmp/prof_$leg
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$leg -o p --output-format csv -- python $ROOT/bench.py $COMMON $FL > $ROOT/gpurun_out/${TAG}_${leg}_bench.json 2> $ROOT/gpurun_out/${TAG}_${leg}_bench.err
  f=$(find /tmp/prof_$leg -name "*kernel_stats.csv" | head -1)
  cp "$f" $ROOT/gpurun_out/${TAG}_${leg}_kernel_stats.csv
  # steps traced: 3 warm-up + 3 timed + 2 instrumented
  python $ROOT/tools/kernel_breakdown.py $ROOT/gpurun_out/${TAG}_${leg}_kernel_stats.csv 8 45 > $ROOT/gpurun_out/${TAG}_${leg}_breakdown.md
  head -30 $ROOT/gpurun_out/${TAG}_${leg}_breakdown.md
  python - <<PY
import json
l=[x for x in open("$ROOT/gpurun_out/${TAG}_${leg}_bench.json") if x.startswith("{")]
d=json.loads(l[-1]); print("$leg", d["value"], d["ms_per_step"], d.get("phases_ms"))
PY
done
