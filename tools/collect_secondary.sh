#!/bin/bash
# Secondary bench lines of SURVEY §8d on the GPU box:  tools/collect_secondary.sh <tag>
#   profiles/<tag>_bench_sft.json    python bench.py --mode sft            (BASELINE config 2)
#   profiles/<tag>_bench_strag.json  python bench.py --eos-uniform 64 256  (rollout lengths U[64, 256])
#   profiles/<tag>_bench_ppg2.json   python bench.py --prompts-per-gpu 2   (16 rows per GPU)
tag=${1:-r2_x}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/evidence_$tag; mkdir -p $out
cd $R
timeout 200 python bench.py --mode sft --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${tag}_bench_sft.json
timeout 200 python bench.py --eos-uniform 64 256 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${tag}_bench_strag.json
timeout 250 python bench.py --prompts-per-gpu 2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${tag}_bench_ppg2.json
for f in sft strag ppg2; do python - <<PY
import json
try:
    d = json.loads(open("$out/${tag}_bench_$f.json").read())
    print("$f", round(d["value"], 2), d["unit"], round(d["ms_per_step"], 1), "ms/step", d.get("phases_ms"))
except Exception as e:
    print("$f failed", e)
PY
done
