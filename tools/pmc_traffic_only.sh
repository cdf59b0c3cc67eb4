#!/bin/bash
# The two PMC passes bench.py's `traffic` fields need (FETCH_SIZE, WRITE_SIZE; separate rocprofv3 runs, --kernel-trace only), for a tag
# whose bench line exists already (profiles/<tag>_bench.json):  tools/pmc_traffic_only.sh <tag>
tag=${1:-r4_q}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/evidence_$tag; mkdir -p $out
[ -f $out/${tag}_bench.json ] || cp $R/profiles/${tag}_bench.json $out/
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-one-stream-profile --no-gpu-baseline-hf"
for name in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ev_pmc; timeout ${T:-80} rocprofv3 --pmc $name --kernel-trace -d /tmp/ev_pmc -o r --output-format csv -- $BENCH > /tmp/ev_pmc_$name.log 2>&1
  python $R/tools/pmc_summarize.py /tmp/ev_pmc $out/${tag}_pmc_$name.csv >> /tmp/ev_pmc_$name.log 2>&1
done
python $R/tools/pmc_to_json.py $out $tag $R 4 | head -30
