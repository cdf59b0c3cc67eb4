cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/sm_stats
R=$GRAFT_REPO_ROOT
timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/sm_stats -o r --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-gpu-baseline-hf --no-one-stream-profile > /tmp/sm.log 2>&1
tail -1 /tmp/sm.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['phases_ms'])
"
f=$(find /tmp/sm_stats -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r5_n_slowmode_kernel_stats.csv
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f"))]
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"])/7e6:8.2f} ms/step {int(r["Calls"])//7:6d} calls/step {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:90]}')
PY
