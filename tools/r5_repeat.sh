cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do
  W=1; [ $i -ge 4 ] && W=3
  timeout 120 python bench.py --steps 5 --warmup $W --no-cpu-baseline --no-secondary --no-gpu-baseline-hf --no-one-stream-profile 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('warmup', d['warmup'], 'ms/step', round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['phases_ms'].items()}, d['rollout_issue'])
"
done
nproc; cat /proc/cpuinfo | grep "model name" | sort | uniq -c | head -2; uptime
