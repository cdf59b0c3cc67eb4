R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r4_j; mkdir -p $O
run() { name=$1; shift; timeout 240 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-one-stream-profile "$@" > $O/$name.out 2> $O/$name.err; rc=$?; echo "$name rc=$rc $(tail -1 $O/$name.out | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('loss'), d.get('metrics',{}).get('completion_length'))
except Exception as e: print('no line', e)")"; [ $rc -ne 0 ] && tail -5 $O/$name.err; }
run c800 --completion-len 800
run c1 --completion-len 1
run c65 --completion-len 65
run eos_early --eos-uniform 1 8
run ppg5 --prompts-per-gpu 5 --completion-len 64
run nograph_noshare --no-shared-decode --completion-len 64
run noshared_policy_ppg2 --no-shared-policy --prompts-per-gpu 2 --completion-len 64
run drop0 --lora-dropout 0 --completion-len 64
run sft --mode sft
