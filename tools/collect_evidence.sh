#!/bin/bash
# Evidence of the bench configuration, collected on the GPU box:  tools/collect_evidence.sh <tag>   (e.g. r3_a)
#   profiles/<tag>_bench.json                 the bench line (default command, CPU baseline included unless NOCPU=1)
#   profiles/<tag>_bench_kernel_stats.csv     rocprofv3 --kernel-trace --stats of `bench.py --steps 1 --warmup 1` (3 steps in the trace)
#   profiles/<tag>_pmc_*.csv                  separate rocprofv3 --pmc passes on the SAME command (FETCH_SIZE | WRITE_SIZE | SQ busy counters)
#   profiles/<tag>_pmc_gemm.json              dominant-kernel traffic per launch + the kernel-source hash bench.py checks
tag=${1:-r5_x}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/evidence_$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-one-stream-profile --no-gpu-baseline-hf"
if [ "$PMC_ONLY" = "1" ]; then :      # only the counter passes (the bench line and the kernel stats of this state exist already)
elif [ "$NOCPU" = "1" ]; then timeout 300 python $R/bench.py --no-cpu-baseline --steps 5 2>/dev/null | tail -1 > $out/${tag}_bench.json
else timeout 1200 python $R/bench.py --steps 5 2>$out/${tag}_bench.err | tail -1 > $out/${tag}_bench.json; fi
[ "$PMC_ONLY" = "1" ] || { rm -rf /tmp/ev_stats; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ev_stats -o r --output-format csv -- $BENCH > /tmp/ev_stats.log 2>&1
f=$(find /tmp/ev_stats -name "*kernel_stats.csv" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- $BENCH   (MI355X, $tag; 4 GRPO steps in the trace: warm-up, timed, two instrumented)"; cat $f; } > $out/${tag}_bench_kernel_stats.csv; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"; do
  name=$(echo $pass | awk '{print $1}')
  rm -rf /tmp/ev_pmc; timeout 400 rocprofv3 --pmc $pass --kernel-trace -d /tmp/ev_pmc -o r --output-format csv -- $BENCH > /tmp/ev_pmc_$name.log 2>&1
  python $R/tools/pmc_summarize.py /tmp/ev_pmc $out/${tag}_pmc_$name.csv >> /tmp/ev_pmc_$name.log 2>&1
done
python $R/tools/pmc_to_json.py $out $tag $R 4
ls -la $out
