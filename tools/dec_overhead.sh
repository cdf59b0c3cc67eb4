#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for n in qkv o gate_up down; do for t in hbm cached k64; do
  rm -rf /tmp/do_$n$t; timeout 60 rocprofv3 --kernel-trace --stats -d /tmp/do_$n$t -o r --output-format csv -- python $R/tools/dec_overhead_probe.py $n $t > /tmp/do.log 2>&1
  f=$(find /tmp/do_$n$t -name "*kernel_stats.csv" | head -1)
  echo "$n $t: $(grep dec_gemm2 $f | head -1 | awk -F, '{print $(NF-6), "calls", $(NF-5), "avg_ns", $(NF-3)}' | tr -d '"')"
done; done
