#!/bin/bash
# SQ-side counters of the GEMM variants on two shapes (run on the GPU box): tools/gemm_pmc.sh <outdir>
out=${1:-gpurun_out/gemm_pmc}; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for v in 5 6; do for shape in "8192 8192 8192 0" "19488 12288 2048 64"; do
  tag=v${v}_$(echo $shape | tr ' ' '_')
  timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
     --kernel-trace -d /tmp/pmc_$tag -o r --output-format csv -- python $R/tools/gemm_one.py $v $shape 4 > /tmp/pmc_$tag.log 2>&1
  python $R/tools/pmc_summarize.py /tmp/pmc_$tag $R/$out/$tag.csv >> /tmp/pmc_$tag.log 2>&1
  grep -i "gemm_" $R/$out/$tag.csv | head -12
done; done
