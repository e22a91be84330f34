#!/bin/bash
# PMC counters of the attention kernels alone (tools/att_bench.py, grouped audio + image call): one rocprofv3 --pmc pass per
# counter set, kernel-trace only.   tools/pmc_att.sh <tag>   ->   gpurun_out/<tag>_att_pmc_*.md
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
B="python3 $R/tools/att_bench.py --iters 3 --masks 0"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $R/gpurun_out/prof_pmc
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/prof_pmc -o pmc --output-format csv -- $B > /dev/null 2> $R/gpurun_out/${TAG}_att_pmc_$i.err
  python3 $R/tools/profile_summary.py pmc $R/gpurun_out/prof_pmc > $R/gpurun_out/${TAG}_att_pmc_$i.md
  rm -rf $R/gpurun_out/prof_pmc
  echo "pmc set $i done"
  i=$((i+1))
done
