#!/bin/bash
# HBM-side traffic of the attention kernels alone (tools/att_bench.py): FETCH_SIZE and WRITE_SIZE passes only.
#   tools/pmc_att_fetch.sh <tag>   ->   gpurun_out/<tag>_att_fetch.md, <tag>_att_write.md
set -e
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
B="python3 $R/tools/att_bench.py --iters 3 --masks 0"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/prof_pmc
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/prof_pmc -o pmc --output-format csv -- $B > /dev/null 2> $R/gpurun_out/${TAG}_att_$C.err
  python3 $R/tools/profile_summary.py pmc $R/gpurun_out/prof_pmc > $R/gpurun_out/${TAG}_att_$C.md
  rm -rf $R/gpurun_out/prof_pmc
done
