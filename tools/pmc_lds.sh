#!/bin/bash
# LDS counters of the step's kernels (GPU box): bank-conflict cycles against LDS-active cycles per kernel
#   tools/pmc_lds.sh TAG -> gpurun_out/TAG_pmc_LDS.md
set -e
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_pmc
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/prof_pmc -o pmc --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2> $R/gpurun_out/${TAG}_pmc_LDS.err
python3 $R/tools/profile_summary.py pmc $R/gpurun_out/prof_pmc nowrite > $R/gpurun_out/${TAG}_pmc_LDS.md
rm -rf $R/gpurun_out/prof_pmc
