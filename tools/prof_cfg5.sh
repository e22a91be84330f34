#!/bin/bash
# cfg5 (H = 512, bf16 operands) kernel statistics and the timeline of one step: tools/prof_cfg5.sh [tag] -> gpurun_out/<tag>_cfg5_*.md
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o trace --output-format csv -- python3 $R/bench.py --config cfg5 --steps 3 --warmup 2 --no-cpu-baseline --eager > /dev/null 2> $R/gpurun_out/${TAG}_cfg5_rocprof.err
python3 $R/tools/profile_summary.py stats $R/gpurun_out/prof_stats > $R/gpurun_out/${TAG}_cfg5_kernel_stats.md
python3 $R/tools/profile_summary.py timeline $R/gpurun_out/prof_stats > $R/gpurun_out/${TAG}_cfg5_timeline.md
rm -rf $R/gpurun_out/prof_stats
head -12 $R/gpurun_out/${TAG}_cfg5_kernel_stats.md
