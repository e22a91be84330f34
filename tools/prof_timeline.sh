#!/bin/bash
# kernel-trace timeline of one replayed step of `bench.py <args>` (GPU box):  tools/prof_timeline.sh TAG [bench args]
#   -> gpurun_out/TAG_kernel_stats.md, gpurun_out/TAG_timeline_graph.md
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/prof_stats
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o trace --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary "$@" > $R/gpurun_out/${TAG}_rocprof_bench.json 2> $R/gpurun_out/${TAG}_rocprof.err
python3 $R/tools/profile_summary.py stats $R/gpurun_out/prof_stats > $R/gpurun_out/${TAG}_kernel_stats.md
python3 $R/tools/profile_summary.py timeline $R/gpurun_out/prof_stats 6 > $R/gpurun_out/${TAG}_timeline_graph.md
rm -rf $R/gpurun_out/prof_stats
