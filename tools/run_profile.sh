#!/bin/bash
# Kernel-trace profile of the default bench on the GPU box: tools/run_profile.sh <tag>  ->  gpurun_out/<tag>_kernel_stats.md
set -e
TAG=${1:-snap}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench.err
python3 $GRAFT_REPO_ROOT/tools/profile_summary.py stats $OUT > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.md
rm -rf $OUT
