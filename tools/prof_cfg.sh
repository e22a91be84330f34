#!/bin/bash
# kernel-trace stats + one-step timeline of a secondary configuration: tools/prof_cfg.sh cfg5 [tag [extra bench flags]]
set -e
CFG=${1:-cfg5}; shift || true
TAG=${1:-r05}; shift || true
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$CFG
rm -rf $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- python3 $R/bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline --no-secondary "$@" > $R/gpurun_out/${TAG}_${CFG}_rocprof_bench.json 2> $R/gpurun_out/${TAG}_${CFG}_rocprof.err
python3 $R/tools/profile_summary.py stats $OUT > $R/gpurun_out/${TAG}_${CFG}_kernel_stats.md
python3 $R/tools/profile_summary.py timeline $OUT 4 > $R/gpurun_out/${TAG}_${CFG}_timeline_graph.md || true
rm -rf $OUT
