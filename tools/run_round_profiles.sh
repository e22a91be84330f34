#!/bin/bash
# Round-end measurement set on the GPU box (one call): kernel-trace stats, three separate PMC passes (FETCH_SIZE,
# WRITE_SIZE, SQ counters; never combined with trace domains other than kernel-trace), then the default bench.
#   tools/run_round_profiles.sh r01   ->  gpurun_out/r01_*.md|json  (copy the ones to keep into profiles/)
set -e
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
rm -rf $R/gpurun_out/prof_stats
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o trace --output-format csv -- $B > $R/gpurun_out/${TAG}_rocprof_bench.json 2> $R/gpurun_out/${TAG}_rocprof.err
python3 $R/tools/profile_summary.py stats $R/gpurun_out/prof_stats > $R/gpurun_out/${TAG}_kernel_stats.md
python3 $R/tools/profile_summary.py timeline $R/gpurun_out/prof_stats > $R/gpurun_out/${TAG}_timeline.md
# (the default picks one of the eagerly issued, instrumented steps at the end of the run; step 6 is a REPLAYED graph step)
python3 $R/tools/profile_summary.py timeline $R/gpurun_out/prof_stats 6 > $R/gpurun_out/${TAG}_timeline_graph.md
rm -rf $R/gpurun_out/prof_stats
echo "stats done"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/prof_pmc
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/prof_pmc -o pmc --output-format csv -- $B > /dev/null 2> $R/gpurun_out/${TAG}_pmc_$C.err
  python3 $R/tools/profile_summary.py pmc $R/gpurun_out/prof_pmc > $R/gpurun_out/${TAG}_pmc_$C.md
  rm -rf $R/gpurun_out/prof_pmc
  echo "pmc $C done"
done
rm -rf $R/gpurun_out/prof_pmc
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $R/gpurun_out/prof_pmc -o pmc --output-format csv -- $B > /dev/null 2> $R/gpurun_out/${TAG}_pmc_SQ.err
python3 $R/tools/profile_summary.py pmc $R/gpurun_out/prof_pmc > $R/gpurun_out/${TAG}_pmc_SQ.md
rm -rf $R/gpurun_out/prof_pmc
echo "pmc SQ done"
cp $R/profiles/pmc_traffic.json $R/gpurun_out/${TAG}_pmc_traffic.json
cd $R && python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
tail -c 600 gpurun_out/${TAG}_bench_default.json
