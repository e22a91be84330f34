#!/bin/bash
# Secondary measurements of a round (GPU box, one call): the long-sequence configuration (cfg4) with kernel stats and PMC
# traffic, the ragged-length and training-mode (drop_prob 0.2) lines of the metric configuration, the whole-model figure.
#   tools/run_extra_profiles.sh r03   ->  gpurun_out/r03_cfg4_*, r03_ragged_bench.json, r03_drop02_bench.json, r03_full_model.txt
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
B4="python3 $R/bench.py --config cfg4 --steps 3 --warmup 2 --no-cpu-baseline --eager"
rm -rf $R/gpurun_out/prof_stats
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o trace --output-format csv -- $B4 > /dev/null 2> $R/gpurun_out/${TAG}_cfg4_rocprof.err
python3 $R/tools/profile_summary.py stats $R/gpurun_out/prof_stats > $R/gpurun_out/${TAG}_cfg4_kernel_stats.md
rm -rf $R/gpurun_out/prof_stats
echo "cfg4 stats done"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/prof_pmc
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/prof_pmc -o pmc --output-format csv -- $B4 > /dev/null 2> $R/gpurun_out/${TAG}_cfg4_pmc_$C.err
  python3 $R/tools/profile_summary.py pmc $R/gpurun_out/prof_pmc cfg4 > $R/gpurun_out/${TAG}_cfg4_pmc_$C.md
  rm -rf $R/gpurun_out/prof_pmc
  echo "cfg4 pmc $C done"
done
cp $R/profiles/pmc_traffic.json $R/gpurun_out/${TAG}_pmc_traffic.json
cd $R
python3 bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_cfg4_bench.json 2> gpurun_out/${TAG}_cfg4_bench.err
echo "cfg4 bench done"
python3 bench.py --ragged --no-cpu-baseline > gpurun_out/${TAG}_ragged_bench.json 2> gpurun_out/${TAG}_ragged_bench.err
python3 bench.py --drop-prob 0.2 --no-cpu-baseline > gpurun_out/${TAG}_drop02_bench.json 2> gpurun_out/${TAG}_drop02_bench.err
python3 bench.py --eager --no-cpu-baseline > gpurun_out/${TAG}_eager_bench.json 2> gpurun_out/${TAG}_eager_bench.err
python3 tools/full_model_bench.py > gpurun_out/${TAG}_full_model.txt 2>&1 || true
tail -3 gpurun_out/${TAG}_full_model.txt
