#!/bin/bash
# Round-4 secondary measurements on the GPU box (one call): ablation tables, host trace, secondary bench lines.
#   bash tools/run_r04_extras.sh   ->  gpurun_out/r04x/*
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04x
mkdir -p $O
cd $R
python tools/att_bench.py > $O/att_ablation.txt 2>/dev/null
python tools/att_bench.py --drop --masks 0,16 > $O/att_ablation_drop.txt 2>/dev/null
: > $O/lstm_bench.txt
for v in 0 2 3 4; do MMB_LSTM_FWD_VARIANT=$v python tools/lstm_bench.py 2>/dev/null | grep variant >> $O/lstm_bench.txt; done
python tools/host_trace.py > $O/host_trace.txt 2>/dev/null
MMB_REGION_FN=0 python tools/host_trace.py > $O/host_trace_modular.txt 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --fresh-lengths > $O/fresh_lengths_bench.json 2>/dev/null
MMB_REGION_FN=0 python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --fresh-lengths > $O/fresh_lengths_modular_bench.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --eager > $O/eager_bench.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --drop-prob 0.2 > $O/drop02_bench.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --ragged > $O/ragged_bench.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --config cfg4 > $O/cfg4_bench.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline --config cfg5 > $O/cfg5_bf16_bench.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline --config cfg5 --dtype f32 > $O/cfg5_f32_bench.json 2>/dev/null
python tools/full_model_bench.py > $O/full_model.txt 2>/dev/null
python tools/decoder_bench.py > $O/decoder_bench.txt 2>/dev/null
ls -la $O
