#!/bin/bash
# Round-5 secondary measurements on the GPU box (one call): phase stamps, host trace, secondary bench lines, cfg4 / cfg5 kernel stats.
#   bash tools/run_r05_extras.sh   ->  gpurun_out/r05x/*
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05x
mkdir -p $O
cd $R
python tools/att_phases.py > $O/att_phases.txt 2>/dev/null
python tools/att_phases.py --drop > $O/att_phases_drop.txt 2>/dev/null
python tools/host_trace.py > $O/host_trace.txt 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --fresh-lengths > $O/fresh_lengths_bench.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --eager > $O/eager_bench.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --drop-prob 0.2 > $O/drop02_bench.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --ragged > $O/ragged_bench.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --config cfg4 > $O/cfg4_bench.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --config cfg1 > $O/cfg1_bench.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline --config cfg5 > $O/cfg5_bf16_bench.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline --config cfg5 --dtype f32 > $O/cfg5_f32_bench.json 2>/dev/null
MMB_DX_ATT=0 python bench.py --steps 60 --warmup 10 --no-secondary --no-cpu-baseline > $O/dx_att_off_bench.json 2>/dev/null
MMB_ATT_SREUSE=0 python bench.py --steps 60 --warmup 10 --no-secondary --no-cpu-baseline > $O/sreuse_off_bench.json 2>/dev/null
python tools/full_model_bench.py > $O/full_model.txt 2>/dev/null
python tools/decoder_bench.py > $O/decoder_bench.txt 2>/dev/null
python tools/gemm_bf16_bench.py > $O/gemm_bf16_bench.txt 2>/dev/null
bash tools/prof_cfg.sh cfg4 r05 > /dev/null 2>&1
bash tools/prof_cfg.sh cfg5 r05 > /dev/null 2>&1
bash tools/prof_cfg.sh cfg2 r05_drop02 --drop-prob 0.2 > /dev/null 2>&1
ls -la $O
