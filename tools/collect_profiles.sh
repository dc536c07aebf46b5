#!/bin/bash
# Collect the round's profile evidence on the GPU box (run through gpurun); outputs under gpurun_out/final/.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd /tmp
# 1. un-profiled bench lines (the numbers the other files are compared with): long run, and the driver's default flags
timeout 600 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_line.json 2> $OUT/bench_line.err
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $OUT/bench_line_default_flags.json 2> $OUT/bench_line_default_flags.err
# 2. kernel trace + stats of the bench command
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trainer-leg > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
python $R/tools/timeline.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1) 10 list > $OUT/bench_timeline.txt 2>&1
# 3. HBM-side traffic (separate passes), per kernel family
bash $R/tools/pmc_traffic.sh > $OUT/traffic.txt 2>&1
cp $R/gpurun_out/pmc_traffic/traffic.json $OUT/traffic.json
# 4. SQ counters: the chip-wide weight-gradient launch (+ three of its ops alone), k_cdma on the full-resolution 96->96 layer, the head GEMM
bash $R/tools/pmc_mega.sh > $OUT/pmc_wgrad_mega.txt 2>&1
bash $R/tools/pmc_wgrad.sh decode_block_1.2 fwd > $OUT/pmc_cdma_fwd.txt 2>&1
bash $R/tools/pmc_wgrad.sh decode_block_1.2 dgrad > $OUT/pmc_cdma_dgrad.txt 2>&1
bash $R/tools/pmc_wgrad.sh output_block.0 fwd > $OUT/pmc_gdma_fwd.txt 2>&1
# 5. the weight-gradient plan: per-block timeline inside the launch, every op alone, every convolution alone
timeout 200 python $R/tools/wgrad_calib.py trace > $OUT/wgrad_mega_timeline.txt 2>&1
timeout 200 python $R/tools/wgrad_calib.py > $OUT/wgrad_ops_alone.txt 2>&1
timeout 200 python $R/tools/conv_bench.py all conv > $OUT/conv_isolated.txt 2>&1
# 6. every BASELINE configuration's shard through train_step
timeout 300 python $R/tools/cfg_bench.py 100 > $OUT/cfg_bench.txt 2>&1
rm -rf $OUT/kt
ls -la $OUT
