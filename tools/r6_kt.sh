#!/bin/bash
# kernel trace + timeline of the bench step, and the whole GPU test-suite (through gpurun)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6kt
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 600 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-trainer-leg > $OUT/bench_line.json 2> $OUT/bench_line.err
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trainer-leg > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
python $R/tools/timeline.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1) 10 list > $OUT/bench_timeline.txt 2>&1
rm -rf $OUT/kt
cd $R
if [ "${1:-}" = "tests" ]; then timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt; fi
