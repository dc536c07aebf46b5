#!/bin/bash
# Kernel trace of `DenoiserTrainer.train()` over a synthetic HDF5 file (bench.py's trainer leg): which kernels run in a training step?
# (H11: only k_* kernels between print intervals.)  Run through gpurun.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_trainer
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python -c "
import sys; sys.path[:0]=['$R/selfsupervised-denoising_amd','$R']
import bench
print(bench.trainer_leg(32, 64, 200, workers=4))
" > $OUT/run.log 2>&1
tail -2 $OUT/run.log
python - "$(find $OUT/kt -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 230.0
print("%-90s %8s %10s" % ("kernel", "calls", "per step"))
for r in sorted(rows, key=lambda r: -int(r["Calls"])):
    print("%-90s %8d %10.2f" % (r["Name"][:90], int(r["Calls"]), int(r["Calls"]) / steps))
PY
rm -rf $OUT/kt
