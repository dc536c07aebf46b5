set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/s12
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/kt -o kt --output-format csv -- python -c "
import sys; sys.path[:0]=['$R/selfsupervised-denoising_amd','$R']
import bench
print(bench.trainer_leg(32, 64, 200, workers=8))
" > $OUT/run.log 2>&1
tail -2 $OUT/run.log
T=$(find $OUT/kt -name '*kernel_trace.csv' | head -1)
for s in 100 120 140 160; do python $R/tools/timeline.py $T $s list | head -3; done
python $R/tools/timeline.py $T 150 list > $OUT/trainer_timeline.txt
python - "$T" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
starts = [k[0] for k in ks if "k_conv_thin" in k[2]]
d = [ (b - a) / 1e3 for a, b in zip(starts[60:220], starts[61:221]) ]
d.sort()
print("step period (us) over steps 60..220: min %.1f median %.1f p90 %.1f max %.1f mean %.1f" % (d[0], d[len(d)//2], d[int(len(d)*0.9)], d[-1], sum(d)/len(d)))
PY
rm -rf $OUT/kt
