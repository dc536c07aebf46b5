#!/bin/bash
# HBM-side traffic of the k_conv<3> launches of the bench workload: separate FETCH_SIZE / WRITE_SIZE passes (run via gpurun)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_traffic
mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 250 rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o t --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/$c.log 2>&1
  f=$(find $OUT/$c -name "*counter_collection.csv" | head -1)
  echo "== $c ($f)"
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    k = r.get("Kernel_Name", "")
    if "k_conv<3" in k:
        acc["k_conv<3,*> (all)"].append(float(r["Counter_Value"]))
    if "k_conv" in k or "k_wgrad" in k:
        acc[k.split("(")[0][:44]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-46s launches %5d  mean %12.1f  sum %14.1f" % (k, len(v), sum(v) / len(v), sum(v)))
PY
done
