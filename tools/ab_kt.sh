# usage (through gpurun): tools/ab_kt.sh <variantA> <variantB> [pattern]: per-kernel average durations of the bench command with two library builds
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
A=$1; B=$2; PAT=${3:-k_}
for v in $A $B; do
  rm -rf /tmp/kt_$$; mkdir -p /tmp/kt_$$
  ( cd /tmp && SSDN_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/$v/libssdn_hip.so rocprofv3 --kernel-trace --stats -d /tmp/kt_$$ -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-trainer-leg > /tmp/kt_$$/log 2>&1 )
  echo "== $v"
  python - /tmp/kt_$$ "$PAT" <<'PY'
import sys, csv, glob, re
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows:
    n = r["Name"]
    if re.search(sys.argv[2], n):
        print("  %-60s calls %5s avg %9.1f us  total %9.1f ms" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
