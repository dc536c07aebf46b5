#!/bin/bash
# SQ counter passes over one MFMA-kernel launch: tools/pmc_wgrad.sh <layer> <fwd|dgrad|wgrad> (measurement aid; run through gpurun)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_wgrad
mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p$i --output-format csv -- python $R/tools/conv_bench.py trace ${1:-decode_block_1.2} ${2:-wgrad} > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $set  ($f)"
  python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    if "k_wgrad" in k or "k_conv" in k or "k_cdma" in k or "k_gdma" in k:
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: (len(v), sum(v) / len(v)) for c, v in d.items()})
PY
done
