#!/bin/bash
# round 6: what sits in the 5-7 us gaps of the main stream (kernel trace with every column kept)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
( cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/kt_gaps -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trainer-leg > /tmp/kt_gaps.log 2>&1 )
K=$(find /tmp/kt_gaps -name "*kernel_trace.csv" | head -1)
head -1 $K > gpurun_out/r6/gaps_kernel_trace_head.csv
python - $K <<'PY' > gpurun_out/r6/gaps.txt 2>&1
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one steady-state step: from the 12th k_conv_thin on
starts = [i for i, r in enumerate(rows) if "k_conv_thin" in r["Kernel_Name"]]
a, b = starts[12], starts[13]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = {}
for r in rows[a:b]:
    q = r.get("Queue_Id", "?")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    print("%9.1f %7.1f gap %5.1f q%s scratch %s lds %s vgpr %s sgpr %s wg %s grid %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, r.get("Private_Segment_Size", r.get("Scratch_Size", "?")),
          r.get("Group_Segment_Size", r.get("LDS_Block_Size", "?")), r.get("VGPR_Count", "?"), r.get("SGPR_Count", "?"), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r["Kernel_Name"][:48]))
PY
M=$(find /tmp/kt_gaps -name "*memory_copy_trace.csv" | head -1)
[ -n "$M" ] && head -30 $M > gpurun_out/r6/gaps_memcopy_head.csv
