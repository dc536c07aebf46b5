#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/last.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r6/bench_default.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6/bench_default.json"))
r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "achieved", r["achieved"], "traffic", r["traffic"], "alg", r["algorithmic_bytes_per_launch"], "ratio", (r["traffic"] or 0) / r["algorithmic_bytes_per_launch"])
print(r["traffic_unit"][:260])
print("trainer", d.get("value_trainer_hdf5"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "dp_plan", d["config"].get("dp_plan"))
print("second", d["roofline"].get("second_kernel", {}).get("achieved"))
PY
