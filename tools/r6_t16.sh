#!/bin/bash
# round 6: k_cdma on the 16x16 stage -- whole GPU suite, bench A/B against the previous library (same planner is not possible: the plan changes,
# so the A side is the committed HEAD~ state is not available; compare with the step times of this box's earlier runs), config sweep
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/t16.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for r in 1 2; do
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-trainer-leg 2>&1 | grep -v amdgpu.ids | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done
timeout 300 python tools/cfg_bench.py 100 2>&1 | grep -v amdgpu.ids
