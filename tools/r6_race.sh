#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/race4.txt 2>&1
timeout 900 python tools/ab_sigma_race.py 3 3 16 2>&1 | grep -v amdgpu.ids | cut -c1-160
for r in 1 2; do
  SIGMA_CONCURRENT=0 CFG_ONLY="config 3" timeout 300 python tools/cfg_bench.py 200 2>&1 | grep -v amdgpu.ids
  SIGMA_CONCURRENT=3 CFG_ONLY="config 3" timeout 300 python tools/cfg_bench.py 200 2>&1 | grep -v amdgpu.ids
done
