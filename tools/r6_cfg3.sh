#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/cfg3.txt 2>&1
timeout 900 python -m pytest tests/test_hip_fullsize.py tests/test_hip_denoiser.py -x -q -m gpu 2>&1 | tail -4
for r in 1 2; do
  SIGMA_CONCURRENT=0 CFG_ONLY="config 3" timeout 300 python tools/cfg_bench.py 200 2>&1 | grep -v amdgpu.ids
  SIGMA_CONCURRENT=1 CFG_ONLY="config 3" timeout 300 python tools/cfg_bench.py 200 2>&1 | grep -v amdgpu.ids
done
timeout 300 python tools/cfg_bench.py 100 2>&1 | grep -v amdgpu.ids
