#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
timeout 300 python tools/r6_mode2.py > gpurun_out/r6/mode2.txt 2>&1
timeout 900 python -m pytest tests/test_hip_denoiser.py -x -q -m gpu -k "sigma_network or trajectory" >> gpurun_out/r6/mode2.txt 2>&1
