#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/race5.txt 2>&1
for impl in torch_cur torch_ext; do
echo "== fork through torch.cuda.Event, main stream as $impl (join through the library)"
SSDN_FORK_IMPL=$impl timeout 600 python tools/ab_sigma_race.py 2 2 12 2>&1 | grep -v amdgpu.ids | cut -c1-150
done
