#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/chain_trace.txt 2>&1
T=$PWD/tools/_variants/chainT/libssdn_hip.so
for d in fwd bwd; do
echo "=== $d"; SSDN_LIB=$T SSDN_HIP_LIB=$T timeout 300 python tools/chain_bench.py 32 64 $d 2>&1 | grep -v amdgpu.ids
done
