#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/ab7.txt 2>&1
NT=$PWD/tools/_variants/newT/libssdn_hip.so
for d in 0 8; do
    echo "== trace decode_block_1.2 fwd SSDN_CDMA_DEPHASE=$d"
    SSDN_HIP_LIB=$NT SSDN_CDMA_DEPHASE=$d timeout 300 python tools/conv_bench.py trace decode_block_1.2 fwd 2>&1 | grep -v amdgpu.ids
done
