#!/bin/bash
# round 6, step 2: correctness of the new k_cdma + ablations of a tuning build (where does a launch's time go now?)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
OUT=gpurun_out/r6/ab2.txt
exec > $OUT 2>&1
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -8
R5=$PWD/tools/_variants/r5base/libssdn_hip.so
NT=$PWD/tools/_variants/newT/libssdn_hip.so
SSDN_HIP_LIB=$R5 timeout 300 python tools/cmp_libs.py dump /tmp/a.pt 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/cmp_libs.py dump /tmp/b.pt 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/cmp_libs.py diff /tmp/a.pt /tmp/b.pt 2>&1 | tail -16
L="decode_block_1.2 decode_block_2.2"
for ab in 0 8 14 6 10 12 2 4 1 64; do
  echo "== newT SSDN_CDMA_ABLATE=$ab"; SSDN_HIP_LIB=$NT SSDN_CDMA_ABLATE=$ab CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
done
