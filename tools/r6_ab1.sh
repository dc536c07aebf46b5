#!/bin/bash
# round 6, step 1: new k_cdma (role-specialised K loop) against the round-5 library on one box
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
OUT=gpurun_out/r6/ab1.txt
exec > $OUT 2>&1
set -x
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -15
R5=$PWD/tools/_variants/r5base/libssdn_hip.so
SSDN_HIP_LIB=$R5 timeout 300 python tools/cmp_libs.py dump /tmp/a.pt 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/cmp_libs.py dump /tmp/b.pt 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/cmp_libs.py diff /tmp/a.pt /tmp/b.pt 2>&1 | tail -30
L="decode_block_1.0 decode_block_1.2 decode_block_2.0 decode_block_2.2 encode_block_1.2 encode_block_2.0"
for r in 1 2; do
  echo "== r5base (round $r)"; SSDN_HIP_LIB=$R5 CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
  echo "== new (round $r)"; CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
done
