#!/bin/bash
# round 6, step 8: all-full specialisation (one K-loop body per role), with the barrier in front of / behind the step's last K-step
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/ab8.txt 2>&1
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -4
R5=$PWD/tools/_variants/r5base/libssdn_hip.so
MV=$PWD/tools/_variants/mv1/libssdn_hip.so
SSDN_HIP_LIB=$R5 timeout 300 python tools/cmp_libs.py dump /tmp/a.pt 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/cmp_libs.py dump /tmp/b.pt 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/cmp_libs.py diff /tmp/a.pt /tmp/b.pt 2>&1 | tail -3 | cut -c1-200
SSDN_HIP_LIB=$MV timeout 300 python tools/cmp_libs.py dump /tmp/c.pt 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/cmp_libs.py diff /tmp/a.pt /tmp/c.pt 2>&1 | tail -3 | cut -c1-200
L="decode_block_1.0 decode_block_1.2 decode_block_2.0 decode_block_2.2 encode_block_1.2 encode_block_2.0"
for r in 1 2; do
  echo "== r5base (round $r)"; SSDN_HIP_LIB=$R5 CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
  echo "== new MOVE=0 (round $r)"; CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
  echo "== new MOVE=1 (round $r)"; SSDN_HIP_LIB=$MV CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
done
