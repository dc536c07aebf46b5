#!/bin/bash
# round 6, step 4: new k_cdma (rows issued early, post-store nops) -- tests, bit-identity against round 5, A/B, dephase sweep
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/ab4.txt 2>&1
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -6
R5=$PWD/tools/_variants/r5base/libssdn_hip.so
NT=$PWD/tools/_variants/newT/libssdn_hip.so
SSDN_HIP_LIB=$R5 timeout 300 python tools/cmp_libs.py dump /tmp/a.pt 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/cmp_libs.py dump /tmp/b.pt 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/cmp_libs.py diff /tmp/a.pt /tmp/b.pt 2>&1 | grep -v identical | cut -c1-200
L="decode_block_1.0 decode_block_1.2 decode_block_2.0 decode_block_2.2 encode_block_1.2 encode_block_2.0"
for r in 1 2; do
  echo "== r5base (round $r)"; SSDN_HIP_LIB=$R5 CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
  echo "== new (round $r)"; CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
done
for d in 0 2 4 6 8 12 16; do
  echo "== newT SSDN_CDMA_DEPHASE=$d"; SSDN_HIP_LIB=$NT SSDN_CDMA_DEPHASE=$d CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py decode_block_1.2 decode_block_2.2 decode_block_1.0 2>&1 | grep -v amdgpu.ids
done
for ab in 8 14 12 10 64; do
  echo "== newT SSDN_CDMA_ABLATE=$ab"; SSDN_HIP_LIB=$NT SSDN_CDMA_ABLATE=$ab CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py decode_block_1.2 2>&1 | grep -v amdgpu.ids
done
