#!/bin/bash
# round 6, step 6: phase stamps of the new k_cdma (where in a tile do the stores / the row stream cost?), LDS epilogue against the direct one
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/ab6.txt 2>&1
NT=$PWD/tools/_variants/newT/libssdn_hip.so
for ab in 0 64 128 8 4; do
  for role in fwd dgrad; do
    echo "== trace decode_block_1.2 $role SSDN_CDMA_ABLATE=$ab"
    SSDN_HIP_LIB=$NT SSDN_CDMA_ABLATE=$ab timeout 300 python tools/conv_bench.py trace decode_block_1.2 $role 2>&1 | grep -v amdgpu.ids
  done
done
for ab in 0 1024; do
  echo "== newT SSDN_CDMA_ABLATE=$ab (1024: epilogue through LDS, coalesced 1 KiB stores)"
  SSDN_HIP_LIB=$NT SSDN_CDMA_ABLATE=$ab CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py decode_block_1.2 decode_block_1.0 decode_block_2.2 2>&1 | grep -v amdgpu.ids
done
