#!/bin/bash
# round 6, step 5: what the tile stream and the stores cost (tuning knobs), and the whole step with both libraries
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/ab5.txt 2>&1
R5=$PWD/tools/_variants/r5base/libssdn_hip.so
NT=$PWD/tools/_variants/newT/libssdn_hip.so
for ab in 0 128 136 256 512 768 64 192; do
  echo "== newT SSDN_CDMA_ABLATE=$ab  (128 rows of image 0 only, 256 nt stores, 512 nt row DMA, 64 stores dropped, 8 no epilogue)"
  SSDN_HIP_LIB=$NT SSDN_CDMA_ABLATE=$ab CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py decode_block_1.2 decode_block_1.0 2>&1 | grep -v amdgpu.ids
done
for r in 1 2; do
  echo "== bench r5base (round $r)"; SSDN_HIP_LIB=$R5 timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-trainer-leg 2>&1 | grep -v amdgpu.ids | cut -c1-1500
  echo "== bench new (round $r)"; timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-trainer-leg 2>&1 | grep -v amdgpu.ids | cut -c1-1500
done
