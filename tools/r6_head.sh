#!/bin/bash
# round 6: tile head of k_cdma (first fragment reads before the loader set-up, accumulators initialised straight from LDS)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/head.txt 2>&1
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -3
PD=$PWD/tools/_variants/predefer/libssdn_hip.so
NT=$PWD/tools/_variants/newT/libssdn_hip.so
SSDN_HIP_LIB=$PD timeout 300 python tools/cmp_libs.py dump /tmp/a.pt 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/cmp_libs.py dump /tmp/b.pt 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/cmp_libs.py diff /tmp/a.pt /tmp/b.pt 2>&1 | tail -2 | cut -c1-200
L="decode_block_1.0 decode_block_1.2 decode_block_2.0 decode_block_2.2 encode_block_1.2 encode_block_2.0"
for r in 1 2 3; do
  echo "== before (round $r)"; SSDN_HIP_LIB=$PD CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
  echo "== new (round $r)"; CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
done
echo "== trace"; SSDN_HIP_LIB=$NT timeout 300 python tools/conv_bench.py trace decode_block_1.2 fwd 2>&1 | grep "median\|per-WG"
for r in 1 2; do
for v in predefer new; do
echo "== bench $v"; LIBV=$PWD/tools/_variants/$v/libssdn_hip.so; [ $v = new ] && LIBV=$PWD/selfsupervised-denoising_amd/ssdn/hip/libssdn_hip.so
SSDN_HIP_LIB=$LIBV timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-trainer-leg 2>&1 | grep -v amdgpu.ids | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done
done
