#!/bin/bash
# round 6: k_conv_chain with one layer body per column-tile count (weight fragments keep their registers from item to item)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/chain.txt 2>&1
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -3
OLD=$PWD/tools/_variants/r6final/libssdn_hip.so
for r in 1 2; do
for d in fwd bwd; do
echo "== chain_bench $d old"; SSDN_HIP_LIB=$OLD timeout 300 python tools/chain_bench.py 32 64 $d 2>&1 | grep -v amdgpu.ids | tail -4
echo "== chain_bench $d new"; timeout 300 python tools/chain_bench.py 32 64 $d 2>&1 | grep -v amdgpu.ids | tail -4
done
done
for r in 1 2 3; do
for v in old new; do
echo "== bench $v"; LIBV=$OLD; [ $v = new ] && LIBV=$PWD/selfsupervised-denoising_amd/ssdn/hip/libssdn_hip.so
SSDN_HIP_LIB=$LIBV timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-trainer-leg 2>&1 | grep -v amdgpu.ids | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done
done
