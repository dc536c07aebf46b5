#!/bin/bash
# round 6: k_conv_chain -- start-up planes in flight together; the 96-channel 16x16 layers inside the chains (SSDN_CHAIN_MODE=5)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/chain3.txt 2>&1
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -3
echo "== ops tests, wide chains"; SSDN_CHAIN_MODE=5 timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "chain" 2>&1 | tail -3
OLD=$PWD/tools/_variants/r6final/libssdn_hip.so
T=$PWD/tools/_variants/chainT/libssdn_hip.so
for d in fwd bwd; do
echo "== chain_bench $d old"; SSDN_HIP_LIB=$OLD timeout 300 python tools/chain_bench.py 32 64 $d 2>&1 | grep "chain 1"
echo "== chain_bench $d new"; timeout 300 python tools/chain_bench.py 32 64 $d 2>&1 | grep "chain 1"
done
for d in fwd bwd; do
echo "=== trace $d"; SSDN_LIB=$T SSDN_HIP_LIB=$T timeout 300 python tools/chain_bench.py 32 64 $d 2>&1 | grep -v amdgpu.ids
done
for d in fwd bwd; do
echo "=== trace wide $d"; SSDN_CHAIN_DEBUG=1 SSDN_CHAIN_MODE=5 SSDN_LIB=$T SSDN_HIP_LIB=$T timeout 300 python tools/chain_bench.py 32 64 $d 2>&1 | grep -v amdgpu.ids | grep -v "^chain 0\|^chain 1" | head -120
done
for r in 1 2 3; do
for v in old new wide; do
echo "== bench $v"; LIBV=$OLD; [ $v != old ] && LIBV=$PWD/selfsupervised-denoising_amd/ssdn/hip/libssdn_hip.so
M=1; [ $v = wide ] && M=5
SSDN_CHAIN_MODE=$M SSDN_HIP_LIB=$LIBV timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-trainer-leg 2>&1 | grep -v amdgpu.ids | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done
done
