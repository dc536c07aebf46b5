#!/bin/bash
# round 6: deferred epilogue stores of k_cdma -- tests, bit-identity against round 5, A/B against the state before
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/defer.txt 2>&1
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -4
R5=$PWD/tools/_variants/r5base/libssdn_hip.so
PD=$PWD/tools/_variants/predefer/libssdn_hip.so
SSDN_HIP_LIB=$R5 timeout 300 python tools/cmp_libs.py dump /tmp/a.pt 2>&1 | grep -v amdgpu.ids
for rep in 1 2 3; do
timeout 300 python tools/cmp_libs.py dump /tmp/b.pt 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/cmp_libs.py diff /tmp/a.pt /tmp/b.pt 2>&1 | grep -v identical | cut -c1-200
done
L="decode_block_1.0 decode_block_1.2 decode_block_2.0 decode_block_2.2 encode_block_1.2 encode_block_2.0"
for r in 1 2; do
  echo "== before (round $r)"; SSDN_HIP_LIB=$PD CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
  echo "== deferred stores (round $r)"; CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
done
for r in 1 2; do
for v in r5base predefer new; do
echo "== bench $v"; LIBV=$PWD/tools/_variants/$v/libssdn_hip.so; [ $v = new ] && LIBV=$PWD/selfsupervised-denoising_amd/ssdn/hip/libssdn_hip.so
SSDN_HIP_LIB=$LIBV timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-trainer-leg 2>&1 | grep -v amdgpu.ids | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done
done
