#!/bin/bash
# round 6, step 9: KIND 2 (decode_block_1.0 forward), hoisted sign-byte loads, tap 1 requested before the epilogue
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/ab9.txt 2>&1
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -4
R5=$PWD/tools/_variants/r5base/libssdn_hip.so
MV=$PWD/tools/_variants/mv1/libssdn_hip.so
NT=$PWD/tools/_variants/newT/libssdn_hip.so
SSDN_HIP_LIB=$R5 timeout 300 python tools/cmp_libs.py dump /tmp/a.pt 2>&1 | grep -v amdgpu.ids
for rep in 1 2 3; do
timeout 300 python tools/cmp_libs.py dump /tmp/b.pt 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/cmp_libs.py diff /tmp/a.pt /tmp/b.pt 2>&1 | grep -v identical | cut -c1-200
done
L="decode_block_1.0 decode_block_1.2 decode_block_2.0 decode_block_2.2 encode_block_1.2 encode_block_2.0"
for r in 1 2; do
  echo "== mv1: previous state (round $r)"; SSDN_HIP_LIB=$MV CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
  echo "== new (round $r)"; CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
done
for role in fwd dgrad; do
  echo "== trace decode_block_1.2 $role"
  SSDN_HIP_LIB=$NT timeout 300 python tools/conv_bench.py trace decode_block_1.2 $role 2>&1 | grep -v amdgpu.ids | grep -v "HW_ID\|XCD 0\|WG start"
done
for r in 1 2; do
echo "== bench r5base"; SSDN_HIP_LIB=$R5 timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-trainer-leg 2>&1 | grep -v amdgpu.ids | cut -c1-260
echo "== bench new"; timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-trainer-leg 2>&1 | grep -v amdgpu.ids | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done
