#!/bin/bash
# round 6: k_cdma with the L2 warm-up of its weights.  usage (through gpurun): tools/r6_cdw.sh <before> <after>
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/cdw.txt 2>&1
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -2
A=$PWD/tools/_variants/$1/libssdn_hip.so; B=$PWD/tools/_variants/$2/libssdn_hip.so
L="decode_block_1.0 decode_block_1.2 decode_block_2.0 decode_block_2.2 encode_block_1.2 encode_block_2.0"
for r in 1 2; do
  echo "== $1 (round $r)"; SSDN_HIP_LIB=$A CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
  echo "== $2 (round $r)"; SSDN_HIP_LIB=$B CONV_BENCH_ONLY_DEFAULT=1 timeout 300 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
done
for r in 1 2 3; do
for v in $1 $2; do
echo "== bench $v"
SSDN_HIP_LIB=$PWD/tools/_variants/$v/libssdn_hip.so timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-trainer-leg 2>&1 | grep -v amdgpu.ids | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done
done
bash tools/ab_kt.sh $1 $2 "k_cdma|k_conv_chain"
