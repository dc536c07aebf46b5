#!/bin/bash
# usage (through gpurun): tools/r6_gdw.sh <before> <after> <kernel pattern>: head/ops tests with the product library, bench A/B, kernel-trace A/B
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/gdw.txt 2>&1
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_denoiser.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2 3; do
for v in $1 $2; do
echo "== bench $v"
SSDN_HIP_LIB=$PWD/tools/_variants/$v/libssdn_hip.so timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-trainer-leg 2>&1 | grep -v amdgpu.ids | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done
done
bash tools/ab_kt.sh $1 $2 "$3"; bash tools/ab_kt.sh $1 $2 "$3"
