#!/bin/bash
# round 6, step 3: which change breaks bit-identity?  (variants against the round-5 library, same inputs)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/ab3.txt 2>&1
R5=$PWD/tools/_variants/r5base/libssdn_hip.so
SSDN_HIP_LIB=$R5 timeout 300 python tools/cmp_libs.py dump /tmp/a.pt 2>&1 | grep -v amdgpu.ids
for v in "$@"; do
  for rep in 1 2; do
    echo "== variant $v (rep $rep)"
    SSDN_HIP_LIB=$PWD/tools/_variants/$v/libssdn_hip.so timeout 300 python tools/cmp_libs.py dump /tmp/b.pt 2>&1 | grep -v amdgpu.ids
    timeout 300 python tools/cmp_libs.py diff /tmp/a.pt /tmp/b.pt 2>&1 | grep -v identical | cut -c1-230
  done
done
