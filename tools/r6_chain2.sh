#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
bash tools/r6_chain.sh
bash tools/r6_chain_tr.sh
