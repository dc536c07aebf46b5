#!/bin/bash
# (needs a library built with `make -C selfsupervised-denoising_amd/csrc clean all TUNING=1`: the ablation bits and stamps are compiled out otherwise)
# tuning aid: phase trace + ablations of k_cdma on the full-resolution 96->96 layer (run through gpurun)
cd $GRAFT_REPO_ROOT
export CONV_BENCH_ONLY_DEFAULT=1
python tools/conv_bench.py trace decode_block_1.2 fwd 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py trace decode_block_1.2 dgrad 2>&1 | grep -v amdgpu.ids
for ab in ${ABLATES:-0 2 4 8 14}; do
  echo "== SSDN_CDMA_ABLATE=$ab (1 no MFMA, 2 no weight DMA, 4 no tile DMA, 8 no epilogue)"
  SSDN_CDMA_ABLATE=$ab python tools/conv_bench.py decode_block_1.2 decode_block_2.2 encode_block_1.2 2>&1 | grep -v amdgpu.ids
done
