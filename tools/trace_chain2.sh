#!/bin/bash
# kernel timeline of a training step with the chained launches limited to 8x8 images (ssdn_conv_set_chain(2)): measurement aid
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_chain2
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt --output-format csv -- python -c "
import sys; sys.path[:0]=['$R/selfsupervised-denoising_amd','$R']
import torch, bench as B
from ssdn.hip import lib as L
from ssdn.denoiser import Denoiser
from ssdn.datasets import DevicePatchStream, NoisyDataset
from ssdn.params import NoiseAlgorithm
dev=torch.device('cuda',0)
nd=NoisyDataset(None,'gauss25',NoiseAlgorithm.SELFSUPERVISED_DENOISING,pad_uniform=False,pad_multiple=32,square=True,training_mode=True)
u8=torch.randint(0,256,(32,3,64,64),dtype=torch.uint8).pin_memory()
d=Denoiser(B.make_cfg(),device='cuda:0'); d.train()
st=DevicePatchStream(None,nd,dev,seed=1).attach(d)
L.load().ssdn_conv_set_chain(int(sys.argv[1]))
for i in range(20):
    d.train_step(st.prepare(st.upload(u8), torch.arange(32)), 3e-4, None)
torch.cuda.synchronize()
" ${1:-2} > $OUT/run.log 2>&1
python $R/tools/timeline.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1) 10 list
rm -rf $OUT/kt
