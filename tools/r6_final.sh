#!/bin/bash
# round 6: the round's profile evidence (through gpurun): tools/collect_profiles.sh + per-configuration kernel stats, convergence record,
# same-box A/B against the round-5 library, the 2-rank gloo run of the bench on one GPU
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
bash tools/collect_profiles.sh > gpurun_out/final_collect.log 2>&1
OUT=$R/gpurun_out/final
cd /tmp
for c in "config 3" "config 4" "config 5"; do
  tag=$(echo $c | tr -d ' ')
  CFG_ONLY="$c" timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$tag -o kt --output-format csv -- python $R/tools/cfg_bench.py 30 > $OUT/kt_$tag.log 2>&1
  cp $(find $OUT/kt_$tag -name "*kernel_stats.csv" | head -1) $OUT/${tag}_kernel_stats.csv
  rm -rf $OUT/kt_$tag
done
cd $R
timeout 900 python tools/convergence.py --steps 300 --out $OUT/convergence.json > $OUT/convergence.log 2>&1
R5=$R/tools/_variants/r5base/libssdn_hip.so
for r in 1 2 3; do
  for v in r5base final; do
    LIBV=$R5; [ $v = final ] && LIBV=$R/selfsupervised-denoising_amd/ssdn/hip/libssdn_hip.so
    echo "== $v (round $r)" >> $OUT/ab_round5_library.txt
    SSDN_HIP_LIB=$LIBV timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-trainer-leg 2>&1 | grep -v amdgpu.ids | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('patches/s', d['value'], 'ms_per_step', d['ms_per_step'], 'k_cdma<3,*> TF/s', d['roofline']['achieved'], 'frac', d['roofline']['frac'])" >> $OUT/ab_round5_library.txt 2>&1
  done
done
SSDN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline --no-trainer-leg > $OUT/bench_2rank_gloo_one_gpu.json 2> $OUT/bench_2rank_gloo_one_gpu.err
ls -la $OUT
