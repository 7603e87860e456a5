#!/bin/bash
# training step at the reference's per-GPU batch (128): wall vs kernel time.  usage: tools/gpu_r6_b128.sh <tag>
TAG=${1:-r6b128}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
export MDT_TRAIN_BENCH_MODES=train MDT_TRAIN_BENCH_OPT=fused
for i in 1 2; do python tools/train_bench.py 128 | grep B=; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/train -o train -- python $R/tools/train_bench.py 128 > $OUT/train_run.txt 2> $OUT/train.err )
DB=$(find $OUT/train -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/train_kernel_stats.txt; tail -1 $OUT/train_run.txt; head -30 $OUT/train_kernel_stats.txt | cut -c1-150
python tools/prof_gaps.py $DB 330 | head -5
MDT_HIP_DW_STREAM=0 python tools/train_bench.py 128 | grep B=
