#!/bin/bash
# PMC passes (MFMA-busy; HBM-side requests + L2 hit) over the training step and the masked-image head at B = 1024, each in its own
# rocprofv3 run with kernel-trace only.  usage: tools/gpu_train_pmc.sh <tag>
TAG=${1:-trainpmc}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
cd /tmp
for W in train mae; do
  if [ $W = train ]; then CMD="python $R/tools/train_bench.py 1024"; export MDT_TRAIN_BENCH_MODES=train MDT_TRAIN_BENCH_OPT=fused; WHAT="'MDT_TRAIN_BENCH_OPT=fused tools/train_bench.py 1024' (train mode: 23 denoiser steps, this package's FusedAdamW)";
  else CMD="python $R/tools/mae_bench.py 1024"; WHAT="'tools/mae_bench.py 1024' (masked-image head, forward + backward)"; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/$W/p1 -o p -- $CMD > $OUT/${W}_p1.txt 2> $OUT/${W}_p1.err
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/$W/p2 -o p -- $CMD > $OUT/${W}_p2.txt 2> $OUT/${W}_p2.err
  python $R/tools/bench_pmc_summary.py $OUT/$W "$WHAT" 40 > $OUT/${W}_pmc.txt
  head -30 $OUT/${W}_pmc.txt | cut -c1-200
  find $OUT/$W -name "*kernel_trace.csv" -delete; find $OUT/$W -name "*counter_collection.csv" -delete
done
