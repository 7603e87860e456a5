#!/bin/bash
# rocprofv3 kernel stats of the denoiser training step (B = 1024, eval mode): gpurun_out/train_prof/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/train_prof; mkdir -p $R/gpurun_out/train_prof
cd $R
MDT_TRAIN_BENCH_MODES=${MODE:-eval} rocprofv3 --kernel-trace --stats -d gpurun_out/train_prof -o train -- python tools/train_bench.py 1024 > gpurun_out/train_prof/run.log 2>&1
grep "B=" gpurun_out/train_prof/run.log
