#!/bin/bash
# rocprofv3 kernel stats of the masked-image head alone (B = 1024): gpurun_out/mae_prof/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/mae_prof; mkdir -p $R/gpurun_out/mae_prof
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/mae_prof -o mae -- python tools/mae_bench.py 1024 > gpurun_out/mae_prof/run.log 2>&1
f=$(ls gpurun_out/mae_prof/*/*kernel_stats.csv gpurun_out/mae_prof/*kernel_stats.csv 2>/dev/null | head -1)
echo "stats file: $f"
head -14 "$f" | cut -c1-150
tail -3 gpurun_out/mae_prof/run.log
