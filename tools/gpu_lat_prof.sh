#!/bin/bash
# kernel-trace of the rollout batch (B = 1): per-kernel durations and the gaps between dependent launches
out=gpurun_out/$1; B=${2:-1}; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$out/prof -o lat -- python $R/tools/latency.py $B > $R/$out/prof_run.txt 2> $R/$out/prof.err
db=$(find $R/$out/prof -name "*.db" | head -1)
python $R/tools/prof_summary.py $db > $R/$out/lat_kernel_stats.txt
python $R/tools/prof_gaps.py $db 250 > $R/$out/lat_gaps.txt
cat $R/$out/prof_run.txt $R/$out/lat_gaps.txt
