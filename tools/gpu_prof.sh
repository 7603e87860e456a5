#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary python tool: tools/gpu_prof.sh <outdir> <script> [args...]
out=gpurun_out/$1; shift; mkdir -p $out; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$out/prof -o p -- python $R/$@ > $R/$out/run.txt 2> $R/$out/prof.err
db=$(find $R/$out/prof -name "*.db" | head -1)
python $R/tools/prof_summary.py $db > $R/$out/kernel_stats.txt
find $R/$out/prof -type f -size +20M -delete
cat $R/$out/run.txt; head -34 $R/$out/kernel_stats.txt | cut -c1-160
