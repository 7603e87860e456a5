OUT=$PWD/gpurun_out/r4e; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for g in 0 23; do
( cd /tmp && MDT_HIP_GEO_TALL=$g timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/mae$g -o mae -- python $R/tools/mae_bench.py 1024 > $OUT/mae_run$g.txt 2> $OUT/mae$g.err )
DB=$(find $OUT/mae$g -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/mae_kernel_stats$g.txt; echo "== GEO_TALL=$g"; head -22 $OUT/mae_kernel_stats$g.txt | cut -c1-150
find $OUT/mae$g -type f -size +5M -delete
done
