#!/bin/bash
# round 3, first GPU call: fused-MLP op tests, model parity, bench A/B (fused MLP on / off), kernel stats of the fused build
TAG=${1:-r3a}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
echo "== new op tests"; timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 300 -p no:cacheprovider -k "fused_mlp or slabs" > $OUT/pytest_new.log 2>&1; echo "new exit $?" | tee -a $OUT/summary.txt; tail -15 $OUT/pytest_new.log
echo "== parity"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_persist.py -m gpu -q -n 2 --timeout 600 -p no:cacheprovider -x > $OUT/pytest_parity.log 2>&1; echo "parity exit $?" | tee -a $OUT/summary.txt; tail -8 $OUT/pytest_parity.log
for rep in 1 2; do
echo "== bench fused"; timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_fused_$rep.json 2> $OUT/bench_fused.err; python -c "import json;d=json.load(open('$OUT/bench_fused_$rep.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'])"
echo "== bench unfused"; MDT_HIP_MLP_FUSE_MIN=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_unfused_$rep.json 2> $OUT/bench_unfused.err; python -c "import json;d=json.load(open('$OUT/bench_unfused_$rep.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'])"
done
echo "== kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err )
DB=$(find $OUT/prof -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/bench_kernel_stats.txt; head -14 $OUT/bench_kernel_stats.txt | cut -c1-170
find $OUT -type f -size +20M -delete
