#!/bin/bash
TAG=${1:-r3g}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 2400 python -m pytest tests -m gpu -q -n 3 --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "gpu tests exit $?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest_gpu.log
for rep in 1 2; do timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_$rep.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/bench_$rep.json'));print(d['value'], d['median_ms'], d['p10_ms'], d['p90_ms'], d['roofline']['frac'])"; done
timeout 200 python tools/graph_probe.py 1 2>&1 | tail -4
timeout 200 python tools/graph_probe.py 4 2>&1 | tail -3
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err )
DB=$(find $OUT/prof -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/bench_kernel_stats.txt; grep -n "fold\|k_gemm<1, 1, 4\|smallm\|layernorm\|action_embed\|sigma_emb\|k_attn<48, 4" $OUT/bench_kernel_stats.txt | cut -c1-150
find $OUT -type f -size +20M -delete
