#!/bin/bash
# round-6 baseline / A-B set: bench line, training-step kernel stats (train mode), rollout latency.  usage: tools/gpu_r6_base.sh <tag>
TAG=${1:-r6base}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
echo "== bench"; timeout 420 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<PY
import json
j = json.load(open("$OUT/bench.json"))
o = j.get("other_configs", {})
print("value", j["value"], "ms", j["ms_per_step"], "frac", j["roofline"]["frac"])
for k in ("train_step_mdtv_B1024", "train_step_c3_mdtv_B1024", "rollout_B1_10steps"):
    print(k, {a: b for a, b in o.get(k, {}).items() if a in ("ms_per_step", "ms_per_chunk", "ms_per_chunk_pipelined", "error")})
PY
echo "== training step B=1024 (train mode) kernel stats"
( cd /tmp && MDT_TRAIN_BENCH_MODES=train timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/train -o train -- python $R/tools/train_bench.py 1024 > $OUT/train_run.txt 2> $OUT/train.err )
DB=$(find $OUT/train -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/train_kernel_stats.txt; tail -1 $OUT/train_run.txt; head -40 $OUT/train_kernel_stats.txt | cut -c1-160
( MDT_TRAIN_BENCH_MODES=train timeout 200 python tools/train_bench.py 1024 ) 2>&1 | grep "B=" | tee $OUT/train_bench.txt
find $OUT -type f -size +20M -delete
