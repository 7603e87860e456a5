#!/bin/bash
# End-of-round measurement set (one gpurun call): bench line, kernel stats + PMC of the bench command, rollout (B = 1)
# trace, training-step and MGF-head kernel stats, batch sweep.  usage: tools/gpu_final.sh <tag>
TAG=${1:-final}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
echo "== bench"; timeout 420 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
echo "== kernel stats of the bench command"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err )
DB=$(find $OUT/prof -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/bench_kernel_stats.txt; head -12 $OUT/bench_kernel_stats.txt | cut -c1-150
echo "== PMC passes of the bench command"
( cd /tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d $OUT/pmc/p1 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p1.json 2> $OUT/p1.err
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc/p2 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p2.json 2> $OUT/p2.err )
python tools/bench_pmc_summary.py $OUT/pmc > $OUT/bench_pmc.txt; head -12 $OUT/bench_pmc.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
echo "== rollout batch 1"
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/lat -o lat -- python $R/tools/latency.py 1 > $OUT/lat_run.txt 2> $OUT/lat.err )
DB=$(find $OUT/lat -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/lat_kernel_stats.txt; python tools/prof_gaps.py $DB 250 > $OUT/lat_gaps.txt; head -9 $OUT/lat_gaps.txt | cut -c1-150
python tools/prof_call.py $DB 250 > $OUT/lat_call.txt   # the last call launch by launch (profiles/rNN_rollout_floor.txt reads it)
echo "== batch sweep"; timeout 300 python tools/latency.py 1 2 4 8 16 32 64 128 256 512 1024 2>&1 | grep B= | tee $OUT/sweep.txt
echo "== training step B=1024"
( cd /tmp && MDT_TRAIN_BENCH_MODES=train MDT_TRAIN_BENCH_OPT=fused timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/train -o train -- python $R/tools/train_bench.py 1024 > $OUT/train_run.txt 2> $OUT/train.err )
DB=$(find $OUT/train -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/train_kernel_stats.txt; cat $OUT/train_run.txt | tail -2
( export MDT_TRAIN_BENCH_OPT=fused; timeout 200 python tools/train_bench.py 128; timeout 200 python tools/train_bench.py 1024 ) 2>&1 | grep "B=" | tee $OUT/train_bench.txt
echo "== MGF head"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/mae -o mae -- python $R/tools/mae_bench.py 1024 > $OUT/mae_run.txt 2> $OUT/mae.err )
DB=$(find $OUT/mae -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/mae_kernel_stats.txt
timeout 200 python tools/mae_bench.py 128 1024 2>&1 | grep "B=" | tee $OUT/mae_bench.txt
echo "== PMC passes of the training step and the masked-image head (tools/gpu_train_pmc.sh)"
bash tools/gpu_train_pmc.sh $TAG/trainpmc > $OUT/trainpmc.log 2>&1; tail -2 $OUT/trainpmc/train_pmc.txt | cut -c1-200; tail -1 $OUT/trainpmc/mae_pmc.txt | cut -c1-200
find $OUT -type f -size +20M -delete
