#!/bin/bash
# MFMA-busy / HBM-side traffic of the WHOLE sampler (bench.py's own command), two separate --pmc passes with
# --kernel-trace only (never combined with sys/hip/hsa trace domains).  usage: tools/gpu_bench_pmc.sh <tag>
TAG=${1:-benchpmc}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d $OUT/p1 -o p -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p1.json 2> $OUT/p1.err; echo "pass1 exit $?"
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/p2 -o p -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p2.json 2> $OUT/p2.err; echo "pass2 exit $?"
cd $OLDPWD
python tools/bench_pmc_summary.py $OUT > $OUT/bench_pmc.txt; cat $OUT/bench_pmc.txt | head -40
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +30M -delete
