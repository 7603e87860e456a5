#!/bin/bash
# round 3: (a) the fixed C3 test, (b) bench with the CPU legs, (c) XCD-mapping A/B (product build vs -DMDT_NO_XCD_REMAP), (d) PMC passes of the bench command
TAG=${1:-r3c}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 600 python -m pytest tests/test_c3_step.py -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_c3.log 2>&1; echo "c3 exit $?" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest_c3.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; grep "cpu baseline leg" $OUT/bench.err | cut -c1-260
for rep in 1 2 3; do
  for v in product noremap; do
    if [ $v = noremap ]; then export MDT_HIP_LIB=$R/mdt_policy_amd/csrc/libmdt_hip_noremap.so; else unset MDT_HIP_LIB; fi
    timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/ab_${v}_$rep.json 2>/dev/null
    python -c "import json;d=json.load(open('$OUT/ab_${v}_$rep.json'));print('$v', d['value'], d['median_ms'], d['p10_ms'], d['p90_ms'], d['roofline']['dominant_kernel']['avg_us'])" | tee -a $OUT/xcd_ab.txt
  done
done
unset MDT_HIP_LIB
echo "== PMC passes"
( cd /tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d $OUT/pmc/p1 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p1.json 2> $OUT/p1.err
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc/p2 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p2.json 2> $OUT/p2.err
  for v in noremap; do MDT_HIP_LIB=$R/mdt_policy_amd/csrc/libmdt_hip_noremap.so timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_noremap/p2 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p2n.json 2> $OUT/p2n.err; done )
python tools/bench_pmc_summary.py $OUT/pmc > $OUT/bench_pmc.txt; head -12 $OUT/bench_pmc.txt | cut -c1-200
python tools/bench_pmc_summary.py $OUT/pmc_noremap > $OUT/bench_pmc_noremap.txt; head -8 $OUT/bench_pmc_noremap.txt | cut -c1-200
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +30M -delete; find $OUT -type f -size +30M -delete
