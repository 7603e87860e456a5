#!/bin/bash
# rollout batches: the cross-attention inside the c_fc launch (k_xattn_gemm_smallm) against its own launch
TAG=${1:-rollout_ab}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "inside_the_linear or collapsed" > $OUT/pytest_op.log 2>&1; echo "op tests exit $?"; tail -4 $OUT/pytest_op.log
for rep in 1 2; do
echo "cross-attention inside c_fc (B <= 2)"; LAT_DEV_SIG=1 timeout 300 python tools/latency.py 1 2 4 8 16 2>&1 | grep B= | tee -a $OUT/lat_fused.txt
echo "own launch"; LAT_DEV_SIG=1 MDT_HIP_XATTN_FC_MAX_B=0 timeout 300 python tools/latency.py 1 2 4 8 16 2>&1 | grep B= | tee -a $OUT/lat_own.txt
done
echo "inside c_fc up to B = 8"; LAT_DEV_SIG=1 MDT_HIP_XATTN_FC_MAX_B=8 timeout 300 python tools/latency.py 4 8 2>&1 | grep B=
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/lat -o lat -- python $R/tools/latency.py 1 > $OUT/lat_run.txt 2> $OUT/lat.err )
DB=$(find $OUT/lat -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/lat_kernel_stats.txt; head -9 $OUT/lat_kernel_stats.txt | cut -c1-150
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "g1_ or g3_ or golden or persist or graph" > $OUT/pytest_sel.log 2>&1; echo "selected tests exit $?"; tail -4 $OUT/pytest_sel.log
