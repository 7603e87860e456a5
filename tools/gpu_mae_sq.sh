#!/bin/bash
# SQ-side counters of the masked-image head's kernels (occupancy, waits, instruction mix): one --pmc pass, kernel-trace only
TAG=${1:-maesq}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
cd /tmp
MDT_HIP_GLU_TALL=0 timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/p1 -o p -- python $R/tools/mae_bench.py 1024 > $OUT/p1.txt 2> $OUT/p1.err
MDT_HIP_GLU_TALL=0 timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 -d $OUT/p2 -o p -- python $R/tools/mae_bench.py 1024 > $OUT/p2.txt 2> $OUT/p2.err
cd $R; python tools/pmc_table.py $OUT/p1 16 > $OUT/sq1.txt; python tools/pmc_table.py $OUT/p2 16 > $OUT/sq2.txt; cat $OUT/sq1.txt | cut -c1-230; cat $OUT/sq2.txt | cut -c1-230
find $OUT -name "*.csv" -size +1M -delete
