#!/bin/bash
# PMC counters for the isolated GEMM shapes (own run, kernel-trace only).  usage: tools/gpu_pmc.sh <tag>
TAG=${1:-pmc}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python tools/gemm_micro.py 50 2>/dev/null | tee $OUT/micro.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $OUT/p1 -o p -- python $OLDPWD/tools/gemm_micro.py 5 > /dev/null 2> $OUT/p1.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD -d $OUT/p2 -o p -- python $OLDPWD/tools/gemm_micro.py 5 > /dev/null 2> $OUT/p2.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $OUT/p3 -o p -- python $OLDPWD/tools/gemm_micro.py 5 > /dev/null 2> $OUT/p3.err
find $OUT -name "*.csv" | head; tail -3 $OUT/p1.err
find $OUT -name "*kernel_trace.csv" -delete
