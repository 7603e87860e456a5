#!/bin/bash
# what slows a wave that multiplies ALONE on its SIMD (k_mlp, partner held back for the whole first product): stamps with the
# k-steps' weight loads / LDS reads removed (timing only).  Builds: python -m mdt_policy_amd.build -DMDT_DEBUG_TIMING [-DMDT_EXP_NOLOAD] [-DMDT_EXP_NOLDS]
# --out=mdt_policy_amd/csrc/libmdt_hip_dbg[_nold|_nolds|_none].so
TAG=${1:-alone}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for lib in dbg dbg_nold dbg_nolds dbg_none; do for v in 511 0; do echo "== $lib skew $v"; MDT_HIP_MLP_SKEW=$v MDT_HIP_LIB=$R/mdt_policy_amd/csrc/libmdt_hip_$lib.so timeout 100 python tools/mlp_phases.py 2>&1 | grep "k_mlp:\|wave\|phase-1\|phase-2\|total"; done; done | tee $OUT/phases.txt
