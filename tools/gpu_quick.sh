#!/bin/bash
# quick check of a build: the GPU tests (or a -k subset), then the batch-size latencies.  usage: tools/gpu_quick.sh <tag> ["pytest -k expr"] [batches]
TAG=${1:-quick}; K=${2:-}; BS=${3:-"1 256"}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ -n "$K" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -4; else timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4; fi
timeout 300 python tools/latency.py $BS 2>&1 | grep "B=" | tee $OUT/lat.txt
