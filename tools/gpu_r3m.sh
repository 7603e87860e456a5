#!/bin/bash
TAG=${1:-r3m}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
echo "== serial gpu suite (as the driver runs it)"; /usr/bin/time -v timeout 3000 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_serial.log 2> $OUT/pytest_serial.time; tail -3 $OUT/pytest_serial.log; grep "Elapsed" $OUT/pytest_serial.time
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== 2 ranks sharing the GPU over gloo (exercises the N > 1 bench path)"
MDT_BENCH_SHARE_GPU=1 MDT_BENCH_BACKEND=gloo MDT_BENCH_VERIFY_GATHER=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err; echo "exit $?"; python -c "
import json;d=json.loads([l for l in open('$OUT/bench_2rank.json') if l.startswith('{')][-1]);print(d['n_gpus'], d['value'], d['median_ms'], d['collective'])"; tail -3 $OUT/bench_2rank.err
