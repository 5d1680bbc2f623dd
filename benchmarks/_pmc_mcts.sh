#!/bin/bash
# HBM traffic of the checkers tree-search kernel (FETCH_SIZE / WRITE_SIZE in separate passes)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mcts
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python3 $GRAFT_REPO_ROOT/benchmarks/bench_mcts.py --no-cpu-baseline --moves 4 --warmup 1 > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python3 $GRAFT_REPO_ROOT/benchmarks/bench_mcts.py --no-cpu-baseline --moves 4 --warmup 1 > $OUT/write.log 2>&1 || exit 1
ls $OUT
