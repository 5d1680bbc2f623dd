#!/bin/bash
# HBM traffic of the Go network forward: FETCH_SIZE / WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_net_traffic
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python3 $GRAFT_REPO_ROOT/benchmarks/bench_net.py --shape go --batch 2048 --steps 1 > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python3 $GRAFT_REPO_ROOT/benchmarks/bench_net.py --shape go --batch 2048 --steps 1 > $OUT/write.log 2>&1 || exit 1
ls $OUT
