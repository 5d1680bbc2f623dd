#!/bin/bash
# rocprofv3 evidence for bench.py's dominant kernel (go_playout_kernel): kernel-trace stats of the exact
# bench command, then FETCH_SIZE / WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM section).
# Usage (on the GPU box, from the repo root): bash benchmarks/_prof_bench.sh <tag>
TAG=${1:-r02}
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python3 $ROOT/bench.py --no-cpu-baseline --no-nn-rollouts --no-extras --steps 20 > $OUT/stats.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python3 $ROOT/bench.py --no-cpu-baseline --no-nn-rollouts --no-extras --steps 5 --warmup 2 > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python3 $ROOT/bench.py --no-cpu-baseline --no-nn-rollouts --no-extras --steps 5 --warmup 2 > $OUT/write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o sq -- python3 $ROOT/bench.py --no-cpu-baseline --no-nn-rollouts --no-extras --steps 5 --warmup 2 > $OUT/sq.log 2>&1 || exit 1
find $OUT -name "*.csv" | head -20
# the whole default command (both legs + extras), kernel trace only
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o full -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/full.log 2>&1 || exit 1
