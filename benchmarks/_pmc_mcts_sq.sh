#!/bin/bash
# instruction mix / wait fractions of the checkers tree-search kernel
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mcts
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT -o sq -- python3 $GRAFT_REPO_ROOT/benchmarks/bench_mcts.py --no-cpu-baseline --moves 4 --warmup 1 > $OUT/sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SMEM --output-format csv -d $OUT -o sq2 -- python3 $GRAFT_REPO_ROOT/benchmarks/bench_mcts.py --no-cpu-baseline --moves 4 --warmup 1 > $OUT/sq2.log 2>&1 || exit 1
