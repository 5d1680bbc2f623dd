#!/bin/bash
# PMC pass over the Go network forward (one counter set per run; see MI355X_MICROARCH.md rocprofv3 section)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_net
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o sq1 -- python $GRAFT_REPO_ROOT/benchmarks/bench_net.py --shape go --batch 2048 --steps 1 > $OUT/sq1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT -o sq2 -- python $GRAFT_REPO_ROOT/benchmarks/bench_net.py --shape go --batch 2048 --steps 1 > $OUT/sq2.log 2>&1 || exit 1
ls $OUT
