#!/bin/bash
# effective shader clock of the board convolution: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (MI355X_MICROARCH.md, DVFS give-back)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_net_clock
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT -o clk -- python3 $GRAFT_REPO_ROOT/benchmarks/bench_net.py --shape go --batch 4096 --steps 2 > $OUT/clk.log 2>&1 || exit 1
ls $OUT
