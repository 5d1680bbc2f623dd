#!/bin/bash
# where a Go playout wave's cycles go when it is not issuing: instruction fetch, waits, the second VALU counter.
# Usage (GPU box, repo root): bash benchmarks/_pmc_go_issue.sh [boards]
ROOT=$GRAFT_REPO_ROOT
B=${1:-65536}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_go_issue_$B
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_WAIT_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU --output-format csv -d $OUT -o a -- python3 $ROOT/bench.py --boards $B --no-cpu-baseline --no-nn-rollouts --no-extras --steps 5 --warmup 2 > $OUT/a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES --output-format csv -d $OUT -o b -- python3 $ROOT/bench.py --boards $B --no-cpu-baseline --no-nn-rollouts --no-extras --steps 5 --warmup 2 > $OUT/b.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "go_playout_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k, "%.4g" % (sum(v) / len(v)), "x", len(v))
PY
