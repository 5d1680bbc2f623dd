#!/bin/bash
# same-box A/B of the network forward: the product library, then each variant library named on the command line
# (built by _build_prof_lib.sh).  The box's copy of the tree is scratch: variants are copied over elf_amd/libelfhip.so.
mkdir -p gpurun_out
set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_net.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/ab_base_tests.log 2>&1
tail -n 1 gpurun_out/ab_base_tests.log
run() {
  for r in 1 2; do
    timeout -k 10 120 python benchmarks/bench_net.py --shape go --batch 4096 --steps 6 > gpurun_out/ab_$1_go_$r.json 2>>gpurun_out/ab_err.log
  done
  timeout -k 10 120 python benchmarks/bench_net.py --shape checkers --batch 32768 --steps 6 > gpurun_out/ab_$1_ck_1.json 2>>gpurun_out/ab_err.log
}
run base
i=0
for v in "$@"; do
  i=$((i+1))
  cp "benchmarks/_prof/libelfhip_$v.so" elf_amd/libelfhip.so
  timeout -k 10 300 python -u -m pytest tests/test_gpu_net.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/ab_var${i}_tests.log 2>&1
  tail -n 1 gpurun_out/ab_var${i}_tests.log
  run var$i
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_*_*.json')):
    d=json.load(open(f)); print(f, round(d['ms_per_forward'],2), 'ms', round(d['roofline']['achieved'],1), 'TF')
PY
