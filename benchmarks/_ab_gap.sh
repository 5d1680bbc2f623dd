#!/bin/bash
# same-box A/B: product library vs a variant library built by _build_prof_lib.sh (the box's copy of the tree is scratch)
VAR=${1:-CONV_GAP_DMA}
mkdir -p gpurun_out
set -e
for r in 1 2; do
  timeout -k 10 120 python benchmarks/bench_net.py --shape go --batch 4096 --steps 6 > gpurun_out/ab_base_go_$r.json 2>gpurun_out/ab_err.log
  timeout -k 10 120 python benchmarks/bench_net.py --shape checkers --batch 32768 --steps 6 > gpurun_out/ab_base_ck_$r.json 2>>gpurun_out/ab_err.log
done
cp benchmarks/_prof/libelfhip_$VAR.so elf_amd/libelfhip.so
timeout -k 10 300 python -u -m pytest tests/test_gpu_net.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/ab_var_tests.log 2>&1
for r in 1 2; do
  timeout -k 10 120 python benchmarks/bench_net.py --shape go --batch 4096 --steps 6 > gpurun_out/ab_var_go_$r.json 2>>gpurun_out/ab_err.log
  timeout -k 10 120 python benchmarks/bench_net.py --shape checkers --batch 32768 --steps 6 > gpurun_out/ab_var_ck_$r.json 2>>gpurun_out/ab_err.log
done
tail -n 2 gpurun_out/ab_var_tests.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_*_*.json')):
    d=json.load(open(f)); print(f, round(d['ms_per_forward'],2), 'ms', round(d['roofline']['achieved'],1), 'TF')
PY
