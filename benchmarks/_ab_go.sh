#!/bin/bash
# same-box A/B of the Go playout kernel: tests on the product library, then bench product / variants
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_go.py tests/test_gpu_go_mcts.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/g_tests.log 2>&1; tail -n 1 gpurun_out/g_tests.log
t() { timeout -k 10 120 python bench.py --no-extras --no-nn-rollouts --no-cpu-baseline > gpurun_out/g.json 2>>gpurun_out/g_err.log || exit 1; python -c "import json;d=json.load(open('gpurun_out/g.json'));print('$1', '%.4e'%d['value'], 'moves/s; kernel', round(d['roofline']['kernel_ms'],4), 'ms')"; }
t base; t base
for v in "$@"; do cp "benchmarks/_prof/libelfhip_$v.so" elf_amd/libelfhip.so; t $v; t $v; done
