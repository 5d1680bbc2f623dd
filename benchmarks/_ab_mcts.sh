#!/bin/bash
# same-box A/B of the checkers search: the product library, then variant libraries (built by _build_prof_lib.sh or saved copies)
mkdir -p gpurun_out
t() { timeout -k 10 200 python benchmarks/bench_mcts.py --no-cpu-baseline > gpurun_out/m.json 2>>gpurun_out/m_err.log || exit 1; python -c "import json;d=json.load(open('gpurun_out/m.json'));print('$1', '%.3e'%d['value'], 'rollouts/s; kernel', round(d['roofline']['kernel_ms'],2), 'ms; move', round(d['ms_per_move'],2), 'ms')"; }
timeout -k 10 300 python -u -m pytest tests/test_gpu_mcts_quad.py tests/test_gpu_mcts.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/m_tests.log 2>&1; tail -n 1 gpurun_out/m_tests.log
t base; t base
for v in "$@"; do cp "benchmarks/_prof/libelfhip_$v.so" elf_amd/libelfhip.so; t $v; done
