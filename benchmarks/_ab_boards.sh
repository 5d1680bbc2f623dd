#!/bin/bash
# same-box A/B of the checkers-family playout kernels: tests on the product library, then bench product / variants
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_american.py tests/test_gpu_russian.py tests/test_gpu_ugolki.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/b_tests.log 2>&1; tail -n 1 gpurun_out/b_tests.log
t() { timeout -k 10 200 python benchmarks/bench_boards.py --no-cpu > gpurun_out/b.json 2>>gpurun_out/b_err.log || exit 1; python -c "
import json
for line in open('gpurun_out/b.json'):
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line); print('$1', d['metric'], '%.3e'%d['value'], round(d['roofline']['frac'],3))"; }
t base
for v in "$@"; do cp "benchmarks/_prof/libelfhip_$v.so" elf_amd/libelfhip.so; t $v; done
