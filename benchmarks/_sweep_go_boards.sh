#!/bin/bash
# Go playout rate against the number of concurrent boards (waves per SIMD): how much a second wave per SIMD is worth.
# The product launches 64-thread blocks (33 KB of LDS tables each: four blocks = four waves per CU whatever the board
# count); a variant library with 256-thread blocks lets more boards really become more waves per SIMD.
mkdir -p gpurun_out
run() { for b in 65536 131072 262144; do
  timeout -k 10 200 python bench.py --boards $b --no-cpu-baseline --no-extras --no-nn-rollouts > gpurun_out/gb_$b.json 2>>gpurun_out/gb_err.log || exit 1
  python -c "import json;d=json.loads(open('gpurun_out/gb_$b.json').read().strip().splitlines()[-1]);print('$1',$b,'%.3e'%d['value'],d['ms_per_step'])"
done; }
run base
for v in "$@"; do cp "benchmarks/_prof/libelfhip_$v.so" elf_amd/libelfhip.so; run $v; done
