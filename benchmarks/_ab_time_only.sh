#!/bin/bash
# forward time of variant libraries next to the product's on the same box (no tests: use _ab_libs.sh for variants that must stay correct)
mkdir -p gpurun_out
t() { timeout -k 10 120 python benchmarks/bench_net.py --shape go --batch 4096 --steps 6 > gpurun_out/t.json 2>>gpurun_out/t_err.log || exit 1; python -c "import json;d=json.load(open('gpurun_out/t.json'));print('$1',round(d['ms_per_forward'],2),'ms')"; }
t base; t base
for v in "$@"; do cp "benchmarks/_prof/libelfhip_$v.so" elf_amd/libelfhip.so; t $v; t $v; done
