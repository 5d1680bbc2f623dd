#!/bin/bash
# Go playout rate against the number of concurrent boards.  Up to one wave per SIMD (65,536 boards on the MI355X) the
# plain kernel runs; above that the slab launches the kernel built for two waves per SIMD (go.hip, go_playout_body).
# Variant libraries named on the command line (benchmarks/_build_prof_lib.sh) are timed after the product.
mkdir -p gpurun_out
run() { for b in 65536 131072 262144; do
  timeout -k 10 200 python bench.py --boards $b --no-cpu-baseline --no-extras --no-nn-rollouts > gpurun_out/gb_$b.json 2>>gpurun_out/gb_err.log || exit 1
  python -c "import json;d=json.loads(open('gpurun_out/gb_$b.json').read().strip().splitlines()[-1]);print('$1',$b,'%.3e'%d['value'],d['ms_per_step'])"
done; }
run base
for v in "$@"; do cp "benchmarks/_prof/libelfhip_$v.so" elf_amd/libelfhip.so; run $v; done
