#!/bin/bash
mkdir -p gpurun_out
for w in 256 192 128 64; do
  export ELFHIP_NET_PERSIST=$w
  timeout -k 10 120 python benchmarks/bench_net.py --shape go --batch 4096 --steps 4 > gpurun_out/pw.json 2>>gpurun_out/pw_err.log || exit 1
  python -c "import json;d=json.load(open('gpurun_out/pw.json'));print('workgroups $w ->',round(d['ms_per_forward'],2),'ms; x wgs/256 =',round(d['ms_per_forward']*$w/256,2))"
done
