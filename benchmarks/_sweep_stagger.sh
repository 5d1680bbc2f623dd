#!/bin/bash
mkdir -p gpurun_out
for cfg in "4 -1" "4 40000" "8 20000" "8 10000" "16 10000" "16 5000" "2 80000" "4 0"; do
  set -- $cfg
  if [ "$2" = "-1" ]; then unset ELFHIP_NET_STAGGER; else export ELFHIP_NET_STAGGER=$2; fi
  export ELFHIP_NET_STAGGER_PHASES=$1
  timeout -k 10 120 python benchmarks/bench_net.py --shape go --batch 4096 --steps 6 > gpurun_out/stg.json 2>>gpurun_out/stg_err.log || exit 1
  python -c "import json;d=json.load(open('gpurun_out/stg.json'));print('phases $1 stagger $2 ->',round(d['ms_per_forward'],2),'ms')"
done
