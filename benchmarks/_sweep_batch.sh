#!/bin/bash
# Go forward time per board at several batch sizes (does a batch whose activations fit the 256-MB Infinity Cache run faster per board?)
mkdir -p gpurun_out
for b in 256 512 1024 2048 4096 8192; do
  timeout -k 10 120 python benchmarks/bench_net.py --shape go --batch $b --steps 8 > gpurun_out/sb_$b.json 2>>gpurun_out/sb_err.log || exit 1
  python -c "import json;d=json.load(open('gpurun_out/sb_$b.json'));print($b,round(d['ms_per_forward'],3),'ms',round(d['ms_per_forward']*1000/$b,3),'us/board',round(d['roofline']['achieved'],1),'TF')"
done
