# A/B of the skip-stream extension (ELFHIP_NET_SPLIT_SKIP=0: round-1 arithmetic); timing only
for d in 1 0; do
ELFHIP_NET_SPLIT_SKIP=$d timeout -k 10 200 python benchmarks/bench_net.py --shape go --batch 32768 --steps 3 > gpurun_out/net_go_split$d.json 2>&1 || exit 1
ELFHIP_NET_SPLIT_SKIP=$d timeout -k 10 200 python benchmarks/bench_net.py --shape checkers --batch 65536 --steps 5 > gpurun_out/net_ck_split$d.json 2>&1 || exit 1
done
