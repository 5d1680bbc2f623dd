# two ranks of bench.py on ONE GPU over gloo (the multi-process path without an RCCL communicator), small sizes
export ELF_BENCH_BACKEND=gloo ELF_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 8 --warmup 2 --boards 8192 --nn-games 256 --nn-batches 6 > gpurun_out/rehearse2.json 2> gpurun_out/rehearse2.err
