#!/usr/bin/env python3
"""bench.py - headline benchmark: Go 19x19 random-legal-move playouts, 65,536 concurrent boards per
GPU (BASELINE.json configs[1]: pure dynamics - play / capture / Zobrist / ko / superko / legal mask).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE launch of the persistent playout kernel that advances every board PLIES plies
(boards stay resident in HBM; state lives in registers inside the launch).  Games shard across
GPUs with no data-path collective (weak scaling: 65,536 boards per GPU).  Rank 0 prints one JSON
line with the whole-job moves/s, the kernel's roofline record and a CPU baseline (the oracle timed
on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BOARDS_PER_GPU = 65536
PLIES = 64                 # plies per launch ("step")
ALGO_BYTES_PER_MOVE = 280  # state 112 R + 112 W + mask 48 + hash append 8
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec
SEED = 0x5EED0001


def cpu_baseline(threads=16, boards_per_thread=256, plies=8192):
    """Oracle (oracle/go.c, kind 'port') on the host cores; bounded sample of the same workload."""
    from concurrent.futures import ThreadPoolExecutor
    from tests import _oracle
    _oracle.oracle()
    threads = max(1, min(threads, os.cpu_count() or 1))
    slabs = [_oracle.GoBoards(boards_per_thread) for _ in range(threads)]

    def run(i):
        return slabs[i].playout(plies, seed=SEED, board_id0=i * boards_per_thread)  # ctypes drops the GIL

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        moves = sum(ex.map(run, range(threads)))
    dt = time.perf_counter() - t0
    return {"value": moves / dt, "unit": "moves/s", "cores": threads, "kind": "port",
            "sample": "%d threads x %d boards x %d plies of the same random-legal playout (oracle/go.c), %.1f s"
                      % (threads, boards_per_thread, plies, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)   # 64 launches x 4.19e6 moves = 2^28 moves
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--boards", type=int, default=BOARDS_PER_GPU)
    ap.add_argument("--plies", type=int, default=PLIES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from elf_amd.boards import BoardBatch, GO19

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("ELF_BENCH_ONE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL ("nccl") over xGMI, one rank per GPU.  ELF_BENCH_BACKEND=gloo + ELF_BENCH_ONE_GPU=1 rehearse the
        # multi-process path with every rank on GPU 0 (a one-GPU box cannot form an RCCL communicator).
        backend = os.environ.get("ELF_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    stream = torch.cuda.Stream()
    bb = BoardBatch(GO19, args.boards, device=local)
    bb.set_stream(stream)
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    first_id = rank * args.boards  # distinct games on every rank

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            bb.playout(args.plies, seed=SEED, first_board_id=first_id, counter=counter)
    barrier()
    counter.zero_()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for a, b in evs:
            a.record(stream)
            bb.playout(args.plies, seed=SEED, first_board_id=first_id, counter=counter)
            b.record(stream)
    barrier()
    dt = time.perf_counter() - t0
    moves = int(counter.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in evs) / max(1, len(evs))

    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    tot = torch.tensor([moves], dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt_all, moves_all = float(tmax.item()), int(tot.item())

    if rank == 0:
        moves_per_launch = moves / max(1, args.steps)
        achieved = ALGO_BYTES_PER_MOVE * moves_per_launch / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("go_playout_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "go19_board_moves_per_sec", "value": moves_all / dt_all, "unit": "moves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_all / max(1, args.steps) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "Go 19x19 random-legal-move playouts, %d concurrent boards per GPU, %d plies per launch "
                                   "(BASELINE.json configs[1]; build-defined Go rules, parity unpinned vs ELF)"
                                   % (args.boards, args.plies),
                       "boards_per_gpu": args.boards, "plies_per_step": args.plies, "parallelism": "games sharded, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "go_playout_kernel", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_move": ALGO_BYTES_PER_MOVE},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
