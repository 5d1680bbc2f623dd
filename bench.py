#!/usr/bin/env python3
"""bench.py - headline benchmark: Go 19x19 random-legal-move playouts, 65,536 concurrent boards per
GPU (BASELINE.json configs[1]: pure dynamics - play / capture / Zobrist / ko / superko / legal mask).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE launch of the persistent playout kernel that advances every board PLIES plies
(boards stay resident in HBM; state lives in registers inside the launch).  Games shard across
GPUs with no data-path collective (weak scaling: 65,536 boards per GPU).  Rank 0 prints one JSON
line with the whole-job moves/s, the kernel's roofline record and a CPU baseline (the oracle timed
on the host cores, N=1 only).  The same line carries, under "nn_rollouts", the other half of
BASELINE.json's metric measured in the same run: NN-evaluated MCTS rollouts/s on configs[3]
(4,096 Go games per GPU, 1,600 rollouts/move, 20x224 ResNet in bf16 MFMA) with its MFMA roofline.
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


def nn_rollouts_leg(games, steps, warmup, local, world, barrier):
    """Second half of BASELINE.json's metric: NN-evaluated MCTS rollouts/s on configs[3] (Go 19x19 self-play,
    4,096 games per GPU, 1,600 rollouts/move, 8 leaves per batch, virtual loss 1, 20-block x 224 ResNet in bf16
    MFMA, seeded random weights).  One step = one search batch over all games: select (descents + Go moves +
    18 planes) -> network forward on games*8 leaves -> backup.  Only leaves that went through the network count."""
    import torch
    import torch.distributed as dist
    from elf_amd.boards import BoardBatch, GO19
    from elf_amd.mcts import TreeSearchBatch
    from elf_amd.net import PolicyValueNet, random_state_dict
    n, B, R, blocks, dim = games, 8, 1600, 20, 224
    bb = BoardBatch(GO19, n, device=local)
    bb.playout(40, seed=SEED)                    # synthetic openings: 40 random-legal plies
    bb.import_states(bb.export_states())         # clean game record for the planes / superko set
    max_nodes = R + R // 2 + 64
    ts = TreeSearchBatch(GO19, n, device=local, num_rollouts_per_thread=R, num_rollouts_per_batch=B, virtual_loss=1, c_puct=1.5,
                         max_nodes=max_nodes, max_edges=max_nodes * 300)
    cfg = dict(planes=18, board=19, dim=dim, num_block=blocks, num_action=362)
    net = PolicyValueNet(max_batch=n * B, device=local, **cfg)
    net.load_state_dict(random_state_dict(seed=0, **cfg))
    assert ts.set_roots(bb).cpu().numpy().all()
    feat = torch.empty((n * B, 18, 19, 19), dtype=torch.float32, device="cuda")
    need = torch.empty((n, B), dtype=torch.uint8, device="cuda")
    pi = torch.empty((n * B, 362), dtype=torch.float32, device="cuda")
    v = torch.empty((n * B, 1), dtype=torch.float32, device="cuda")
    L = ts.L

    def step(ev=None):
        if ev: ev[0].record()
        L.elfhip_mcts_select(ts.h, None, feat.data_ptr(), need.data_ptr())
        if ev: ev[1].record()
        L.elfhip_net_forward(net.h, feat.data_ptr(), n * B, pi.data_ptr(), None, v.data_ptr())
        if ev: ev[2].record()
        L.elfhip_mcts_backup(ts.h, pi.data_ptr(), v.data_ptr())
        if ev: ev[3].record()

    for _ in range(warmup):
        step()
    barrier()
    s0, _ = ts.stats()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        step(evs[k])
    barrier()
    dt = time.perf_counter() - t0
    s1, flags = ts.stats()
    evals = s1["evals"] - s0["evals"]
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    tot = torch.tensor([evals], dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    sel = sum(e[0].elapsed_time(e[1]) for e in evs) / steps
    fwd = sum(e[1].elapsed_time(e[2]) for e in evs) / steps
    bak = sum(e[2].elapsed_time(e[3]) for e in evs) / steps
    macs = 361 * (9 * 18 * dim + blocks * 2 * 9 * dim * dim + 2 * dim) + 361 * 362 + 361 * 256 + 256
    tf = 2.0 * macs * n * B / (fwd * 1e-3) / 1e12
    return {"metric": "go_nn_evaluated_mcts_rollouts_per_sec", "value": int(tot.item()) / float(tmax.item()), "unit": "rollouts/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": float(tmax.item()) / steps * 1e3, "dtype": "bf16",
            "data": "synthetic", "scaling": "weak",
            "config": {"workload": "Go 19x19 self-play, %d games per GPU, %d rollouts/move, %d leaves/batch, vloss 1, %d-block x %d "
                                   "ResNet bf16 (BASELINE.json configs[3]); one step = one search batch over all games"
                                   % (n, R, B, blocks, dim)},
            "time_split_ms": {"select_expand_planes": sel, "network": fwd, "backup": bak},
            "parked_trees": int((flags != 0).sum().item()),
            "roofline": {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0, "traffic": None,
                         "kernel": "conv3x3_board_kernel<224,19,1> (41 launches per forward)", "kernel_ms": fwd,
                         "gflop_per_position": 2.0 * macs / 1e9, "positions_per_sec": n * B / (fwd * 1e-3)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)   # 64 launches x 4.19e6 moves = 2^28 moves
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--boards", type=int, default=BOARDS_PER_GPU)
    ap.add_argument("--plies", type=int, default=PLIES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-nn-rollouts", action="store_true", help="skip the configs[3] leg (NN-evaluated MCTS rollouts/s)")
    ap.add_argument("--nn-games", type=int, default=4096)
    ap.add_argument("--nn-steps", type=int, default=6)
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

    # free the playout slab before the self-play leg (it needs ~160 GB of tree arenas and activations)
    del bb
    nn = None
    if not args.no_nn_rollouts:
        try:
            nn = nn_rollouts_leg(args.nn_games, args.nn_steps, 2, local, world, barrier)
        except Exception as e:   # the headline line must not be lost to the secondary leg
            nn = {"metric": "go_nn_evaluated_mcts_rollouts_per_sec", "value": None, "error": "%s: %s" % (type(e).__name__, e)}
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
        if nn is not None:
            out["nn_rollouts"] = nn
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
