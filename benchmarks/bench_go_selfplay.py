#!/usr/bin/env python3
"""Secondary benchmark (BASELINE.json configs[3]): Go 19x19 self-play with the 20-block x 224 ResNet,
4,096 games per GPU, 1,600 rollouts/move, 8 leaves per batch, virtual loss 1, bf16 MFMA evaluation.
One "step" = one batch of the tree search over all games: select (descents + lane-per-leaf Go moves
+ 18 planes) -> network forward on games*8 leaves -> backup.  Prints one JSON line with
NN-evaluated rollouts/s (only leaves that really went through the network count),
the time split between search kernels and network, and the network's MFMA roofline.
Random-init weights (seeded), synthetic openings.  Not the driver's bench.py; run by hand:
    python benchmarks/bench_go_selfplay.py [--games N] [--steps K] [--blocks 20 --dim 224]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--rollouts", type=int, default=1600)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--dim", type=int, default=224)
    ap.add_argument("--open-plies", type=int, default=40, help="random-legal plies played before the search starts")
    ap.add_argument("--max-nodes", type=int, default=0)
    ap.add_argument("--max-edges", type=int, default=0)
    a = ap.parse_args()
    import numpy as np
    import torch
    from elf_amd.boards import BoardBatch, GO19
    from elf_amd.mcts import TreeSearchBatch
    from elf_amd.net import PolicyValueNet, random_state_dict
    n, B = a.games, a.batch
    bb = BoardBatch(GO19, n)
    bb.playout(a.open_plies, seed=0x5EED0001)
    # the playout kernel keeps no stone history; re-import so that search + planes start from a clean game record
    bb.import_states(bb.export_states())
    max_nodes = a.max_nodes or (a.rollouts + a.rollouts // 2 + 64)
    max_edges = a.max_edges or max_nodes * 300
    ts = TreeSearchBatch(GO19, n, num_rollouts_per_thread=a.rollouts, num_rollouts_per_batch=B, virtual_loss=1, c_puct=1.5,
                         max_nodes=max_nodes, max_edges=max_edges)
    cfg = dict(planes=18, board=19, dim=a.dim, num_block=a.blocks, num_action=362)
    net = PolicyValueNet(max_batch=n * B, **cfg)
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

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    s0, _ = ts.stats()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(a.steps)]
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(evs[k])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1, flags = ts.stats()
    sel = sum(e[0].elapsed_time(e[1]) for e in evs) / a.steps
    fwd = sum(e[1].elapsed_time(e[2]) for e in evs) / a.steps
    bak = sum(e[2].elapsed_time(e[3]) for e in evs) / a.steps
    evals = s1["evals"] - s0["evals"]
    roll = s1["rollouts"] - s0["rollouts"]
    macs = 361 * (9 * 18 * a.dim + a.blocks * 2 * 9 * a.dim * a.dim + 2 * a.dim) + 361 * 362 + 361 * 256 + 256
    tf = 2.0 * macs * n * B / (fwd * 1e-3) / 1e12
    out = {"metric": "go_nn_evaluated_mcts_rollouts_per_sec", "value": evals / dt, "unit": "rollouts/s", "n_gpus": 1,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "Go 19x19 self-play, %d games, %d rollouts/move, %d leaves/batch, vloss 1, %d-block x %d ResNet "
                                  "(BASELINE.json configs[3]); one step = one search batch over all games" % (n, a.rollouts, B, a.blocks, a.dim),
                      "open_plies": a.open_plies, "max_nodes": max_nodes, "max_edges": max_edges},
           "rollouts_per_sec_all": roll / dt, "evaluated_fraction": evals / max(1, roll),
           "ms_select_expand_features": sel, "ms_network": fwd, "ms_backup": bak,
           "mean_depth": (s1["depth_sum"] - s0["depth_sum"]) / max(1, roll),
           "mean_edges_per_new_node": (s1["edge_sum"] - s0["edge_sum"]) / max(1, evals),
           "parked_trees": int((flags != 0).sum().item()),
           "roofline": {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0, "traffic": None,
                        "kernel": "conv3x3_kernel (network forward, all layers)", "positions_per_sec": n * B / (fwd * 1e-3)}}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
