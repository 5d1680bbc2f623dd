#!/usr/bin/env python3
"""Secondary benchmark (BASELINE.json configs[2]): American checkers batched self-play, 16,384 games,
uniform-policy MCTS, 256 rollouts/move on one MI355X.  Prints one JSON line (rollouts/s, roofline
with the measured depth / branching per SURVEY.md 8(d), CPU baseline = oracle/mcts.cpp).
Not the driver's bench.py; run by hand:  python benchmarks/bench_mcts.py [--games N] [--moves M]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_baseline(threads=16, games_per_thread=64, moves=40, batch=8):
    from concurrent.futures import ThreadPoolExecutor
    from tests._oracle import AmGame, OracleMcts

    def run(i):
        tot = 0
        for g_ in range(games_per_thread):
            g = AmGame(); m = OracleMcts(num_rollouts=256, rollouts_per_batch=batch, virtual_loss=1, c_puct=1.5, policy_kind=0)
            for mv in range(moves):
                if g.terminated():
                    break
                r = m.run(g); g.forward(r["best"]); m.advance(r["best"])
            tot += m.stats()["rollouts"]
        return tot

    threads = max(1, min(threads, os.cpu_count() or 1))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        rollouts = sum(ex.map(run, range(threads)))
    dt = time.perf_counter() - t0
    return {"value": rollouts / dt, "unit": "rollouts/s", "cores": threads, "kind": "port",
            "sample": "%d threads x %d games x %d moves x 256 rollouts (oracle/mcts.cpp, libstdc++ containers like the reference), %.1f s"
                      % (threads, games_per_thread, moves, dt)}


def measure(games=16384, moves=12, warmup=2, batch=8):
    """the GPU half: one JSON-able dict (bench.py also calls this for the driver line's `extras`)"""
    import types
    a = types.SimpleNamespace(games=games, moves=moves, warmup=warmup, batch=batch)
    import torch
    from elf_amd.boards import BoardBatch, AMERICAN
    from elf_amd.mcts import TreeSearchBatch
    n = a.games
    bb = BoardBatch(AMERICAN, n)
    ts = TreeSearchBatch(AMERICAN, n, num_rollouts_per_thread=256, num_rollouts_per_batch=a.batch, virtual_loss=1, c_puct=1.5)
    best = torch.empty(n, dtype=torch.int16, device="cuda")

    def move():
        ts.set_roots(bb)
        ts.run_uniform()
        ts.L.elfhip_mcts_result(ts.h, best.data_ptr(), None, None, None, None, None, None, None, None, 0, None)
        bb.forward(best)
        ts.treeAdvance(best)

    for _ in range(a.warmup):
        move()
    torch.cuda.synchronize()
    s0, _ = ts.stats()
    evs = []
    t0 = time.perf_counter()
    for _ in range(a.moves):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts.set_roots(bb)
        e0.record(); ts.run_uniform(); e1.record()
        ts.L.elfhip_mcts_result(ts.h, best.data_ptr(), None, None, None, None, None, None, None, None, 0, None)
        bb.forward(best)
        ts.treeAdvance(best)
        evs.append((e0, e1))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1, flags = ts.stats()
    roll = s1["rollouts"] - s0["rollouts"]
    D = (s1["depth_sum"] - s0["depth_sum"]) / max(1, roll)
    newn = max(1, s1["evals"] - s0["evals"])
    b = (s1["edge_sum"] - s0["edge_sum"]) / newn
    S = 48
    bytes_per_rollout = D * (32 + 20 * b + 24) + (32 + S + 20 * b)
    kern_ms = sum(x.elapsed_time(y) for x, y in evs) / len(evs)
    roll_per_launch = roll / a.moves
    achieved = bytes_per_rollout * roll_per_launch / (kern_ms * 1e-3) / 1e9
    out = {"metric": "checkers_mcts_rollouts_per_sec", "value": roll / dt, "unit": "rollouts/s", "n_gpus": 1,
           "moves": a.moves, "ms_per_move": dt / a.moves * 1e3, "dtype": "f32/u64", "data": "synthetic",
           "config": {"workload": "American checkers batched self-play, %d games, uniform-policy MCTS, 256 rollouts/move, batch %d, vloss 1, c_puct 1.5 (BASELINE.json configs[2])" % (n, a.batch)},
           "parked_trees": int((flags != 0).sum().item()),
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                        "traffic": None, "kernel": "quad_search_kernel<AmQuad> (four trees per wave)" if os.environ.get("ELFHIP_MCTS_QUAD", "1") != "0" else "mcts_search_kernel<0>", "kernel_ms": kern_ms,
                        "mean_depth": D, "mean_branching": b, "algorithmic_bytes_per_rollout": bytes_per_rollout}}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=16384)
    ap.add_argument("--moves", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    out = measure(a.games, a.moves, a.warmup, a.batch)
    if not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(batch=a.batch)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
