#!/usr/bin/env python3
"""Random-legal playouts of the three reference games (SURVEY.md 8(d) "same harness for checkers, where the
oracle exists"): 65,536 boards, 64 plies per launch, moves/s + HBM roofline fraction with the algorithmic
bytes per move of SURVEY 8(d) (American 152 B, Russian 104 B, Ugolki 120 B).  CPU column = the oracle port on
the host threads.  One JSON line per game.   python benchmarks/bench_boards.py [--boards N] [--steps K]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GAMES = [("american", 0, 152, 6, "am_"), ("russian", 1, 104, 5, "ru_"), ("ugolki", 2, 120, 4, "ug_")]
SEED = 0x5EED0001


def cpu(prefix, words, threads=16, boards=4096, plies=256):
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from tests import _oracle
    L = _oracle.oracle()
    threads = max(1, min(threads, os.cpu_count() or 1))
    slabs = []
    for t in range(threads):
        s = np.zeros((boards, words), dtype=np.uint64)
        for b in range(boards):
            getattr(L, prefix + "clear")(s[b])
        slabs.append(s)

    def run(i):
        return getattr(L, prefix + "playout")(slabs[i].reshape(-1), boards, i * boards, 0, plies, SEED)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        moves = sum(ex.map(run, range(threads)))
    dt = time.perf_counter() - t0
    return {"value": moves / dt, "unit": "moves/s", "cores": threads, "kind": "port",
            "sample": "%d threads x %d boards x %d plies (oracle), %.1f s" % (threads, boards, plies, dt)}


def cpu_reference(threads=16, boards=256, plies=4096):
    """The reference's own C++ (GameBoard.cc compiled unmodified into oracle/_ref/libref.so) driven through the
    same playout: American checkers only (its GetValidMovesBinary formats a string and searches a std::map per
    move, which is where its time goes).  None when oracle/_ref was not built."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from tests import _oracle
    R, L = _oracle.ref(), _oracle.oracle()
    if R is None or not hasattr(R, "ref_am_playout"):
        return None
    threads = max(1, min(threads, os.cpu_count() or 1))
    slabs = []
    for t in range(threads):
        s = np.zeros((boards, 6), dtype=np.uint64)
        for b in range(boards):
            L.am_clear(s[b])
        slabs.append(s)

    def run(i):
        return R.ref_am_playout(slabs[i].reshape(-1), boards, i * boards, 0, plies, SEED)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        moves = sum(ex.map(run, range(threads)))
    dt = time.perf_counter() - t0
    return {"value": moves / dt, "unit": "moves/s", "cores": threads, "kind": "reference",
            "sample": "%d threads x %d boards x %d plies through the compiled reference GameBoard.cc, %.1f s" % (threads, boards, plies, dt)}


def measure(boards=65536, plies=64, steps=20, warmup=3, no_cpu=True):
    """one dict per game (bench.py calls this, GPU half only, for the driver line's `extras`)"""
    import types
    a = types.SimpleNamespace(boards=boards, plies=plies, steps=steps, warmup=warmup, no_cpu=no_cpu)
    import torch
    from elf_amd.boards import BoardBatch
    outs = []
    for name, gid, bytes_per_move, words, prefix in GAMES:
        bb = BoardBatch(gid, a.boards)
        counter = torch.zeros(1, dtype=torch.int64, device="cuda")
        for _ in range(a.warmup):
            bb.playout(a.plies, seed=SEED, counter=counter)
        torch.cuda.synchronize()
        counter.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(a.steps):
            bb.playout(a.plies, seed=SEED, counter=counter)
        e1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        moves = int(counter.item())
        kern_ms = e0.elapsed_time(e1) / a.steps
        ach = bytes_per_move * (moves / a.steps) / (kern_ms * 1e-3) / 1e9
        out = {"metric": "%s_board_moves_per_sec" % name, "value": moves / dt, "unit": "moves/s", "n_gpus": 1, "steps": a.steps,
               "ms_per_step": dt / a.steps * 1e3, "dtype": "u64", "data": "synthetic",
               "config": {"workload": "%s random-legal playouts, %d boards, %d plies per launch" % (name, a.boards, a.plies)},
               "roofline": {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None,
                            "kernel": prefix + "playout_kernel", "kernel_ms": kern_ms, "algorithmic_bytes_per_move": bytes_per_move}}
        if not a.no_cpu:
            out["cpu_baseline"] = cpu(prefix, words)
            if name == "american":
                ref = cpu_reference()
                if ref is not None:
                    out["cpu_baseline_port"] = out["cpu_baseline"]
                    out["cpu_baseline"] = ref
        outs.append(out)
        bb.close()
    return outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=65536)
    ap.add_argument("--plies", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    for out in measure(a.boards, a.plies, a.steps, a.warmup, a.no_cpu):
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
