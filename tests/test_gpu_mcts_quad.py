"""The four-trees-per-wave uniform-policy search (csrc/mcts_quad.h) against the one-tree-per-wave kernel it replaces
for run_uniform (ELFHIP_MCTS_QUAD=0 selects the latter): identical trees bit for bit - edges in iteration order, visit
counts, rewards, priors, chosen moves, search statistics - over several moves with persistent trees, for the three
checkers-family games, with a tree count that is not a multiple of four, and when arenas overflow (parked trees).
The oracle comparisons of test_gpu_mcts.py / test_gpu_mcts_games.py run on the new kernel by default.
The kernel keeps the first 32 steps of a trajectory in LDS and deeper ones in memory; the deeper path was verified by
running these files against a build with -DELFHIP_QUAD_TRAJ_LDS=2 (benchmarks/_build_prof_lib.sh): same results."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(game, n, moves, opts, quad, monkeypatch, start_plies=12):
    import torch
    from elf_amd.boards import BoardBatch
    from elf_amd.mcts import TreeSearchBatch
    monkeypatch.setenv("ELFHIP_MCTS_QUAD", "1" if quad else "0")
    bb = BoardBatch(game, n)
    bb.playout(start_plies, seed=1234, first_board_id=0)       # different positions per tree
    ts = TreeSearchBatch(game, n, **opts)
    out = []
    for _ in range(moves):
        ok = ts.set_roots(bb)
        ts.run_uniform()
        r = ts.result(edge_cap=96, with_policy=True)
        st, fl = ts.stats()
        out.append({k: v.cpu().numpy().copy() for k, v in r.items()} | {"flags": fl.cpu().numpy().copy(), "stats": dict(st), "ok": ok.cpu().numpy().copy()})
        act = r["best"].clone()
        act[fl != 0] = -1
        bb.forward(act)
        ts.treeAdvance(act)
    return out


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x["stats"] == y["stats"]
        for k in x:
            if k == "stats":
                continue
            assert np.array_equal(x[k], y[k], equal_nan=True), k


@pytest.mark.parametrize("game_name", ["AMERICAN", "RUSSIAN", "UGOLKI"])
def test_quad_kernel_builds_the_same_trees(game_name, monkeypatch):
    from elf_amd import boards
    game = getattr(boards, game_name)
    opts = dict(num_rollouts_per_thread=64, num_rollouts_per_batch=8, virtual_loss=1, c_puct=1.5)
    a = _run(game, 37, 5, opts, True, monkeypatch)
    b = _run(game, 37, 5, opts, False, monkeypatch)
    _same(a, b)
    assert a[-1]["stats"]["rollouts"] == 37 * 5 * 64
    assert not a[-1]["flags"].any()


def test_quad_kernel_options_and_batch_one(monkeypatch):
    from elf_amd.boards import AMERICAN
    for opts in (dict(num_rollouts_per_thread=40, num_rollouts_per_batch=1, virtual_loss=0, c_puct=0.8),
                 dict(num_rollouts_per_thread=50, num_rollouts_per_batch=5, virtual_loss=3, c_puct=2.5, unexplored_q_zero=True, persistent_tree=False),
                 dict(num_rollouts_per_thread=48, num_rollouts_per_batch=16, virtual_loss=1, c_puct=1.0, root_unexplored_q_zero=True, use_prior=False)):
        _same(_run(AMERICAN, 10, 4, opts, True, monkeypatch, start_plies=40), _run(AMERICAN, 10, 4, opts, False, monkeypatch, start_plies=40))


def test_quad_kernel_parks_trees_like_the_wave_kernel(monkeypatch):
    """node and edge arenas too small: both kernels park the same trees at the same point"""
    from elf_amd.boards import AMERICAN
    for extra in (dict(max_nodes=40, max_edges=40 * 14), dict(max_nodes=200, max_edges=150)):
        opts = dict(num_rollouts_per_thread=64, num_rollouts_per_batch=8, virtual_loss=1, c_puct=1.5, **extra)
        a = _run(AMERICAN, 9, 3, opts, True, monkeypatch)
        b = _run(AMERICAN, 9, 3, opts, False, monkeypatch)
        _same(a, b)
        assert a[-1]["flags"].any()
