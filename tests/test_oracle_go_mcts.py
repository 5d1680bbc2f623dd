"""The oracle tree search over Go (oracle/mcts.cpp, GoTraits): CPU-only sanity of the template
instantiation - visit accounting, determinism, persistent tree, pass/terminal handling.
Go has no reference in ELF (parity unpinned); the SEARCH code is the one pinned for checkers."""
import numpy as np

from tests import _oracle
from tests._oracle import GoBoards, OracleMcts


def test_uniform_search_accounting_and_determinism():
    g = GoBoards(1)
    for a in (3 * 19 + 3, 15 * 19 + 15, 3 * 19 + 15):
        assert g.forward(a)
    res = []
    for _ in range(2):
        o = OracleMcts(num_rollouts=64, rollouts_per_batch=4, virtual_loss=1, c_puct=1.5, game="go")
        r = o.run(g.ptr(0))
        res.append(r)
        st = o.stats()
        assert st["rollouts"] == 64
        assert len(r["edges"]) == 361 - 3 + 1          # every empty point + pass
        assert r["total_visits"] <= 64 and r["total_visits"] >= 64 - 4
        assert sum(v for (_, v, _, _) in r["edges"]) == r["total_visits"]
    assert res[0] == res[1]


def test_scripted_search_persistent_tree_and_terminal():
    g = GoBoards(1)
    o = OracleMcts(num_rollouts=48, rollouts_per_batch=3, virtual_loss=1, c_puct=1.0, policy_kind=1, policy_seed=7, game="go")
    kept = 0
    for mv in range(6):
        r = o.run(g.ptr(0))
        best = r["best"]
        vis = dict((a, v) for (a, v, _, _) in r["edges"])
        assert vis[best] == max(vis.values())
        if mv > 0:
            kept += r["total_visits"] - 48      # visits carried over by treeAdvance
        assert g.forward(best)
        o.advance(best)
    assert kept > 0
    # two passes end the game: the root is terminal, search returns no edges
    g2 = GoBoards(1)
    assert g2.forward(361) and g2.forward(361) and g2.terminated()
    o2 = OracleMcts(num_rollouts=8, rollouts_per_batch=2, game="go")
    r = o2.run(g2.ptr(0))
    assert r["edges"] == [] and r["best"] == -1 and r["root_value"] in (1.0, -1.0)
