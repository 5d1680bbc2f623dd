"""Pins oracle/mcts.cpp against the known answers captured from the COMPILED reference tree search
(SURVEY.md Appendix C -> tests/golden/mcts_appendix_c.json)."""
import json
import os

import numpy as np
import pytest

from tests import _oracle
from tests._oracle import AmGame, OracleMcts


@pytest.fixture(scope="module")
def gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, "mcts_appendix_c.json")))


def test_libstdcxx_sort_and_umap_known_answers(gold):
    L = _oracle.oracle()
    order = np.zeros(170, dtype=np.uint16)
    L.omcts_std_sort_desc(170, np.full(170, 1.0 / 170, dtype=np.float32), order)
    assert list(order[:40]) == gold["sort_uniform_170_first40"]
    it = np.zeros(20, dtype=np.uint16)
    L.omcts_std_umap_order(20, np.arange(20, dtype=np.uint16), it)
    assert list(it) == gold["umap_keys_0_19_iteration"]
    it = np.zeros(7, dtype=np.uint16)
    L.omcts_std_umap_order(7, np.array(gold["start_insertion_order"], dtype=np.uint16), it)
    assert list(it) == gold["start_iteration_order"]


@pytest.mark.parametrize("batch", [1, 8])
def test_reference_known_answer_game(gold, batch):
    ref = gold["batch%d" % batch]
    g = AmGame()
    m = OracleMcts(num_rollouts=256, rollouts_per_batch=batch, virtual_loss=1, c_puct=1.5, policy_kind=0)
    actions, totals = [], []
    for mv in range(12):
        r = m.run(g)
        if mv < 3:
            assert [[a, v] for (a, v, _, _) in r["edges"]] == ref["move%d" % mv], mv
        actions.append(r["best"]); totals.append(r["total_visits"])
        assert g.forward(r["best"])
        m.advance(r["best"])
    assert actions == ref["first12_actions"]
    assert totals == ref["total_visits"]


def test_scripted_policy_is_on_exact_grid():
    L = _oracle.oracle()
    g = AmGame()
    pi = np.zeros(170, dtype=np.float32); v = np.zeros(1, dtype=np.float32)
    L.omcts_scripted_policy(g.board(), 7, pi, v)
    assert np.all(pi * 1024 == np.round(pi * 1024)) and pi.min() > 0 and pi.max() <= 1
    assert v[0] * 1024 == round(float(v[0]) * 1024) and -1 <= v[0] < 1
    assert len(np.unique(pi)) < 170  # ties exist, exercising the unstable-sort order


def test_scripted_search_is_deterministic_and_batch_shapes():
    outs = []
    for rep in range(2):
        g = AmGame()
        m = OracleMcts(num_rollouts=128, rollouts_per_batch=8, virtual_loss=3, c_puct=0.9, policy_kind=1, policy_seed=5)
        seq = []
        for mv in range(20):
            if g.terminated():
                break
            r = m.run(g)
            seq.append((r["best"], r["total_visits"], tuple(r["edges"])))
            g.forward(r["best"]); m.advance(r["best"])
        outs.append(seq)
    assert outs[0] == outs[1] and len(outs[0]) == 20
