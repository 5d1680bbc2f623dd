"""One findMove step against known answers derived by hand from the reference's source (tests/golden/gen_uct.py: a
numpy float32 / float64 restatement of tree_search_base.h:175-201 and tree_search_node.h:218-244,350-364,385-421 that
calls neither the oracle nor the product).  The fixture holds cases on which a fused multiply-add, a float sqrt or
another summation order give a different answer.  CPU: the generator reproduces its own file and oracle/mcts.cpp
agrees with it; GPU: the search kernels' own descent code (select_one, through elfhip_test_find_move) agrees."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests import _oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def _load():
    with open(os.path.join(HERE, "golden", "uct_known_answers.json")) as f:
        return json.load(f)


def _f(u):
    return np.asarray(u, dtype=np.uint32).view(np.float32)


def _arrays(case):
    e = np.asarray(case["edges"], dtype=np.int64)
    return (_f(e[:, 0]).copy(), _f(e[:, 1]).copy(), e[:, 2].astype(np.int32), _f(e[:, 3]).copy())


def test_fixture_is_what_the_generator_derives():
    """the committed file is the generator's output for its seed, and its discriminating cases discriminate"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_uct", os.path.join(HERE, "golden", "gen_uct.py"))
    g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
    data = _load()
    assert data["counts"]["fma"] >= 10 and data["counts"]["sqrtf"] >= 10 and data["counts"]["order"] >= 10
    for case in data["cases"]:
        pr, rw, vis, vl = _arrays(case)
        node = dict(case["node"]); node["meanQ"] = float(_f(node["meanQ"])); node["parentQ"] = float(_f(node["parentQ"]))
        opt = dict(case["opt"]); opt["c_puct"] = float(_f(opt["c_puct"]))
        edges = [[float(pr[i]), float(rw[i]), int(vis[i]), float(vl[i])] for i in range(len(vis))]
        sc, best, mq = g.scores(node, edges, opt)
        assert best == case["expect"]["best"] and g.bits(mq) == case["expect"]["meanQ"]
        assert [g.bits(s) for s in sc] == case["expect"]["scores"]
        for mode in ("fma", "sqrtf"):
            if mode in case["tags"]:
                assert g.scores(node, edges, opt, mode)[1] != best
        if "order" in case["tags"]:
            assert g.bits(g.scores(node, edges, opt, "reverse")[2]) != case["expect"]["meanQ"]


def test_oracle_find_move_matches_known_answers():
    L = _oracle.oracle()
    for ci, case in enumerate(_load()["cases"]):
        pr, rw, vis, vl = _arrays(case)
        n = len(vis)
        # keys 0..n-1 inserted ascending iterate in the library's order `it`; edge j of the case takes key it[j]
        keys = np.arange(n, dtype=np.uint16)
        it = np.zeros(n, dtype=np.uint16)
        L.omcts_std_umap_order(n, keys, it)
        pos = np.empty(n, dtype=np.int64); pos[it.astype(np.int64)] = np.arange(n)     # key -> edge index
        p2, r2, v2, l2 = (np.ascontiguousarray(a[pos]) for a in (pr, rw, vis, vl))
        nd, op = case["node"], case["opt"]
        best = np.zeros(1, dtype=np.int32); mq = np.zeros(1, dtype=np.float32)
        iopt = np.array([op["use_prior"], op["unexplored_q_zero"], op["root_unexplored_q_zero"], 0], dtype=np.int32)
        L.omcts_test_find_move(n, keys, p2, r2, v2, l2, nd["flip"], float(_f(nd["meanQ"])), float(_f(nd["parentQ"])), nd["numVisits"],
                               nd["depth"], iopt, float(_f(op["c_puct"])), best, mq)
        assert int(pos[int(best[0])]) == case["expect"]["best"], (ci, case["tags"])
        assert int(mq.view(np.uint32)[0]) == case["expect"]["meanQ"], (ci, case["tags"])


@pytest.mark.gpu
def test_kernel_find_move_matches_known_answers():
    import torch
    assert torch.cuda.is_available()
    from elf_amd import _lib
    L = _lib.lib()
    bad = []
    for ci, case in enumerate(_load()["cases"]):
        pr, rw, vis, vl = _arrays(case)
        nd, op = case["node"], case["opt"]
        ruqz = op["root_unexplored_q_zero"] if nd["depth"] == 0 else 0       # the hook's node is the root (depth 0)
        opt = _lib.TSOptions(1, 1, 0, op["use_prior"], float(_f(op["c_puct"])), op["unexplored_q_zero"], ruqz, 1, 0, 0, 0)
        best = C.c_int(-1); mq = C.c_float(0)
        _lib.check(L.elfhip_test_find_move(len(vis), pr.ctypes.data, rw.ctypes.data, vis.ctypes.data, vl.ctypes.data, nd["flip"],
                                           float(_f(nd["meanQ"])), float(_f(nd["parentQ"])), nd["numVisits"], C.byref(opt), 0,
                                           C.byref(best), C.byref(mq)))
        got_mq = int(np.asarray(mq.value, dtype=np.float32).view(np.uint32))
        if best.value != case["expect"]["best"] or got_mq != case["expect"]["meanQ"]:
            bad.append((ci, case["tags"], best.value, case["expect"]["best"], got_mq, case["expect"]["meanQ"]))
    assert not bad, bad[:5]
