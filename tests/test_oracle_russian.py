"""Pins oracle/russian.c against goldens from the compiled reference and, in the build container,
against the compiled reference live.  Bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import _oracle


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "russian_ref.npz"))


def test_action_table(gold):
    L = _oracle.oracle()
    for a in range(280):
        f, t = C.c_int(), C.c_int()
        L.ru_action(a, C.byref(f), C.byref(t))
        assert (f.value, t.value) == (int(gold["action_from"][a]), int(gold["action_to"][a]))


def test_replay_golden(gold):
    L = _oracle.oracle()
    st, mk, ac, ov, ob, gm = (gold[k] for k in ("state", "mask", "action", "over", "obs", "game"))
    s0 = np.zeros(5, dtype=np.uint64); L.ru_clear(s0)
    assert np.array_equal(s0, st[0])
    m = np.zeros(5, dtype=np.uint64); o = np.zeros(64, dtype=np.int8)
    kings = 0
    for t in range(len(st)):
        s = st[t].copy()
        L.ru_legal_mask(s, m)
        assert np.array_equal(m, mk[t]), t
        assert L.ru_is_over(s) == ov[t]
        for p in (0, 1):
            L.ru_observation(s, p, o)
            assert np.array_equal(o, ob[t, p]), (t, p)
        kings += int(s[1] != 0 or s[3] != 0)
        if ac[t] >= 0:
            L.ru_play(s, int(ac[t]))
            if t + 1 < len(st) and gm[t + 1] == gm[t]:
                assert np.array_equal(s, st[t + 1]), t
    assert kings > 100   # flying-king positions are covered


def test_features_and_wrapper():
    L = _oracle.oracle()
    s = np.zeros(5, dtype=np.uint64); L.ru_clear(s)
    f = np.zeros(6 * 64, dtype=np.float32); L.ru_features(s, f)
    f = f.reshape(6, 8, 8)
    # Q8: black to move, plane 0 ("our pawns") actually holds WHITE pawns (rows 0-2)
    assert f[0].sum() == 12 and f[0][:3].sum() == 12 and f[2].sum() == 12 and f[4].min() == 1 and f[5].max() == 0
    assert L.ru_evaluate(s) == 0.0 and not L.ru_terminated(s)
    m = np.zeros(5, dtype=np.uint64); L.ru_legal_mask(s, m)
    legal = _oracle.mask_bits(m, 280)
    assert len(legal) == 7
    assert L.ru_forward(s, legal[0]) == 1 and L.ru_forward(s, 279) == 0


@pytest.mark.skipif(_oracle.ref_russian() is None, reason="compiled reference only exists in the build container")
def test_live_against_compiled_reference():
    L, R = _oracle.oracle(), _oracle.ref_russian()
    rng = np.random.default_rng(123)
    plies = 0
    for g in range(150):
        s = np.zeros(5, dtype=np.uint64); r = np.zeros(5, dtype=np.uint64)
        L.ru_clear(s); R.ref_ru_clear(r)
        while True:
            assert np.array_equal(s, r)
            v = np.zeros(280, dtype=np.int32); R.ref_ru_valid_moves(r, v)
            m = np.zeros(5, dtype=np.uint64); L.ru_legal_mask(s, m)
            legal = list(np.nonzero(v)[0])
            assert _oracle.mask_bits(m, 280) == legal
            if not legal or ((int(s[4]) >> 8) & 0xFFFF) >= 250:
                break
            a = int(rng.choice(legal))
            L.ru_play(s, a); R.ref_ru_play(r, a)
            plies += 1
    assert plies > 5000
