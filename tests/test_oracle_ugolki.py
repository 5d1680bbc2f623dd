"""Pins oracle/ugolki.c against goldens from the compiled reference and, in the build container,
against the compiled reference live.  Bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import _oracle


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ugolki_ref.npz"))


def test_action_table(gold):
    L = _oracle.oracle()
    for a in range(417):
        b, f = C.c_uint64(), C.c_uint64()
        assert L.ug_action(a, C.byref(b), C.byref(f)) == 1
        assert (b.value, f.value) == (int(gold["action_bits"][a]), int(gold["action_flags"][a]))
    assert L.ug_action(417, C.byref(b), C.byref(f)) == 0


def test_replay_golden(gold):
    L = _oracle.oracle()
    st, mk, ac, ov, ob, gm, pr, pk = (gold[k] for k in ("state", "mask", "action", "over", "obs", "game", "probe", "probe_ok"))
    s0 = np.zeros(4, dtype=np.uint64); L.ug_clear(s0)
    assert np.array_equal(s0, st[0])
    m = np.zeros(7, dtype=np.uint64); o = np.zeros(64, dtype=np.int8)
    jump_mode = wins = 0
    for t in range(len(st)):
        s = st[t].copy()
        L.ug_legal_mask(s, m)
        assert np.array_equal(m, mk[t]), t
        assert L.ug_is_over(s) == ov[t]
        for p in (0, 1):
            L.ug_observation(s, p, o)
            assert np.array_equal(o, ob[t, p]), (t, p)
        for a, ok in zip(pr[t], pk[t]):
            assert L.ug_try_play(s, int(a)) == ok, (t, a)
        jump_mode += int(s[2] != 0)
        wins += int(ov[t])
        if ac[t] >= 0:
            L.ug_play(s, int(ac[t]))
            if t + 1 < len(st) and gm[t + 1] == gm[t]:
                assert np.array_equal(s, st[t + 1]), t
    assert jump_mode > 300 and wins >= 10


def test_tryplay_equals_mask_on_reachable_states(gold):
    """GameState::forward gates on TryPlay while the policy mask comes from get_legal_moves; on
    reachable states they agree for every action (this is what lets the HIP step use the mask)."""
    L = _oracle.oracle()
    st = gold["state"]
    m = np.zeros(7, dtype=np.uint64)
    for t in range(0, len(st), 7):
        s = st[t].copy()
        L.ug_legal_mask(s, m)
        legal = set(_oracle.mask_bits(m, 418))
        for a in range(417):
            assert L.ug_try_play(s, a) == (1 if a in legal else 0), (t, a)


def test_features_evaluate_and_wrapper(gold):
    L = _oracle.oracle()
    s = np.zeros(4, dtype=np.uint64); L.ug_clear(s)
    f = np.zeros(16 * 64, dtype=np.float32); L.ug_features(s, f)
    f = f.reshape(16, 8, 8)
    assert f[0].sum() == 9 and f[0][5:, 5:].sum() == 9      # black starts in the high corner, seen unrotated
    assert f[1].sum() == 9 and f[1][5:, 5:].sum() == 9      # white's own view is rotated by 180 degrees
    assert f[2].min() == 1 and f[3:].max() == 0
    assert L.ug_evaluate(s) == 0.0 and not L.ug_terminated(s)
    assert L.ug_forward(s, 416) == 0 and L.ug_forward(s, 417) == 0
    # terminal states of the finished golden games
    st, ov = gold["state"], gold["over"]
    vals = set()
    for t in np.nonzero(ov)[0]:
        s = st[t].copy()
        bw, ww = (int(s[3]) >> 1) & 7, (int(s[3]) >> 4) & 7
        v = L.ug_evaluate(s)
        assert v == (-1.0 if (bw > 0 and ww > 0) else (1.0 if bw == 2 else -1.0))
        assert L.ug_forward(s, 0) == 0
        vals.add(v)
    assert vals == {1.0, -1.0}


@pytest.mark.skipif(_oracle.ref_ugolki() is None, reason="compiled reference only exists in the build container")
def test_live_against_compiled_reference():
    L, R = _oracle.oracle(), _oracle.ref_ugolki()
    rng = np.random.default_rng(321)
    plies = 0
    for g in range(30):
        s = np.zeros(4, dtype=np.uint64); r = np.zeros(4, dtype=np.uint64)
        L.ug_clear(s); R.ref_ug_clear(r)
        for _ in range(300):
            assert np.array_equal(s, r)
            v = np.zeros(418, dtype=np.int32); R.ref_ug_valid_moves(r, v)
            m = np.zeros(7, dtype=np.uint64); L.ug_legal_mask(s, m)
            legal = list(np.nonzero(v)[0])
            assert _oracle.mask_bits(m, 418) == legal
            for a in rng.integers(0, 417, size=6):
                assert L.ug_try_play(s, int(a)) == R.ref_ug_tryplay(r, int(a))
            if R.ref_ug_is_over(r) or not legal:
                break
            a = int(rng.choice(legal))
            L.ug_play(s, a); R.ref_ug_play(r, a)
            plies += 1
    assert plies > 5000
