"""Pins the CPU restatement (oracle/american.c) against
 (a) golden vectors generated from the compiled reference (tests/golden/american_ref.npz), and
 (b) the compiled reference itself when oracle/_ref/libref.so is present (build container only).
Bit-exact on every state word, legal mask, observation and continuation flag.
"""
import os

import numpy as np
import pytest

from tests import _oracle


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "american_ref.npz"))


def test_action_table_matches_reference(gold):
    L = _oracle.oracle()
    mv = np.array([L.am_action_move(a) for a in range(170)], dtype=np.int64)
    dr = np.array([L.am_action_dir(a) for a in range(170)], dtype=np.int64)
    assert np.array_equal(mv, gold["action_move"])
    assert np.array_equal(dr, gold["action_dir"])


def test_start_position(gold):
    L = _oracle.oracle()
    s = np.zeros(6, dtype=np.uint64)
    L.am_clear(s)
    assert np.array_equal(s, gold["state"][0])
    m = np.zeros(3, dtype=np.uint64)
    L.am_legal_mask(s, m)
    # legal actions at the start position
    assert _oracle.mask_bits(m, 170) == [7, 8, 9, 10, 32, 33, 34]


def test_replay_golden_games(gold):
    """Replay every golden ply: mask, observation, is_over, play result and next state."""
    L = _oracle.oracle()
    st, mk, ac, ct, ov, ob, gm = (gold[k] for k in ("state", "mask", "action", "cont", "over", "obs", "game"))
    n = len(st)
    m = np.zeros(3, dtype=np.uint64)
    o = np.zeros(64, dtype=np.int8)
    for t in range(n):
        s = st[t].copy()
        L.am_legal_mask(s, m)
        assert np.array_equal(m, mk[t]), t
        assert L.am_is_over(s) == ov[t], t
        for p in (0, 1):
            L.am_observation(s, p, o)
            assert np.array_equal(o, ob[t, p]), (t, p)
        if ac[t] >= 0:
            c = L.am_play(s, int(ac[t]))
            assert c == ct[t], t
            if t + 1 < n and gm[t + 1] == gm[t]:
                assert np.array_equal(s, st[t + 1]), t
    # final states
    finals = gold["final_state"]
    last = [np.nonzero(gm == g)[0][-1] for g in range(len(finals))]
    for g, t in enumerate(last):
        assert np.array_equal(st[t], finals[g])


def test_playout_driver_hash():
    """The seeded playout driver is deterministic (value pinned; also used by GPU parity tests)."""
    L = _oracle.oracle()
    n = 256
    boards = np.zeros((n, 6), dtype=np.uint64)
    for b in range(n):
        L.am_clear(boards[b])
    moves = L.am_playout(boards, n, 0, 0, 300, 0x5EED0001)
    assert moves > 0
    b2 = np.zeros((n, 6), dtype=np.uint64)
    for b in range(n):
        L.am_clear(b2[b])
    # same result when driven in two chunks
    m1 = L.am_playout(b2, n, 0, 0, 100, 0x5EED0001)
    m2 = L.am_playout(b2, n, 0, 100, 200, 0x5EED0001)
    assert m1 + m2 == moves
    assert np.array_equal(boards, b2)


def test_game_wrapper_features_shapes():
    g = _oracle.AmGame()
    f = g.features()
    # history has one board: slots 0..4 are zero, slot 5 holds the start position, black to move
    assert f[:30].sum() == 0
    assert f[30].sum() == 12 and f[31].sum() == 0 and f[32].sum() == 12 and f[33].sum() == 0
    assert f[34].min() == 1 and f[35].max() == 0
    assert g.forward(7) and not g.forward(7 + 49 * 0 + 163)  # second is illegal
    f = g.features()
    assert f[24].sum() == 12 and f[34].max() == 0 and f[35].min() == 1
    assert g.evaluate() == 0.0 and not g.terminated()


@pytest.mark.skipif(_oracle.ref() is None, reason="compiled reference (oracle/_ref) only exists in the build container")
def test_live_against_compiled_reference():
    """Fresh random games, restatement and reference side by side (different seed from the fixtures)."""
    L, R = _oracle.oracle(), _oracle.ref()
    rng = np.random.default_rng(99)
    plies = 0
    for g in range(200):
        s = np.zeros(6, dtype=np.uint64)
        r = np.zeros(6, dtype=np.uint64)
        L.am_clear(s)
        R.ref_am_clear(r)
        while True:
            assert np.array_equal(s, r)
            v = np.zeros(170, dtype=np.int32)
            R.ref_am_valid_moves(r, v)
            m = np.zeros(3, dtype=np.uint64)
            L.am_legal_mask(s, m)
            legal = list(np.nonzero(v)[0])
            assert _oracle.mask_bits(m, 170) == legal
            if not legal or ((int(s[4]) >> 8) & 0xFFFF) >= 250:
                break
            a = int(rng.choice(legal))
            assert L.am_play(s, a) == R.ref_am_play(r, a)
            plies += 1
    assert plies > 5000


def test_reference_driven_playouts_equal_oracle_playouts():
    """ref_am_playout runs the random-legal playout through the compiled reference's own GetValidMovesBinary /
    CheckersPlay (oracle/ref/ref_american.cc); the states after 300 plies (several finished games per board)
    must equal the oracle's playout of the same boards.  It is also the `kind: "reference"` CPU baseline of
    benchmarks/bench_boards.py."""
    R = _oracle.ref()
    if R is None or not hasattr(R, "ref_am_playout"):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    L = _oracle.oracle()
    n = 64
    a = np.zeros((n, 6), dtype=np.uint64)
    for b in range(n):
        L.am_clear(a[b])
    r = a.copy()
    ma = L.am_playout(a.reshape(-1), n, 7, 0, 300, 0x5EED0001)
    mr = R.ref_am_playout(r.reshape(-1), n, 7, 0, 300, 0x5EED0001)
    assert ma == mr and ma > 0
    assert np.array_equal(a, r)
