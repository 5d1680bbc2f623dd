"""Known-answer tests that pin the build-defined Go rules of oracle/go.c.

PARITY UNPINNED vs ELF: the reference fork has no Go source, so these answers are
derived from the rules themselves (capture, suicide, simple ko, positional superko, Tromp-Taylor),
not from reference output.
"""
import numpy as np

from tests import _oracle
from tests._oracle import GO_PASS, GO_NO_POINT, GoBoards, go_grid_to_words, go_words_to_grid


def P(r, c):
    return r * 19 + c


def meta(w):
    m = int(w[20])
    return dict(to_move=m & 1, passes=(m >> 1) & 3, superko=(m >> 3) & 1, ply=(m >> 16) & 0xFFFF,
                ko=(m >> 32) & 0xFFFF, last=(m >> 48) & 0xFFFF, cap_b=int(w[21]) & 0xFFFFFFFF, cap_w=int(w[21]) >> 32)


def test_start_position():
    g = GoBoards()
    assert g.legal() == list(range(362))
    w = g.export()
    assert w[19] == 0 and meta(w) == dict(to_move=0, passes=0, superko=0, ply=0, ko=GO_NO_POINT, last=GO_NO_POINT, cap_b=0, cap_w=0)


def test_corner_capture_and_suicide():
    g = GoBoards()
    assert g.play(0, 1) and g.play(0, 0) and g.play(1, 0)      # B, W in the corner, B captures
    grid = go_words_to_grid(g.export())
    assert grid[0, 0] == 0 and grid[0, 1] == 1 and grid[1, 0] == 1
    m = meta(g.export())
    assert m["cap_b"] == 1 and m["cap_w"] == 0 and m["ko"] == GO_NO_POINT and m["to_move"] == 1
    assert P(0, 0) not in g.legal()                              # white suicide in the corner
    assert not g.play(0, 0)
    assert not g.play(0, 1)                                      # occupied
    assert g.forward(GO_PASS)
    assert P(0, 0) in g.legal()                                  # black may fill its own eye (no eye rule)


def test_edge_and_centre_capture_multi_stone():
    # white two-stone group on the edge captured at once
    grid = np.zeros((19, 19), dtype=np.int8)
    grid[0, 3] = grid[0, 4] = 2
    grid[0, 2] = grid[1, 3] = grid[1, 4] = 1
    g = GoBoards()
    g.load(go_grid_to_words(grid, to_move=0))
    assert g.play(0, 5)
    out = go_words_to_grid(g.export())
    assert out[0, 3] == 0 and out[0, 4] == 0 and meta(g.export())["cap_b"] == 2
    # one move capturing two separate groups (centre)
    grid = np.zeros((19, 19), dtype=np.int8)
    grid[9, 8] = 2; grid[9, 10] = 2
    for (r, c) in [(8, 8), (10, 8), (9, 7), (8, 10), (10, 10), (9, 11)]:
        grid[r, c] = 1
    g.load(go_grid_to_words(grid, to_move=0))
    assert g.play(9, 9)
    out = go_words_to_grid(g.export())
    assert out[9, 8] == 0 and out[9, 10] == 0 and out[9, 9] == 1
    assert meta(g.export())["cap_b"] == 2 and meta(g.export())["ko"] == GO_NO_POINT


def test_capture_beats_suicide():
    # black plays into a point with no liberties but captures -> legal
    grid = np.zeros((19, 19), dtype=np.int8)
    grid[0, 0] = 0; grid[0, 1] = 2; grid[1, 0] = 2; grid[1, 1] = 1; grid[0, 2] = 1; grid[2, 0] = 1
    g = GoBoards()
    g.load(go_grid_to_words(grid, to_move=0))
    assert P(0, 0) in g.legal()
    assert g.play(0, 0)
    out = go_words_to_grid(g.export())
    assert out[0, 1] == 0 and out[1, 0] == 0 and out[0, 0] == 1 and meta(g.export())["cap_b"] == 2


def ko_position():
    grid = np.zeros((19, 19), dtype=np.int8)
    for (r, c) in [(0, 1), (1, 0), (2, 1)]:
        grid[r, c] = 1
    for (r, c) in [(0, 2), (1, 3), (2, 2), (1, 1)]:
        grid[r, c] = 2
    return grid


def test_simple_ko():
    g = GoBoards()
    g.load(go_grid_to_words(ko_position(), to_move=0))
    assert g.play(1, 2)                                          # black takes the ko
    m = meta(g.export())
    assert m["ko"] == P(1, 1) and m["cap_b"] == 1
    assert P(1, 1) not in g.legal() and not g.play(1, 1)         # white may not retake at once
    assert g.play(10, 10) and g.play(10, 12)                     # ko threat and answer
    assert meta(g.export())["ko"] == GO_NO_POINT
    assert P(1, 1) in g.legal() and g.play(1, 1)                 # now white retakes
    assert meta(g.export())["ko"] == P(1, 2)


def test_positional_superko_flag_matches_position_history():
    """Property: along random games the superko flag is raised exactly when the full position
    (compared stone by stone, not by hash) has occurred before in that game."""
    rng = np.random.default_rng(3)
    raised = 0
    for game in range(30):
        g = GoBoards()
        seen = {go_words_to_grid(g.export()).tobytes()}
        for ply in range(722):
            legal = [a for a in g.legal() if a != GO_PASS]
            if g.terminated():
                break
            a = int(rng.choice(legal)) if legal else GO_PASS
            assert g.forward(a)
            if a == GO_PASS:
                continue
            key = go_words_to_grid(g.export()).tobytes()
            flag = meta(g.export())["superko"]
            assert flag == (1 if key in seen else 0), (game, ply)
            seen.add(key)
            raised += flag


def send_two_return_one():
    """S: black p=(0,1); white (1,0),(1,1); black (0,3),(1,2).  B q=(0,0), W w=(0,2) takes two,
    B p=(0,1) takes one back -> the stones of S again (a 3-move cycle no simple-ko rule catches)."""
    grid = np.zeros((19, 19), dtype=np.int8)
    grid[0, 1] = 1; grid[0, 3] = 1; grid[1, 2] = 1
    grid[1, 0] = 2; grid[1, 1] = 2
    return grid


def test_positional_superko_constructed_cycle():
    g = GoBoards()
    g.load(go_grid_to_words(send_two_return_one(), to_move=0))
    start = go_words_to_grid(g.export())
    assert g.play(0, 0) and meta(g.export())["superko"] == 0
    assert g.play(0, 2) and meta(g.export())["superko"] == 0 and meta(g.export())["cap_w"] == 2
    assert meta(g.export())["ko"] == GO_NO_POINT
    assert g.play(0, 1)
    m = meta(g.export())
    assert m["cap_b"] == 1 and m["ko"] == GO_NO_POINT
    assert np.array_equal(go_words_to_grid(g.export()), start)
    assert m["superko"] == 1 and g.terminated() and g.legal() == []


def test_two_passes_and_ply_cap_terminate():
    g = GoBoards()
    assert g.forward(GO_PASS) and not g.terminated() and g.forward(GO_PASS) and g.terminated()
    assert g.legal() == [] and not g.forward(0)
    w = go_grid_to_words(np.zeros((19, 19), dtype=np.int8), to_move=0, ply=722)
    g.load(w)
    assert g.terminated()


def test_tromp_taylor_known_scores():
    g = GoBoards()
    assert g.score() == -7.5
    g.play(3, 3)
    assert g.score() == 361 - 7.5
    grid = np.zeros((19, 19), dtype=np.int8)
    grid[:, 9] = 1
    grid[:, 10] = 2
    g.load(go_grid_to_words(grid))
    assert g.score() == 190 - 171 - 7.5
    grid[5, 3] = 2        # a white stone inside black's side: that region now touches both colours
    g.load(go_grid_to_words(grid))
    assert g.score() == 19 - (171 + 1) - 7.5


def test_zobrist_incremental_equals_from_scratch_and_features():
    g = GoBoards()
    rng = np.random.default_rng(8)
    boards = []
    for ply in range(200):
        legal = [a for a in g.legal() if a != GO_PASS]
        if g.terminated() or not legal:
            break
        g.forward(int(rng.choice(legal)))
        boards.append(go_words_to_grid(g.export()))
        w = g.export()
        assert w[19] == go_grid_to_words(go_words_to_grid(w))[19]
    f = g.features()
    tm = meta(g.export())["to_move"]
    for i in range(8):
        grid = boards[-1 - i]
        assert np.array_equal(f[2 * i], (grid == tm + 1).astype(np.float32))
        assert np.array_equal(f[2 * i + 1], (grid == 2 - tm).astype(np.float32))
    assert f[16 + tm].min() == 1 and f[17 - tm].max() == 0


def test_d4_symmetry_of_legality():
    """Legal masks commute with the 8 board symmetries (rules have no orientation)."""
    g = GoBoards()
    rng = np.random.default_rng(21)
    for _ in range(150):
        legal = [a for a in g.legal() if a != GO_PASS]
        g.forward(int(rng.choice(legal)))
    w = g.export()
    grid = go_words_to_grid(w)
    base = np.zeros(361, dtype=np.int8)
    base[[a for a in g.legal() if a != GO_PASS]] = 1
    base = base.reshape(19, 19)
    ko = meta(w)["ko"]
    for k in range(4):
        for flip in (False, True):
            tg = np.rot90(grid, k)
            tb = np.rot90(base, k)
            if flip:
                tg, tb = tg.T, tb.T
            ww = go_grid_to_words(tg, to_move=meta(w)["to_move"])
            h = GoBoards()
            h.load(ww)
            got = np.zeros(361, dtype=np.int8)
            got[[a for a in h.legal() if a != GO_PASS]] = 1
            got = got.reshape(19, 19)
            if ko != GO_NO_POINT:   # the ko point is not carried by go_grid_to_words: allow it
                kom = np.zeros((19, 19), dtype=np.int8); kom[ko // 19, ko % 19] = 1
                kom = np.rot90(kom, k); kom = kom.T if flip else kom
                got = got & ~kom
            assert np.array_equal(got, tb)
