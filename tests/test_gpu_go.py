"""GPU parity: HIP Go kernels (through the C-ABI) vs oracle/go.c - bit-exact states (stones, ko,
captures, Zobrist hash, superko flag), legal masks, ok flags and Tromp-Taylor values.
PARITY UNPINNED vs ELF (no Go in the reference fork); the oracle is pinned by known answers."""
import numpy as np
import pytest

from tests import _oracle
from tests._oracle import GO_PASS, GoBoards, go_grid_to_words
from tests.test_oracle_go import ko_position, send_two_return_one, meta, P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def _mask(t):
    return t.cpu().numpy().view(np.uint64)


def _step(torch, bb, acts):
    return bb.forward(torch.tensor(np.asarray(acts, dtype=np.int16), device="cuda")).cpu().numpy()


def test_known_answers_on_gpu(torch_cuda):
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, GO19
    bb = BoardBatch(GO19, 4)
    g = GoBoards(4)
    # board 0: ko; board 1: send-two-return-one superko cycle; board 2: corner capture; board 3: empty
    starts = [go_grid_to_words(ko_position(), 0), go_grid_to_words(send_two_return_one(), 0)]
    g.load(starts[0], 0); g.load(starts[1], 1)
    st = g.export_all()
    bb.import_states(st)
    assert np.array_equal(bb.export_states(), st)
    script = [
        [P(1, 2), P(0, 0), P(0, 1), GO_PASS],
        [P(1, 1), P(0, 2), P(0, 0), GO_PASS],   # board0: illegal ko retake
        [P(10, 10), P(0, 1), P(1, 0), 5],
        [P(10, 12), P(5, 5), P(0, 0), 5],       # board1 is terminated (superko) by now; board2: white suicide illegal
        [P(1, 1), 7, GO_PASS, 400],             # 400 out of range
    ]
    for acts in script:
        ok = _step(torch, bb, acts)
        exp = [int(g.forward(a, i)) if 0 <= a <= 361 else 0 for i, a in enumerate(acts)]
        assert list(ok) == exp, (acts, ok, exp)
        assert np.array_equal(bb.export_states(), g.export_all()), acts
        m, term = bb.legal_mask(with_terminal=True)
        assert np.array_equal(_mask(m), np.stack([g.mask(i) for i in range(4)]))
        assert list(term.cpu().numpy()) == [int(g.terminated(i)) for i in range(4)]
    assert meta(bb.export_states()[1])["superko"] == 1


def test_seeded_games_step_by_step_vs_oracle(torch_cuda):
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, GO19
    n = 192
    bb = BoardBatch(GO19, n)
    g = GoBoards(n)
    rng = np.random.default_rng(17)
    for ply in range(420):
        mask = _mask(bb.legal_mask())
        acts = np.full(n, -1, dtype=np.int16)
        for b in range(n):
            m = g.mask(b)
            assert np.array_equal(mask[b], m), (ply, b)
            legal = _oracle.mask_bits(m, 361)
            if g.terminated(b):
                continue
            # mostly random legal points; sometimes a random (possibly illegal) point or a pass
            u = rng.random()
            if u < 0.02:
                acts[b] = rng.integers(0, 362)
            elif legal:
                acts[b] = legal[rng.integers(len(legal))]
            else:
                acts[b] = GO_PASS
        ok = _step(torch, bb, acts)
        exp = np.array([int(g.forward(int(a), b)) if a >= 0 else 0 for b, a in enumerate(acts)], dtype=np.uint8)
        assert np.array_equal(ok, exp), ply
        if ply % 7 == 0 or ply > 400:
            assert np.array_equal(bb.export_states(), g.export_all()), ply
    assert np.array_equal(bb.export_states(), g.export_all())
    v = bb.evaluateGame().cpu().numpy()
    assert np.array_equal(v, np.array([g.evaluate(b) for b in range(n)], dtype=np.float32))


def test_playout_kernel_vs_oracle(torch_cuda):
    """Persistent playout kernel vs the CPU driver, long enough to cross the 722-ply cap, resets
    and (where they occur) superko terminations."""
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, GO19
    n = 512
    bb = BoardBatch(GO19, n)
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    bb.playout(300, seed=0x5EED0001, first_board_id=77, counter=cnt)
    bb.playout(600, seed=0x5EED0001, first_board_id=77, counter=cnt)
    got = bb.export_states()
    g = GoBoards(n)
    moves = g.playout(900, seed=0x5EED0001, board_id0=77)
    ref = g.export_all()
    assert np.array_equal(got, ref)
    assert int(cnt.item()) == moves
    # the run crossed at least one game boundary on every board
    ply = (ref[:, 20] >> np.uint64(16)) & np.uint64(0xFFFF)
    assert ply.max() < 722 and moves < n * 900


def test_dense_playout_kernel_matches_plain(torch_cuda):
    """A slab with more boards than one wave per SIMD holds (65,536 on the MI355X) runs the playout kernel built
    for two waves per SIMD (256-thread blocks, register limit); same body, so its boards must come out exactly
    as the plain kernel's - which the test above holds against the oracle - on the same board ids."""
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, GO19
    n0, n1 = 65536, 2048
    big = BoardBatch(GO19, n0 + n1)
    cb = torch.zeros(1, dtype=torch.int64, device="cuda")
    big.playout(150, seed=0x5EED0003, first_board_id=5, counter=cb)
    big.playout(100, seed=0x5EED0003, first_board_id=5, counter=cb)
    got = big.export_states()
    del big
    parts, moves = [], 0
    for first, n in ((5, n0), (5 + n0, n1)):
        bb = BoardBatch(GO19, n)
        c = torch.zeros(1, dtype=torch.int64, device="cuda")
        bb.playout(150, seed=0x5EED0003, first_board_id=first, counter=c)
        bb.playout(100, seed=0x5EED0003, first_board_id=first, counter=c)
        parts.append(bb.export_states()); moves += int(c.item())
        del bb
    assert np.array_equal(got, np.concatenate(parts))
    assert int(cb.item()) == moves


def test_full_size_properties(torch_cuda):
    """65,536 boards (BASELINE configs[1] size): oracle-free invariants after 200 plies:
    colours disjoint, hash == from-scratch Zobrist, no stone group without liberties, legal mask
    subset of empty points, stone count + captures consistent."""
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, GO19
    n = 65536
    bb = BoardBatch(GO19, n)
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    bb.playout(200, seed=99, counter=cnt)
    s = bb.export_states()
    assert int(cnt.item()) == n * 200
    bl = (s[:, :19] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    wh = (s[:, :19] >> np.uint64(32)).astype(np.uint32)
    assert not np.any(bl & wh) and not np.any((bl | wh) >> np.uint32(19))
    L = _oracle.oracle()
    z = np.array([[L.go_zobrist(c, p) for p in range(361)] for c in range(2)], dtype=np.uint64)
    bits = lambda rows: ((rows[:, :, None] >> np.arange(19, dtype=np.uint32)[None, None, :]) & 1).reshape(len(rows), 361).astype(bool)
    bb_, wb_ = bits(bl), bits(wh)
    h = np.zeros(n, dtype=np.uint64)
    for p in range(361):
        h ^= np.where(bb_[:, p], z[0, p], np.uint64(0)) ^ np.where(wb_[:, p], z[1, p], np.uint64(0))
    assert np.array_equal(h, s[:, 19])
    # stones placed = stones on board + stones captured (no passes can have happened this early)
    nb, nw = bb_.sum(1), wb_.sum(1)
    capb = (s[:, 21] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    capw = (s[:, 21] >> np.uint64(32)).astype(np.int64)
    ply = ((s[:, 20] >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64)
    passes_possible = ply != 200
    ok = (nb + capw == 100) & (nw + capb == 100)
    assert np.all(ok | passes_possible) and ok.mean() > 0.9
    # legal mask: subset of empty points; spot-check 64 boards fully against the oracle
    mask = _mask(bb.legal_mask())
    mb = np.zeros((n, 361), dtype=bool)
    for p in range(361):
        mb[:, p] = (mask[:, p >> 6] >> np.uint64(p & 63)) & np.uint64(1)
    assert not np.any(mb & (bb_ | wb_))
    g = GoBoards(1)
    for b in range(0, n, 1024):
        g.load(s[b])
        m = g.mask()
        ko = (int(s[b, 20]) >> 32) & 0xFFFF
        assert np.array_equal(mask[b], m), b


def test_feature_planes_vs_oracle(torch_cuda):
    """18 planes (8-deep stone history x 2 + colour planes), f32 / bf16 / bit-packed, incl. passes,
    young games (short history) and boards imported mid-game."""
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, GO19, F32, BF16, BITS
    n = 48
    bb = BoardBatch(GO19, n)
    g = GoBoards(n)
    rng = np.random.default_rng(23)

    def check(tag):
        f = bb.extract(F32).cpu().numpy()
        h = bb.extract(BF16).float().cpu().numpy()
        bits = bb.extract(BITS).cpu().numpy().view(np.uint64).reshape(n, 18, 6)
        for b in range(n):
            ref = g.features(b)
            assert np.array_equal(f[b], ref), (tag, b)
            assert np.array_equal(h[b], ref), (tag, b)
            flat = ref.reshape(18, 361)
            for pl in range(18):
                w = np.zeros(6, dtype=np.uint64)
                for p in np.nonzero(flat[pl])[0]:
                    w[p >> 6] |= np.uint64(1) << np.uint64(p & 63)
                assert np.array_equal(bits[b, pl], w), (tag, b, pl)

    check("start")
    for ply in range(40):
        acts = np.full(n, -1, dtype=np.int16)
        for b in range(n):
            if g.terminated(b) or b > ply * 3:      # staggered starts -> different history lengths
                continue
            legal = [a for a in g.legal(b) if a != GO_PASS]
            acts[b] = GO_PASS if (rng.random() < 0.05 or not legal) else legal[rng.integers(len(legal))]
            assert g.forward(int(acts[b]), b)
        bb.forward(torch.tensor(acts, device="cuda"))
        if ply in (0, 1, 5, 7, 8, 9, 20, 39):
            check(ply)
    # import resets the plane history to the single imported position (oracle go_import does the same)
    st = g.export_all()
    bb.import_states(st)
    for b in range(n):
        g.load(st[b], b)
    check("import")
