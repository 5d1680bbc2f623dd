"""GPU parity: HIP American-checkers kernels (through the C-ABI) vs the oracle and golden vectors.
Bit-exact on state words, legal masks, ok flags, planes and evaluateGame."""
import os

import numpy as np
import pytest

from tests import _oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "american_ref.npz"))


def _np_mask(t):
    return t.cpu().numpy().view(np.uint64)


def test_golden_positions_masks_and_steps(torch_cuda, gold):
    """Import every golden position (from the compiled reference), check mask / terminal / step."""
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, AMERICAN
    st, mk, ac, ov = gold["state"], gold["mask"], gold["action"], gold["over"]
    n = len(st)
    bb = BoardBatch(AMERICAN, n)
    bb.import_states(st)
    assert np.array_equal(bb.export_states(), st)
    mask, term = bb.legal_mask(with_terminal=True)
    assert np.array_equal(_np_mask(mask), mk)
    ply = (st[:, 4] >> np.uint64(8)) & np.uint64(0xFFFF)
    assert np.array_equal(term.cpu().numpy() != 0, (ov != 0) | (ply >= 250))
    ok = bb.forward(torch.tensor(ac.astype(np.int16), device="cuda"))
    okh = ok.cpu().numpy()
    assert np.array_equal(okh != 0, (ac >= 0) & (ply < 250))
    after = bb.export_states()
    L = _oracle.oracle()
    for t in range(n):
        s = st[t].copy()
        if okh[t]:
            L.am_play(s, int(ac[t]))
        assert np.array_equal(after[t], s), t
    # golden next-state check (independent of the oracle)
    gm = gold["game"]
    for t in range(n - 1):
        if gm[t + 1] == gm[t] and ac[t] >= 0:
            assert np.array_equal(after[t], st[t + 1]), t


def test_illegal_actions_rejected(torch_cuda, gold):
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, AMERICAN
    st, mk = gold["state"][:512], gold["mask"][:512]
    rng = np.random.default_rng(5)
    acts = rng.integers(0, 170, size=len(st)).astype(np.int16)
    bb = BoardBatch(AMERICAN, len(st))
    bb.import_states(st)
    ok = bb.forward(torch.tensor(acts, device="cuda")).cpu().numpy()
    ply = (st[:, 4] >> np.uint64(8)) & np.uint64(0xFFFF)
    legal = np.array([(int(mk[i, int(a) >> 6]) >> (int(a) & 63)) & 1 for i, a in enumerate(acts)]) & (ply < 250)
    assert np.array_equal(ok, legal.astype(np.uint8))
    after = bb.export_states()
    assert np.array_equal(after[ok == 0], st[ok == 0])


def test_seeded_games_step_by_step_vs_oracle(torch_cuda):
    """4096 boards driven ply by ply with oracle-chosen random legal moves; states, masks, planes."""
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, AMERICAN, F32, BITS
    n = 1024
    L = _oracle.oracle()
    games = [_oracle.AmGame() for _ in range(n)]
    bb = BoardBatch(AMERICAN, n)
    rng = np.random.default_rng(11)
    m = np.zeros(3, dtype=np.uint64)
    for ply in range(60):
        mask = _np_mask(bb.legal_mask())
        acts = np.full(n, -1, dtype=np.int16)
        for b, g in enumerate(games):
            s = g.board()
            L.am_legal_mask(s, m)
            assert np.array_equal(mask[b], m), (ply, b)
            if not g.terminated():
                legal = _oracle.mask_bits(m, 170)
                acts[b] = legal[rng.integers(len(legal))]
        ok = bb.forward(torch.tensor(acts, device="cuda")).cpu().numpy()
        for b, g in enumerate(games):
            if acts[b] >= 0:
                assert g.forward(acts[b]) and ok[b] == 1
            else:
                assert ok[b] == 0
        if ply % 10 == 9 or ply < 8:
            st = bb.export_states()
            f = bb.extract(F32).cpu().numpy()
            fb = bb.extract(BITS).cpu().numpy().view(np.uint64).reshape(n, 36)
            ev = bb.evaluateGame().cpu().numpy()
            for b, g in enumerate(games):
                assert np.array_equal(st[b], g.board()), (ply, b)
                ref = g.features()
                assert np.array_equal(f[b], ref), (ply, b)
                bits = np.array([sum(1 << j for j in np.nonzero(ref[p].reshape(-1))[0]) for p in range(36)], dtype=np.uint64)
                assert np.array_equal(fb[b], bits)
                assert ev[b] == g.evaluate()


def test_playout_kernel_vs_oracle(torch_cuda):
    """Persistent multi-ply playout kernel == CPU playout driver on the same counter RNG."""
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, AMERICAN
    n = 4096
    L = _oracle.oracle()
    bb = BoardBatch(AMERICAN, n)
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    bb.playout(137, seed=0x5EED0001, first_board_id=1000, counter=cnt)
    bb.playout(200, seed=0x5EED0001, first_board_id=1000, counter=cnt)
    got = bb.export_states()
    boards = np.zeros((n, 6), dtype=np.uint64)
    for b in range(n):
        L.am_clear(boards[b])
    moves = L.am_playout(boards, n, 1000, 0, 337, 0x5EED0001)
    assert np.array_equal(got, boards)
    assert int(cnt.item()) == moves


def test_full_size_properties(torch_cuda):
    """65,536 boards (BASELINE C2 size): invariants that need no oracle - piece counts never grow,
    pieces stay on playable squares, colours never overlap, ply within bounds."""
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, AMERICAN
    n = 65536
    bb = BoardBatch(AMERICAN, n)
    bb.playout(64, seed=7)
    s = bb.export_states()
    unused = np.uint64((1 << 8) | (1 << 17) | (1 << 26) | (1 << 35) | (~((1 << 36) - 1) & (2**64 - 1)))
    black = s[:, 0] | s[:, 2]
    white = s[:, 1] | s[:, 3]
    assert not np.any(black & white)
    assert not np.any((black | white) & unused)
    pc = lambda x: np.array([bin(int(v)).count("1") for v in x])
    assert pc(black).max() <= 12 and pc(white).max() <= 12
    ply = (s[:, 4] >> np.uint64(8)) & np.uint64(0xFFFF)
    assert ply.min() >= 1 and ply.max() <= 250
    # kings are a subset of pieces that also move the other way: black kings in backward[0] subset forward[0]
    assert not np.any(s[:, 2] & ~s[:, 0]) and not np.any(s[:, 1] & ~s[:, 3])
