"""GPU parity: HIP Russian-checkers kernels (through the C-ABI) vs the oracle and the golden vectors
generated from the compiled reference.  Bit-exact on state words, legal masks, ok flags, planes."""
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
    return np.load(os.path.join(golden_dir, "russian_ref.npz"))


def _np_mask(t):
    return t.cpu().numpy().view(np.uint64)


def test_golden_positions_masks_and_steps(torch_cuda, gold):
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, RUSSIAN
    st, mk, ac, ov, gm = gold["state"], gold["mask"], gold["action"], gold["over"], gold["game"]
    n = len(st)
    bb = BoardBatch(RUSSIAN, n)
    bb.import_states(st)
    assert np.array_equal(bb.export_states(), st)
    mask, term = bb.legal_mask(with_terminal=True)
    assert np.array_equal(_np_mask(mask), mk)
    ply = (st[:, 4] >> np.uint64(8)) & np.uint64(0xFFFF)
    assert np.array_equal(term.cpu().numpy() != 0, (ov != 0) | (ply >= 250))
    ok = bb.forward(torch.tensor(ac.astype(np.int16), device="cuda")).cpu().numpy()
    assert np.array_equal(ok != 0, (ac >= 0) & (ply < 250))
    after = bb.export_states()
    checked = 0
    for t in range(n - 1):
        if gm[t + 1] == gm[t] and ac[t] >= 0:
            assert np.array_equal(after[t], st[t + 1]), t
            checked += 1
    assert checked > 2500


def test_illegal_actions_rejected(torch_cuda, gold):
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, RUSSIAN
    st, mk = gold["state"][:1024], gold["mask"][:1024]
    rng = np.random.default_rng(5)
    acts = rng.integers(-1, 281, size=len(st)).astype(np.int16)
    bb = BoardBatch(RUSSIAN, len(st))
    bb.import_states(st)
    ok = bb.forward(torch.tensor(acts, device="cuda")).cpu().numpy()
    ply = (st[:, 4] >> np.uint64(8)) & np.uint64(0xFFFF)
    legal = np.array([0 <= a < 280 and (int(mk[i, int(a) >> 6]) >> (int(a) & 63)) & 1 for i, a in enumerate(acts)]) & (ply < 250)
    assert np.array_equal(ok, legal.astype(np.uint8))
    after = bb.export_states()
    assert np.array_equal(after[ok == 0], st[ok == 0])


def test_seeded_games_step_by_step_vs_oracle(torch_cuda):
    """1024 boards driven ply by ply with random legal moves; states, masks, planes, evaluateGame."""
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, RUSSIAN, F32, BF16, BITS
    n = 1024
    L = _oracle.oracle()
    states = np.zeros((n, 5), dtype=np.uint64)
    for b in range(n):
        L.ru_clear(states[b])
    bb = BoardBatch(RUSSIAN, n)
    rng = np.random.default_rng(11)
    m = np.zeros(5, dtype=np.uint64)
    f = np.zeros(6 * 64, dtype=np.float32)
    for ply in range(120):
        mask, term = bb.legal_mask(with_terminal=True)
        mask = _np_mask(mask); term = term.cpu().numpy()
        acts = np.full(n, -1, dtype=np.int16)
        for b in range(n):
            L.ru_legal_mask(states[b], m)
            assert np.array_equal(mask[b], m), (ply, b)
            assert term[b] == L.ru_terminated(states[b])
            if not term[b]:
                legal = _oracle.mask_bits(m, 280)
                acts[b] = legal[rng.integers(len(legal))]
        ok = bb.forward(torch.tensor(acts, device="cuda")).cpu().numpy()
        for b in range(n):
            assert ok[b] == L.ru_forward(states[b], int(acts[b]))
        st = bb.export_states()
        assert np.array_equal(st, states), ply
        if ply % 10 == 9 or ply < 4:
            ff = bb.extract(F32).cpu().numpy()
            fh = bb.extract(BF16).float().cpu().numpy()
            fb = bb.extract(BITS).cpu().numpy().view(np.uint64).reshape(n, 6)
            ev = bb.evaluateGame().cpu().numpy()
            for b in range(n):
                L.ru_features(states[b], f)
                ref = f.reshape(6, 8, 8)
                assert np.array_equal(ff[b], ref), (ply, b)
                assert np.array_equal(fh[b], ref), (ply, b)
                bits = np.array([sum(1 << int(j) for j in np.nonzero(ref[p].reshape(-1))[0]) for p in range(6)], dtype=np.uint64)
                assert np.array_equal(fb[b], bits)
                assert ev[b] == L.ru_evaluate(states[b])
        if ply == 60:   # restart finished games so that late positions (kings) keep coming
            which = torch.tensor(term, device="cuda")
            bb.reset(which)
            for b in range(n):
                if term[b]:
                    L.ru_clear(states[b])


def test_playout_kernel_vs_oracle(torch_cuda):
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, RUSSIAN
    n = 4096
    L = _oracle.oracle()
    bb = BoardBatch(RUSSIAN, n)
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    bb.playout(137, seed=0x5EED0002, first_board_id=77, counter=cnt)
    bb.playout(200, seed=0x5EED0002, first_board_id=77, counter=cnt)
    got = bb.export_states()
    boards = np.zeros((n, 5), dtype=np.uint64)
    for b in range(n):
        L.ru_clear(boards[b])
    moves = L.ru_playout(boards, n, 77, 0, 337, 0x5EED0002)
    assert np.array_equal(got, boards)
    assert int(cnt.item()) == moves
    assert np.count_nonzero(boards[:, 1] | boards[:, 3]) > n // 4   # plenty of king positions reached


def test_full_size_properties(torch_cuda):
    """65,536 boards: invariants without the oracle."""
    from elf_amd.boards import BoardBatch, RUSSIAN
    n = 65536
    bb = BoardBatch(RUSSIAN, n)
    bb.playout(96, seed=9)
    s = bb.export_states()
    dark = np.uint64(0x55AA55AA55AA55AA)
    occ = [s[:, i] for i in range(4)]
    for i in range(4):
        assert not np.any(occ[i] & ~dark)
        for j in range(i + 1, 4):
            assert not np.any(occ[i] & occ[j])
    pc = lambda x: np.array([bin(int(v)).count("1") for v in x])
    assert pc(occ[0] | occ[1]).max() <= 12 and pc(occ[2] | occ[3]).max() <= 12
    assert not np.any(occ[0] & np.uint64(0xFF << 56)) and not np.any(occ[2] & np.uint64(0xFF))   # promoted
    ply = (s[:, 4] >> np.uint64(8)) & np.uint64(0xFFFF)
    assert ply.min() >= 1 and ply.max() <= 250
