"""GPU parity: HIP Ugolki kernels (through the C-ABI) vs the oracle and the golden vectors generated
from the compiled reference.  Bit-exact on state words, legal masks, ok flags, planes."""
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
    return np.load(os.path.join(golden_dir, "ugolki_ref.npz"))


def _np_mask(t):
    return t.cpu().numpy().view(np.uint64)


def test_golden_positions_masks_steps_probes(torch_cuda, gold):
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, UGOLKI
    st, mk, ac, ov, gm, pr, pk = (gold[k] for k in ("state", "mask", "action", "over", "game", "probe", "probe_ok"))
    n = len(st)
    bb = BoardBatch(UGOLKI, n)
    bb.import_states(st)
    assert np.array_equal(bb.export_states(), st)
    mask, term = bb.legal_mask(with_terminal=True)
    assert np.array_equal(_np_mask(mask), mk)
    ply = (st[:, 3] >> np.uint64(8)) & np.uint64(0xFFFF)
    dead = (ov != 0) | (ply >= 500)
    assert np.array_equal(term.cpu().numpy() != 0, dead)
    ok = bb.forward(torch.tensor(ac.astype(np.int16), device="cuda")).cpu().numpy()
    assert np.array_equal(ok != 0, (ac >= 0) & ~dead)
    after = bb.export_states()
    checked = 0
    for t in range(n - 1):
        if gm[t + 1] == gm[t] and ac[t] >= 0:
            assert np.array_equal(after[t], st[t + 1]), t
            checked += 1
    assert checked > 9000
    # TryPlay probes recorded from the compiled reference: forward() accepts exactly those (unless terminated)
    for j in range(pr.shape[1]):
        bb.import_states(st)
        ok = bb.forward(torch.tensor(pr[:, j].copy(), device="cuda")).cpu().numpy()
        assert np.array_equal(ok != 0, (pk[:, j] != 0) & ~dead), j


def test_seeded_games_step_by_step_vs_oracle(torch_cuda):
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, UGOLKI, F32, BF16, BITS
    n = 512
    L = _oracle.oracle()
    states = np.zeros((n, 4), dtype=np.uint64)
    for b in range(n):
        L.ug_clear(states[b])
    bb = BoardBatch(UGOLKI, n)
    rng = np.random.default_rng(13)
    m = np.zeros(7, dtype=np.uint64)
    f = np.zeros(16 * 64, dtype=np.float32)
    for ply in range(100):
        mask, term = bb.legal_mask(with_terminal=True)
        mask = _np_mask(mask); term = term.cpu().numpy()
        acts = np.full(n, -1, dtype=np.int16)
        for b in range(n):
            L.ug_legal_mask(states[b], m)
            assert np.array_equal(mask[b], m), (ply, b)
            assert term[b] == L.ug_terminated(states[b])
            legal = _oracle.mask_bits(m, 418)
            if not term[b]:
                acts[b] = legal[rng.integers(len(legal))] if rng.random() < 0.9 else rng.integers(0, 418)
        ok = bb.forward(torch.tensor(acts, device="cuda")).cpu().numpy()
        for b in range(n):
            assert ok[b] == L.ug_forward(states[b], int(acts[b])), (ply, b, acts[b])
        assert np.array_equal(bb.export_states(), states), ply
        if ply % 10 == 9 or ply < 4:
            ff = bb.extract(F32).cpu().numpy()
            fh = bb.extract(BF16).float().cpu().numpy()
            fb = bb.extract(BITS).cpu().numpy().view(np.uint64).reshape(n, 16)
            ev = bb.evaluateGame().cpu().numpy()
            for b in range(n):
                L.ug_features(states[b], f)
                ref = f.reshape(16, 8, 8)
                assert np.array_equal(ff[b], ref), (ply, b)
                assert np.array_equal(fh[b], ref), (ply, b)
                bits = np.array([sum(1 << int(j) for j in np.nonzero(ref[p].reshape(-1))[0]) for p in range(16)], dtype=np.uint64)
                assert np.array_equal(fb[b], bits)
                assert ev[b] == L.ug_evaluate(states[b])


def test_terminal_values(torch_cuda, gold):
    from elf_amd.boards import BoardBatch, UGOLKI
    st, ov = gold["state"], gold["over"]
    idx = np.nonzero(ov)[0]
    L = _oracle.oracle()
    bb = BoardBatch(UGOLKI, len(idx))
    bb.import_states(st[idx])
    ev = bb.evaluateGame().cpu().numpy()
    want = np.array([L.ug_evaluate(st[t].copy()) for t in idx], dtype=np.float32)
    assert np.array_equal(ev, want) and set(want) == {1.0, -1.0}


def test_playout_kernel_vs_oracle(torch_cuda):
    torch = torch_cuda
    from elf_amd.boards import BoardBatch, UGOLKI
    n = 2048
    L = _oracle.oracle()
    bb = BoardBatch(UGOLKI, n)
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    bb.playout(237, seed=0x5EED0003, first_board_id=5, counter=cnt)
    bb.playout(300, seed=0x5EED0003, first_board_id=5, counter=cnt)   # crosses the 500-ply cap -> restarts
    got = bb.export_states()
    boards = np.zeros((n, 4), dtype=np.uint64)
    for b in range(n):
        L.ug_clear(boards[b])
    moves = L.ug_playout(boards, n, 5, 0, 537, 0x5EED0003)
    assert np.array_equal(got, boards)
    assert int(cnt.item()) == moves


def test_full_size_properties(torch_cuda):
    """65,536 boards: invariants without the oracle - 9 men a side for ever, no overlap."""
    from elf_amd.boards import BoardBatch, UGOLKI
    n = 65536
    bb = BoardBatch(UGOLKI, n)
    bb.playout(128, seed=9)
    s = bb.export_states()
    assert not np.any(s[:, 0] & s[:, 1])
    pc = lambda x: np.array([bin(int(v)).count("1") for v in x])
    assert np.all(pc(s[:, 0]) == 9) and np.all(pc(s[:, 1]) == 9)
    ply = (s[:, 3] >> np.uint64(8)) & np.uint64(0xFFFF)
    assert ply.min() >= 1 and ply.max() <= 500
    ja = s[:, 2]
    assert np.all((pc(ja) == 0) | (pc(ja) == 2))
