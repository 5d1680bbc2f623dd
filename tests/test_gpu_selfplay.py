"""GPU self-play driver (SelfPlayBatch: search -> pick/record -> forward -> terminal/restart) against
the oracle run game by game through the same loop (ClientGameSelfPlay.cc:249-375 semantics,
policy_distri_cutoff = 0 as in the parity configs).  Records must be identical: moves, uint8
policies, predicted values, rewards."""
import numpy as np
import pytest

from tests._oracle import AmGame, OracleMcts

pytestmark = pytest.mark.gpu


def _oracle_game(opts, policy_kind, max_games):
    g = AmGame()
    m = OracleMcts(policy_kind=policy_kind, **opts)
    recs = []
    moves, vals, pols = [], [], []
    while len(recs) < max_games:
        r = m.run(g)
        vis = np.array([e[1] for e in r["edges"]], dtype=np.float32)
        p = vis / np.float32(vis.sum())
        pol = np.zeros(170, dtype=np.uint8)
        q = (p / p.max() * np.float32(255)).astype(np.float32)
        for (a, _, _, _), c in zip(r["edges"], q):
            pol[a] = np.uint8(int(c))
        moves.append(r["best"]); vals.append(np.float32(r["best_q"])); pols.append(pol)
        assert g.forward(r["best"])
        m.advance(r["best"])
        if g.terminated():
            recs.append(dict(moves=np.array(moves, dtype=np.int16), values=np.array(vals, dtype=np.float32),
                             policies=np.stack(pols), reward=g.evaluate()))
            m.clear(); g.reset()
            moves, vals, pols = [], [], []
    return recs


def test_selfplay_records_match_oracle():
    import torch
    from elf_amd.boards import AMERICAN
    from elf_amd.selfplay import SelfPlayBatch
    n = 8
    sp = SelfPlayBatch(AMERICAN, n, num_rollouts_per_thread=24, num_rollouts_per_batch=4, virtual_loss=1, c_puct=1.5)
    for _ in range(400):
        sp.step()
        if sp.win_stats["total_games"] >= 2 * n:
            break
    slabs = sp.take_records()
    assert sp.win_stats["total_games"] >= n
    _, flags = sp.ts.stats()
    assert not flags.cpu().numpy().any()
    # all games are identical (same start, deterministic search): compare each with the oracle's games
    want = _oracle_game(dict(num_rollouts=24, rollouts_per_batch=4, virtual_loss=1, c_puct=1.5), 0, 3)
    seen = 0
    for slab in slabs:
        for i in range(len(slab["game_id"])):
            k = int(slab["seq"][i])
            if k >= len(want):
                continue
            w = want[k]
            nm = int(slab["n_moves"][i])
            assert nm == len(w["moves"])
            assert np.array_equal(slab["moves"][i, :nm], w["moves"])
            assert np.array_equal(slab["values"][i, :nm], w["values"])
            assert np.array_equal(slab["policies"][i, :nm], w["policies"])
            assert float(slab["reward"][i]) == w["reward"]
            assert (slab["moves"][i, nm:] == -1).all()
            seen += 1
    assert seen >= n


def test_selfplay_with_network_and_diverse_moves_runs():
    """End-to-end plumbing with the bf16 network as evaluator and policy_distri_cutoff > 0: games
    finish, records are well-formed, sampled moves are legal (forward never rejects)."""
    import torch
    from elf_amd.boards import AMERICAN
    from elf_amd.net import PolicyValueNet
    from elf_amd.selfplay import SelfPlayBatch
    from oracle.model_ref import seeded_model
    cfg = dict(planes=36, board=8, dim=32, num_block=2, num_action=170)
    n = 64
    net = PolicyValueNet(max_batch=n * 8, **cfg)
    net.load_state_dict(seeded_model(seed=0, **cfg).state_dict())
    sp = SelfPlayBatch(AMERICAN, n, net=net, policy_distri_cutoff=12, seed=0, num_rollouts_per_thread=16,
                       num_rollouts_per_batch=8, virtual_loss=1, c_puct=1.5)
    moved = 0
    for _ in range(120):
        before = sp.nmoves.clone()
        fin = sp.step()
        moved += int(((sp.nmoves - before) == 1).sum().item()) + fin
        assert moved == (_ + 1) * n          # every game moved every step (no illegal picks)
    assert sp.win_stats["total_games"] > 0
    for slab in sp.take_records():
        nm = slab["n_moves"]
        assert (nm > 0).all() and (nm < 250).all()
        assert set(np.unique(slab["reward"])) <= {-1.0, 1.0}
        for i in range(len(nm)):
            assert slab["policies"][i, : nm[i]].max(axis=1).min() == 255   # every recorded policy has its max at 255
    # different seeds diversify the openings
    first = sp.moves[:, 0].cpu().numpy()
    assert len(np.unique(first)) > 1


def test_selfplay_goes_on_past_an_arena_overflow():
    """An arena that holds one move's rollouts but not the subtrees a persistent tree carries along: launches park
    the trees that fill it; the driver clears those and searches them again instead of stalling their games."""
    from elf_amd import _lib
    from elf_amd.boards import AMERICAN
    from elf_amd.selfplay import SelfPlayBatch
    n = 32
    sp = SelfPlayBatch(AMERICAN, n, num_rollouts_per_thread=64, num_rollouts_per_batch=8, virtual_loss=1, c_puct=1.5,
                       max_nodes=72, max_edges=72 * 14)
    for _ in range(300):
        sp.step()
        if sp.win_stats["total_games"] >= n:
            break
    assert sp.win_stats["total_games"] >= n                       # every slot finished a game: nothing stalled
    assert getattr(sp.ts, "arena_restarts", 0) > 0                # and the overflow did happen
    assert not bool(sp.ts.parked().any())
    for slab in sp.take_records():
        for i in range(len(slab["game_id"])):
            nm = int(slab["n_moves"][i])
            assert nm > 0 and (slab["moves"][i][:nm] >= 0).all()
    # an arena that cannot hold even a fresh search is a configuration error, reported as such
    tiny = SelfPlayBatch(AMERICAN, 4, num_rollouts_per_thread=64, num_rollouts_per_batch=8, virtual_loss=1, c_puct=1.5,
                         max_nodes=16, max_edges=16 * 14)
    with pytest.raises(_lib.ElfHipError):
        for _ in range(10):
            tiny.step()


def _oracle_game_sampled(opts, seed, cutoff, max_games):
    """_oracle_game with the game's own std::mt19937 (GameBase::_rng): one draw for the AI's seed at init_ai, then
    mcts_make_diverse_move samples the move while ply <= cutoff (oracle/mcts.cpp orng_*: the library's own
    std::uniform_real_distribution on the normalised visit counts)."""
    from tests import _oracle
    L = _oracle.oracle()
    rng = L.orng_create(seed)
    L.orng_next_u32(rng)                      # params.seed = _rng()  (ClientGameSelfPlay.cc:59)
    g = AmGame()
    m = OracleMcts(policy_kind=0, **opts)
    recs, moves, vals, pols, sampled = [], [], [], [], 0
    while len(recs) < max_games:
        r = m.run(g)
        vis = np.array([e[1] for e in r["edges"]], dtype=np.int32)
        move = r["best"]
        if 1 + len(moves) <= cutoff:          # getPly() <= policy_distri_cutoff
            move = r["edges"][L.orng_sample_policy(rng, len(vis), vis)][0]
            sampled += move != r["best"]
        p = vis.astype(np.float32) / np.float32(vis.sum())
        q = (p / p.max() * np.float32(255)).astype(np.float32)
        pol = np.zeros(170, dtype=np.uint8)
        for (a, _, _, _), c in zip(r["edges"], q):
            pol[a] = np.uint8(int(c))
        moves.append(move); vals.append(np.float32(r["best_q"])); pols.append(pol)
        assert g.forward(move)
        m.advance(move)
        if g.terminated():
            recs.append(dict(moves=np.array(moves, dtype=np.int16), values=np.array(vals, dtype=np.float32),
                             policies=np.stack(pols), reward=g.evaluate()))
            m.clear(); g.reset()
            moves, vals, pols = [], [], []
    L.orng_destroy(rng)
    return recs, sampled


@pytest.mark.parametrize("seed", [77, 0])
def test_sampled_moves_match_the_reference_stream(seed):
    """policy_distri_cutoff > 0 (the reference's default is 20): the sampled moves come from per-game std::mt19937
    streams through std::uniform_real_distribution exactly as elf_utils::sample_multinomial draws them, so whole
    games - moves, policies, values, rewards - equal the oracle loop's bit for bit, also across game restarts."""
    from elf_amd.boards import AMERICAN
    from elf_amd.selfplay import SelfPlayBatch
    n, cutoff = 4, 12
    opts = dict(num_rollouts=24, rollouts_per_batch=4, virtual_loss=1, c_puct=1.5)
    sp = SelfPlayBatch(AMERICAN, n, policy_distri_cutoff=cutoff, seed=seed, num_rollouts_per_thread=24, num_rollouts_per_batch=4,
                       virtual_loss=1, c_puct=1.5)
    for _ in range(1500):
        sp.step()
        if min(int(v) for v in sp.seq.cpu().numpy()) >= 2:
            break
    slabs = sp.take_records()
    total_sampled = 0
    for slot in range(n):
        s = seed if seed else (0x9E3779B9 * (slot + 1)) & 0xFFFFFFFF
        want, sampled = _oracle_game_sampled(opts, s, cutoff, 2)
        total_sampled += sampled
        got = {}
        for slab in slabs:
            for i in range(len(slab["game_id"])):
                if int(slab["game_id"][i]) == slot:
                    got[int(slab["seq"][i])] = (slab, i)
        for k in range(2):
            slab, i = got[k]
            w = want[k]
            nm = int(slab["n_moves"][i])
            assert nm == len(w["moves"]), (slot, k)
            assert np.array_equal(slab["moves"][i, :nm], w["moves"]), (slot, k)
            assert np.array_equal(slab["values"][i, :nm], w["values"])
            assert np.array_equal(slab["policies"][i, :nm], w["policies"])
            assert float(slab["reward"][i]) == w["reward"]
    assert total_sampled > 0          # the sampling did pick something other than the most visited move
