"""Two AIs per game and the policy-only move, through the C-ABI.

Reference: ClientGameSelfPlay::restart / act (common/ClientGameSelfPlay.cc:155-196, 334-357): _ai1
("actor_black") and, when white_ver >= 0, _ai2 ("actor_white") with its own c_puct / rollouts; the AI that
owns the side to move searches, each AI keeps its OWN persistent tree and replays the other side's moves
into it (MCTSAI_T::advanceMoves, mcts.h:189-202).  With black/white_use_policy_network_only the move is
MCTSAI_T::actPolicyOnly (mcts.h:107-113).  The oracle plays the same games with two OracleMcts objects."""
import numpy as np
import pytest

from tests import _oracle
from tests._oracle import AmGame, OracleMcts

pytestmark = pytest.mark.gpu


def _oopts(o):
    return dict(num_rollouts=o["num_rollouts_per_thread"], rollouts_per_batch=o["num_rollouts_per_batch"],
                virtual_loss=o["virtual_loss"], c_puct=o["c_puct"])


def _openings(torch, n, start_plies, seed):
    from elf_amd.boards import BoardBatch, AMERICAN
    L = _oracle.oracle()
    rng = np.random.default_rng(seed)
    games = [AmGame() for _ in range(n)]
    bb = BoardBatch(AMERICAN, n)
    for ply in range(start_plies):
        acts = np.full(n, -1, dtype=np.int16)
        for i, g in enumerate(games):
            if ply >= i % (start_plies + 1) or g.terminated():
                continue
            m = np.zeros(3, dtype=np.uint64); L.am_legal_mask(g.board(), m)
            legal = _oracle.mask_bits(m, 170)
            acts[i] = legal[rng.integers(len(legal))]
            assert g.forward(acts[i])
        bb.forward(torch.tensor(acts, device="cuda"))
    return games, bb


def _scripted_evaluator(torch, n, B, seed):
    L = _oracle.oracle()

    def evaluator(feat, states, need):
        s = states.cpu().numpy().view(np.uint64).reshape(n * B, 6)
        pi = np.zeros((n * B, 170), dtype=np.float32); v = np.zeros(n * B, dtype=np.float32)
        vv = np.zeros(1, dtype=np.float32)
        for k in range(n * B):
            L.omcts_scripted_policy(np.ascontiguousarray(s[k]), seed, pi[k], vv)
            v[k] = vv[0]
        return torch.tensor(pi, device="cuda"), torch.tensor(v, device="cuda")
    return evaluator


def _players(bb):
    return (bb.export_states()[:, 4] & np.uint64(1)).astype(np.uint8)   # include/elfhip.h: word 4 bit 0 = side to move


@pytest.mark.parametrize("policy_kind", [0, 1])
def test_two_ais_with_overrides_match_oracle(policy_kind):
    import torch
    from elf_amd.boards import AMERICAN
    from elf_amd.mcts import TreeSearchBatch
    n, seed = 12, 5
    o1 = dict(num_rollouts_per_thread=24, num_rollouts_per_batch=4, virtual_loss=1, c_puct=1.5)
    o2 = dict(num_rollouts_per_thread=16, num_rollouts_per_batch=2, virtual_loss=1, c_puct=2.25)   # white_puct / white_mcts_rollout_per_*
    games, bb = _openings(torch, n, 5, seed)
    ts = [TreeSearchBatch(AMERICAN, n, **o1), TreeSearchBatch(AMERICAN, n, **o2)]
    ev = [_scripted_evaluator(torch, n, o["num_rollouts_per_batch"], seed) for o in (o1, o2)]
    orc = [[OracleMcts(policy_kind=policy_kind, policy_seed=seed, **_oopts(o)) for _ in range(n)] for o in (o1, o2)]
    used = [0, 0]
    for mv in range(40):
        pl = _players(bb)
        term = bb.terminated().cpu().numpy().astype(bool)
        move = torch.full((n,), -1, dtype=torch.int16, device="cuda")
        val = torch.zeros(n, dtype=torch.float32, device="cuda")
        for k in (0, 1):
            act = torch.tensor(((pl == k) & ~term).astype(np.uint8), device="cuda")
            ts[k].set_active(act)
            assert ts[k].set_roots(bb).cpu().numpy().all()
            if policy_kind == 0:
                ts[k].run_uniform()
            else:
                ts[k].run(ev[k])
            ts[k].L.elfhip_mcts_pick_moves(ts[k].h, None, 0, None, move.data_ptr(), None, val.data_ptr())
        got_m, got_v = move.cpu().numpy(), val.cpu().numpy()
        want = np.full(n, -1, dtype=np.int16)
        for i, g in enumerate(games):
            if g.terminated():
                continue
            k = int(pl[i])
            r = orc[k][i].run(g)
            used[k] += 1
            want[i] = r["best"]
            assert got_m[i] == r["best"], (mv, i, k)
            assert got_v[i] == np.float32(r["best_q"]) or (np.isnan(got_v[i]) and np.isnan(r["best_q"]))
            assert g.forward(r["best"])
            for kk in (0, 1):
                orc[kk][i].advance(r["best"])
        assert np.array_equal(got_m, want)
        bb.forward(move)
        for k in (0, 1):
            ts[k].treeAdvance(move)
    assert used[0] > 0 and used[1] > 0
    for k in (0, 1):
        st, flags = ts[k].stats()
        assert not (flags.cpu().numpy() & ~32).any()
        tot = {key: sum(o.stats()[key] for o in orc[k]) for key in ("rollouts", "evals", "depth_sum", "edge_sum")}
        assert st == tot, (k, st, tot)


def test_policy_only_for_black_matches_oracle():
    """black_use_policy_network_only: black moves by actPolicyOnly on the shared _ai1 tree (root evaluated only
    when the persistent tree has not visited it yet), white by the full search."""
    import torch
    from elf_amd.boards import AMERICAN
    from elf_amd.mcts import TreeSearchBatch
    n, seed = 10, 9
    o1 = dict(num_rollouts_per_thread=24, num_rollouts_per_batch=4, virtual_loss=1, c_puct=1.5)
    games, bb = _openings(torch, n, 4, seed)
    ts = TreeSearchBatch(AMERICAN, n, **o1)
    ev = _scripted_evaluator(torch, n, 4, seed)
    orc = [OracleMcts(policy_kind=1, policy_seed=seed, **_oopts(o1)) for _ in range(n)]
    n_policy = n_fresh = 0
    for mv in range(40):
        pl = _players(bb)
        term = bb.terminated().cpu().numpy().astype(bool)
        alive = ~term
        ts.set_active(torch.tensor(alive.astype(np.uint8), device="cuda"))
        assert ts.set_roots(bb).cpu().numpy().all()
        pol_only = (pl == 0) & alive
        visited = ts.result()["n_edges"].cpu().numpy() > 0
        move = torch.full((n,), -1, dtype=torch.int16, device="cuda")
        val = torch.zeros(n, dtype=torch.float32, device="cuda")
        # full search for white
        ts.set_active(torch.tensor(((pl == 1) & alive).astype(np.uint8), device="cuda"))
        ts.run(ev)
        ts.L.elfhip_mcts_pick_moves(ts.h, None, 0, None, move.data_ptr(), None, val.data_ptr())
        # policy only for black: one batch where the root is unvisited, then strongest prior
        ts.set_active(torch.tensor((pol_only & ~visited).astype(np.uint8), device="cuda"))
        st, ft, ne = ts.select(want_states=True, want_features=True)
        pi, v = ev(ft, st, ne)
        ts.backup(pi, v)
        ts.set_active(torch.tensor(alive.astype(np.uint8), device="cuda"))
        which = torch.tensor(pol_only.astype(np.uint8), device="cuda")
        ts.L.elfhip_mcts_pick_policy_only(ts.h, which.data_ptr(), move.data_ptr(), val.data_ptr())
        got_m, got_v = move.cpu().numpy(), val.cpu().numpy()
        for i, g in enumerate(games):
            if g.terminated():
                assert got_m[i] == -1
                continue
            orc[i].set_policy_only(pl[i] == 0)
            r = orc[i].run(g)
            if pl[i] == 0:
                n_policy += 1
                n_fresh += int(not visited[i])
            assert got_m[i] == r["best"], (mv, i, int(pl[i]))
            assert got_v[i] == np.float32(r["best_q"]) or (np.isnan(got_v[i]) and np.isnan(r["best_q"])), (mv, i)
            assert g.forward(r["best"])
            orc[i].advance(r["best"])
        bb.forward(move)
        ts.treeAdvance(move)
    assert n_policy > 20 and 0 < n_fresh
    st, flags = ts.stats()
    assert not (flags.cpu().numpy() & ~32).any()
    tot = {key: sum(o.stats()[key] for o in orc) for key in ("rollouts", "evals", "depth_sum", "edge_sum")}
    assert st["evals"] == tot["evals"] and st["edge_sum"] == tot["edge_sum"]


def _two_ai_generic(game_id, n, nmoves, make_games, side_of, run_oracle, forward, export_ok, na, edge_cap):
    """Two handles with different options over the same games, uniform evaluator; the handle that owns the side to
    move searches (set_active), both advance.  Compared per move with two oracle searches per game."""
    import torch
    from elf_amd.mcts import TreeSearchBatch
    o1 = dict(num_rollouts_per_thread=16, num_rollouts_per_batch=4, virtual_loss=1, c_puct=1.5)
    o2 = dict(num_rollouts_per_thread=12, num_rollouts_per_batch=2, virtual_loss=1, c_puct=2.5)
    games, bb, orc = make_games(o1, o2)
    ts = [TreeSearchBatch(game_id, n, **o1), TreeSearchBatch(game_id, n, **o2)]
    used = [0, 0]
    for mv in range(nmoves):
        pl = side_of(bb)
        term = bb.terminated().cpu().numpy().astype(bool)
        move = torch.full((n,), -1, dtype=torch.int16, device="cuda")
        for k in (0, 1):
            ts[k].set_active(torch.tensor(((pl == k) & ~term).astype(np.uint8), device="cuda"))
            assert ts[k].set_roots(bb).cpu().numpy().all()
            ts[k].run_uniform()
            ts[k].L.elfhip_mcts_pick_moves(ts[k].h, None, 0, None, move.data_ptr(), None, None)
        got = move.cpu().numpy()
        for i in range(n):
            if term[i]:
                assert got[i] == -1
                continue
            k = int(pl[i])
            r = run_oracle(orc[k][i], games, i)
            used[k] += 1
            assert got[i] == r["best"], (mv, i, k)
            forward(games, i, r["best"])
            for kk in (0, 1):
                orc[kk][i].advance(r["best"])
        bb.forward(move)
        for k in (0, 1):
            ts[k].treeAdvance(move)
        assert export_ok(bb, games)
    assert used[0] > 0 and used[1] > 0
    for k in (0, 1):
        st, flags = ts[k].stats()
        assert not (flags.cpu().numpy() & ~32).any()
        tot = {key: sum(o.stats()[key] for o in orc[k]) for key in ("rollouts", "evals", "depth_sum", "edge_sum")}
        assert st == tot, (k, st, tot)


@pytest.mark.parametrize("game", ["russian", "ugolki"])
def test_two_ais_other_checkers_games(game):
    import torch
    from elf_amd.boards import BoardBatch
    from tests._oracle import FlatGame
    gid, n = dict(russian=1, ugolki=2)[game], 6
    meta_word = 4 if game == "russian" else 3

    def make(o1, o2):
        games = [FlatGame(game) for _ in range(n)]
        bb = BoardBatch(gid, n)
        rng = np.random.default_rng(3)
        for ply in range(4):
            acts = np.full(n, -1, dtype=np.int16)
            for i, g in enumerate(games):
                if ply >= i % 5:
                    continue
                legal = g.legal()
                acts[i] = legal[rng.integers(len(legal))]
                assert g.forward(acts[i])
            bb.forward(torch.tensor(acts, device="cuda"))
        orc = [[OracleMcts(policy_kind=0, game=game, **_oopts(o)) for _ in range(n)] for o in (o1, o2)]
        return games, bb, orc

    _two_ai_generic(gid, n, 14, make,
                    side_of=lambda bb: (bb.export_states()[:, meta_word] & np.uint64(1)).astype(np.uint8),
                    run_oracle=lambda o, games, i: o.run(games[i]),
                    forward=lambda games, i, a: games[i].forward(a),
                    export_ok=lambda bb, games: np.array_equal(bb.export_states(), np.stack([g.board() for g in games])),
                    na=None, edge_cap=128)


def test_two_ais_go():
    import torch
    from elf_amd.boards import BoardBatch, GO19
    from tests._oracle import GoBoards
    n = 4

    def make(o1, o2):
        games = GoBoards(n)
        bb = BoardBatch(GO19, n)
        rng = np.random.default_rng(5)
        for ply in range(6):
            acts = np.full(n, -1, dtype=np.int16)
            for i in range(n):
                if ply >= (i * 3) % 7:
                    continue
                legal = [a for a in games.legal(i) if a != 361]
                acts[i] = legal[rng.integers(len(legal))]
                assert games.forward(acts[i], i)
            bb.forward(torch.tensor(acts, device="cuda"))
        orc = [[OracleMcts(policy_kind=0, game="go", **_oopts(o)) for _ in range(n)] for o in (o1, o2)]
        return games, bb, orc

    _two_ai_generic(GO19, n, 8, make,
                    side_of=lambda bb: (bb.export_states()[:, 20] & np.uint64(1)).astype(np.uint8),
                    run_oracle=lambda o, games, i: o.run(games.ptr(i)),
                    forward=lambda games, i, a: games.forward(a, i),
                    export_ok=lambda bb, games: np.array_equal(bb.export_states(), games.export_all()),
                    na=None, edge_cap=362)
