"""BASELINE.json configs[0] on the GPU: Go 19x19 self-play, 64 games, 64 rollouts/move, tiny
2-block x 32 ResNet (bf16 MFMA forward in libelfhip).  "MCTS visit counts must be bit-exact given
identical network outputs": every (leaf state -> pi, V) the HIP network produced is recorded and
replayed to the oracle search through its evaluator callback; edges, visits, rewards and chosen
moves must be identical for all 64 games."""
import numpy as np
import pytest

from tests import _oracle
from tests._oracle import GoBoards, OracleMcts

pytestmark = pytest.mark.gpu


def test_c1_go_selfplay_visits_match_oracle_given_same_network_outputs():
    import torch
    from elf_amd.boards import BoardBatch, GO19
    from elf_amd.mcts import TreeSearchBatch
    from elf_amd.net import PolicyValueNet
    from oracle.model_ref import seeded_model
    n, R, B = 64, 64, 8
    cfg = dict(planes=18, board=19, dim=32, num_block=2, num_action=362)
    net = PolicyValueNet(max_batch=n * B, **cfg)
    net.load_state_dict(seeded_model(seed=0, **cfg).state_dict())
    rng = np.random.default_rng(5)
    games = GoBoards(n)
    bb = BoardBatch(GO19, n)
    for ply in range(24):   # different openings
        acts = np.full(n, -1, dtype=np.int16)
        for i in range(n):
            if ply >= i % 25:
                continue
            legal = [a for a in games.legal(i) if a != 361]
            acts[i] = legal[rng.integers(len(legal))]
            assert games.forward(acts[i], i)
        bb.forward(torch.tensor(acts, device="cuda"))
    ts = TreeSearchBatch(GO19, n, num_rollouts_per_thread=R, num_rollouts_per_batch=B, virtual_loss=1, c_puct=1.5,
                         max_nodes=256, max_edges=60000)
    oracles = [OracleMcts(num_rollouts=R, rollouts_per_batch=B, virtual_loss=1, c_puct=1.5, policy_kind=2, game="go")
               for _ in range(n)]
    table = {}

    def evaluator(feat, states, need):
        out = net.forward(feat.contiguous())
        pi, v = out["pi"], out["V"].reshape(-1)
        s = states.cpu().numpy().view(np.uint64).reshape(n * B, 22)
        nd = need.cpu().numpy().reshape(-1)
        pih, vh = pi.cpu().numpy(), v.cpu().numpy()
        for k in np.nonzero(nd)[0]:
            table[s[k].tobytes()] = (pih[k].copy(), float(vh[k]))
        return pi, v

    def lookup(w):
        return table[w.tobytes()]

    for o in oracles:
        o.set_callback(lookup, 22, 362)
    evals = 0
    for mv in range(2):
        assert ts.set_roots(bb).cpu().numpy().all()
        ts.run(evaluator)
        res = {k: v.cpu().numpy() for k, v in ts.result(edge_cap=362).items()}
        assert not ts.stats()[1].cpu().numpy().any()
        acts = np.full(n, -1, dtype=np.int16)
        for i, o in enumerate(oracles):
            r = o.run(games.ptr(i))
            k = int(res["n_edges"][i])
            got = [(int(res["edge_action"][i, j]), int(res["edge_visits"][i, j]), float(res["edge_reward"][i, j]),
                    float(res["edge_prior"][i, j])) for j in range(k)]
            assert got == r["edges"], (mv, i)
            assert int(res["best"][i]) == r["best"] and float(res["best_q"][i]) == r["best_q"]
            acts[i] = r["best"]
            assert games.forward(r["best"], i)
            o.advance(r["best"])
        act = torch.tensor(acts, device="cuda")
        assert bb.forward(act).cpu().numpy().all()
        ts.treeAdvance(act)
    st, _ = ts.stats()
    assert st["rollouts"] == 2 * n * R
    assert st["evals"] == sum(o.stats()["evals"] for o in oracles) and st["evals"] > n * R


def test_go_selfplay_driver_runs_with_network():
    """SelfPlayBatch plumbing for Go: moves are legal, records advance, the policy slab has its max at 255."""
    import torch
    from elf_amd.boards import GO19
    from elf_amd.net import PolicyValueNet
    from elf_amd.selfplay import SelfPlayBatch
    from oracle.model_ref import seeded_model
    cfg = dict(planes=18, board=19, dim=32, num_block=2, num_action=362)
    n = 32
    net = PolicyValueNet(max_batch=n * 8, **cfg)
    net.load_state_dict(seeded_model(seed=1, **cfg).state_dict())
    sp = SelfPlayBatch(GO19, n, net=net, policy_distri_cutoff=30, seed=0, num_rollouts_per_thread=32,
                       num_rollouts_per_batch=8, virtual_loss=1, c_puct=1.5, max_nodes=192, max_edges=60000)
    for step in range(6):
        sp.step()
        assert int(sp.nmoves.min().item()) == step + 1 or sp.win_stats["total_games"] > 0
    assert not sp.ts.stats()[1].cpu().numpy().any()
    pol = sp.policies[:, :6].cpu().numpy()
    assert (pol.max(axis=2) == 255).all()
    assert len(np.unique(sp.moves[:, 0].cpu().numpy())) > 1
