"""Training-sample extraction (elf_amd/train_samples.py) against the oracle: replaying a recorded game to
ply t with the oracle's GameState gives the same planes; mcts_scores / offline_a / winner / move_idx follow
GameFeature.h:68-140 (recomputed here independently)."""
import numpy as np
import pytest

from tests._oracle import AmGame
from tests.test_gpu_selfplay import _oracle_game

pytestmark = pytest.mark.gpu


def test_std_mt19937_stream():
    from elf_amd.train_samples import std_mt19937
    g = std_mt19937(5489)
    assert [g() for _ in range(3)] == [3499211612, 581869302, 3890346734]     # std::mt19937 default-seed outputs


def test_train_batch_matches_oracle_replay():
    from elf_amd.boards import AMERICAN
    from elf_amd.train_samples import TrainSampler, std_mt19937
    games = _oracle_game(dict(num_rollouts=16, rollouts_per_batch=4, virtual_loss=1, c_puct=1.5), 0, 2)
    recs = []
    for i in range(12):
        w = games[i % 2]
        keep = len(w["moves"]) if i % 3 else 5       # some records hold policies for the first plies only
        recs.append(dict(moves=w["moves"], values=w["values"], reward=w["reward"], selfplay_ver=7 + i,
                         policies=[bytes(p) for p in w["policies"][:keep]]))
    ts = TrainSampler(AMERICAN, len(recs), num_future_actions=1)
    rng = std_mt19937(123)
    out = ts.sample(recs, rng=rng)
    ref_rng = std_mt19937(123)
    s = out["s"].cpu().numpy()
    for b, r in enumerate(recs):
        t = ref_rng() % len(r["moves"])
        assert int(out["move_idx"][b]) == t
        g = AmGame()
        for a in r["moves"][:t]:
            assert g.forward(int(a))
        assert np.array_equal(s[b], g.features()), b
        want = np.zeros(170, dtype=np.float32)
        if t < len(r["policies"]):
            p = np.frombuffer(r["policies"][t], dtype=np.uint8).astype(np.float32)
            want = p / np.float32(sum(np.float32(x) for x in p))
        else:
            want[int(r["moves"][t])] = 1.0
        assert np.array_equal(out["mcts_scores"][b].cpu().numpy(), want)
        assert int(out["offline_a"][b, 0]) == int(r["moves"][t])
        assert float(out["winner"][b]) == (1.0 if r["reward"] > 0 else -1.0)
        assert int(out["selfplay_ver"][b]) == 7 + b
        assert float(out["checkers_predicted_value"][b]) == float(r["values"][t])
    # an illegal recorded move is refused loudly
    bad = [dict(r) for r in recs]
    bad[3] = dict(bad[3], moves=np.array([169, 169, 169], dtype=np.int16), policies=[])
    with pytest.raises(ValueError):
        ts.sample(bad, move_to=[min(2, len(r["moves"]) - 1) for r in bad])
