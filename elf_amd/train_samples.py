"""Training-sample extraction: the step AFTER the self-play path.  The reference
replays a recorded game to a random ply on a CPU thread and fills one row of the `train` batch
(GameStateExtOffline::fromRecord / switchRandomMove / switchBeforeMove, game/GameStateExt.h:144-190;
GameFeature::extractMCTSPi / extractOfflineAction / extractWinner / extractMoveIdx,
common/GameFeature.h:68-140; batch keys game.py:399-413).  Here a whole batch of records is replayed
at once on a board slab in HBM (reset, then one masked `forward` per ply) and the planes come from the
feature kernel; the per-row scalars are array arithmetic on the records.

Records are the dicts `GameContext.takeRecords()` / `SelfPlayBatch` produce: moves (int16), policies
(list of bytes / uint8 rows, possibly fewer than moves), values, reward, and optionally selfplay_ver."""
import numpy as np
import torch

from .boards import BoardBatch


def std_mt19937(seed):
    """raw 32-bit outputs of std::mt19937(seed) (numpy's legacy seeding is init_genrand)"""
    bg = np.random.RandomState(seed)._bit_generator
    return lambda: int(bg.random_raw())


class TrainSampler:
    def __init__(self, game, batchsize, device=0, num_future_actions=1):
        self.bb = BoardBatch(game, batchsize, device)
        self.B = int(batchsize)
        self.K = int(num_future_actions)
        self.A = self.bb.info.num_actions
        self.device = self.bb.device

    def sample(self, records, rng=None, move_to=None):
        """records: B record dicts.  move_to: optional explicit plies (switchBeforeMove); otherwise
        move_to = rng() % (#moves - num_future_actions + 1) per record in order (switchRandomMove).
        Returns the `train` batch: s, offline_a, winner, mcts_scores, move_idx, selfplay_ver
        (+ num_move, predicted_value)."""
        B, K, A = self.B, self.K, self.A
        assert len(records) == B
        n_moves = np.array([len(r["moves"]) for r in records], dtype=np.int64)
        if (n_moves <= K - 1).any():
            raise ValueError("#moves smaller than num_future_actions - 1")       # the reference exits here
        if move_to is None:
            move_to = np.array([rng() % (int(n) - K + 1) for n in n_moves], dtype=np.int64)
        move_to = np.asarray(move_to, dtype=np.int64)
        assert (move_to >= 0).all() and (move_to < n_moves).all()
        # switchBeforeMove: reset, then forward moves[0 .. move_to)
        self.bb.reset()
        steps = int(move_to.max()) if B else 0
        acts = np.full((steps, B), -1, dtype=np.int16)
        for b, r in enumerate(records):
            m = np.asarray(r["moves"], dtype=np.int16)
            acts[: move_to[b], b] = m[: move_to[b]]
        acts_d = torch.from_numpy(acts).to(self.device)
        ok_all = torch.ones(B, dtype=torch.uint8, device=self.device)
        for i in range(steps):
            ok = self.bb.forward(acts_d[i])
            ok_all &= ok | (acts_d[i] < 0).to(torch.uint8)
        if not bool(ok_all.all().item()):
            raise ValueError("a recorded move was rejected by forward(): record does not belong to this game")
        s = self.bb.extract()
        mcts = np.zeros((B, A), dtype=np.float32)
        offline_a = np.zeros((B, K), dtype=np.int64)
        winner = np.zeros(B, dtype=np.float32)
        ver = np.zeros(B, dtype=np.int64)
        pv = np.zeros(B, dtype=np.float32)
        for b, r in enumerate(records):
            t = int(move_to[b])
            pol = r.get("policies") or []
            if t < len(pol):                                             # extractMCTSPi
                p = np.frombuffer(bytes(pol[t]), dtype=np.uint8).astype(np.float32)
                sum_v = np.float32(0.0)
                for x in p:                                              # float accumulation in index order
                    sum_v = np.float32(sum_v + x)
                mcts[b] = p / sum_v
            else:
                mcts[b, int(r["moves"][t])] = 1.0
            offline_a[b] = np.asarray(r["moves"][t: t + K], dtype=np.int64)   # extractOfflineAction
            winner[b] = 1.0 if r["reward"] > 0 else -1.0                 # fromRecord
            ver[b] = int(r.get("selfplay_ver", -1))                      # request.vers.black_ver
            vals = r.get("values")
            pv[b] = float(vals[t]) if vals is not None and t < len(vals) else 0.0
        dev = self.device
        return dict(s=s, offline_a=torch.from_numpy(offline_a).to(dev), winner=torch.from_numpy(winner).to(dev),
                    mcts_scores=torch.from_numpy(mcts).to(dev), move_idx=torch.from_numpy(move_to.astype(np.int32)).to(dev),
                    selfplay_ver=torch.from_numpy(ver).to(dev), checkers_num_move=torch.from_numpy(n_moves.astype(np.int32)).to(dev),
                    checkers_predicted_value=torch.from_numpy(pv).to(dev))
