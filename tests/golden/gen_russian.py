"""Generate tests/golden/russian_ref.npz from the COMPILED REFERENCE (oracle/_ref/libref_russian.so):
seeded random-legal games through ClearBoard / GetValidMovesBinary / CheckersPlay / GetObservation.
Per ply: state (5 u64, layout oracle/russian.h) BEFORE the move, 280-bit mask, action, is_over,
observations for both players; plus the i_to_m action table.  Run in the build container only."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import _oracle  # noqa: E402

SEED, N_GAMES = 20240608, 60


def main():
    R = _oracle.ref_russian()
    assert R is not None
    rng = np.random.default_rng(SEED)
    states, masks, actions, overs, obs, games = [], [], [], [], [], []
    for g in range(N_GAMES):
        s = np.zeros(5, dtype=np.uint64); R.ref_ru_clear(s)
        while True:
            v = np.zeros(280, dtype=np.int32); R.ref_ru_valid_moves(s, v)
            legal = np.nonzero(v)[0]
            m = np.zeros(5, dtype=np.uint64)
            for a in legal:
                m[a >> 6] |= np.uint64(1) << np.uint64(a & 63)
            o = np.zeros((2, 64), dtype=np.int32)
            R.ref_ru_observation(s, 0, o[0]); R.ref_ru_observation(s, 1, o[1])
            over = R.ref_ru_is_over(s)
            ply = (int(s[4]) >> 8) & 0xFFFF
            states.append(s.copy()); masks.append(m); overs.append(over); obs.append(o.astype(np.int8)); games.append(g)
            if over or ply >= 250:
                actions.append(-1)
                break
            a = int(rng.choice(legal)); actions.append(a)
            R.ref_ru_play(s, a)
    fr, to = np.zeros(280, dtype=np.int32), np.zeros(280, dtype=np.int32)
    for a in range(280):
        x, y = C.c_int(), C.c_int()
        R.ref_ru_action(a, C.byref(x), C.byref(y)); fr[a], to[a] = x.value, y.value
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "russian_ref.npz")
    np.savez_compressed(out, state=np.array(states, dtype=np.uint64), mask=np.array(masks, dtype=np.uint64),
                        action=np.array(actions, dtype=np.int16), over=np.array(overs, dtype=np.uint8),
                        obs=np.array(obs, dtype=np.int8), game=np.array(games, dtype=np.int32), action_from=fr, action_to=to,
                        meta=np.array(["reference: ARVILab/ELF russian_checkers CheckersBoard.cc; g++ 11.4.0; seed %d" % SEED]))
    print("wrote", out, "plies:", len(states))


if __name__ == "__main__":
    main()
