"""Generate tests/golden/american_ref.npz from the COMPILED REFERENCE (oracle/_ref/libref.so).

Run in the build container only (needs /root/reference to have been compiled by oracle/Makefile):
    python tests/golden/gen_american.py
The output is data (states, masks, observations, actions) - no reference source text.

Per recorded ply t of a seeded random-legal game driven through the reference's
ClearBoard / GetValidMovesBinary / CheckersPlay / GetObservation:
  state[t]   6 x u64   flat board words BEFORE the move (layout: oracle/american.h)
  mask[t]    3 x u64   GetValidMovesBinary as a 170-bit mask
  action[t]  i16       move played (uniform over the legal set, numpy PCG64 seed below)
  cont[t]    u8        CheckersPlay return value (1 = multi-jump continues)
  over[t]    u8        CheckersIsOver BEFORE the move
  obs[t]     2 x 64 i8 GetObservation(board, BLACK) and (board, WHITE)
  game[t]    i32       game index;  final_state[g] = state after the last move of game g
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import _oracle  # noqa: E402

SEED = 20240607
N_GAMES = 48


def main():
    R = _oracle.ref()
    assert R is not None, "oracle/_ref/libref.so missing: run `make -C oracle` in the build container"
    rng = np.random.default_rng(SEED)
    states, masks, actions, conts, overs, obs, games, finals = [], [], [], [], [], [], [], []
    for g in range(N_GAMES):
        s = np.zeros(6, dtype=np.uint64)
        R.ref_am_clear(s)
        while True:
            ply = (int(s[4]) >> 8) & 0xFFFF
            v = np.zeros(170, dtype=np.int32)
            R.ref_am_valid_moves(s, v)
            legal = np.nonzero(v)[0]
            over = R.ref_am_is_over(s)
            m = np.zeros(3, dtype=np.uint64)
            for a in legal:
                m[a >> 6] |= np.uint64(1) << np.uint64(a & 63)
            o = np.zeros((2, 64), dtype=np.int32)
            R.ref_am_observation(s, 0, o[0])
            R.ref_am_observation(s, 1, o[1])
            if over or ply >= 250:
                finals.append(s.copy())
                # record the terminal position too (action -1)
                states.append(s.copy()); masks.append(m); actions.append(-1); conts.append(0)
                overs.append(over); obs.append(o.astype(np.int8)); games.append(g)
                break
            a = int(rng.choice(legal))
            states.append(s.copy()); masks.append(m); actions.append(a); overs.append(over)
            obs.append(o.astype(np.int8)); games.append(g)
            conts.append(R.ref_am_play(s, a))
    # action table
    mv = np.zeros(170, dtype=np.int64)
    dr = np.zeros(170, dtype=np.int64)
    import ctypes as C
    for a in range(170):
        x, y = C.c_int64(), C.c_int64()
        R.ref_am_action(a, C.byref(x), C.byref(y))
        mv[a], dr[a] = x.value, y.value
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "american_ref.npz")
    np.savez_compressed(
        out,
        state=np.array(states, dtype=np.uint64), mask=np.array(masks, dtype=np.uint64),
        action=np.array(actions, dtype=np.int16), cont=np.array(conts, dtype=np.uint8),
        over=np.array(overs, dtype=np.uint8), obs=np.array(obs, dtype=np.int8),
        game=np.array(games, dtype=np.int32), final_state=np.array(finals, dtype=np.uint64),
        action_move=mv, action_dir=dr,
        meta=np.array(["reference: ARVILab/ELF american_checkers GameBoard.cc; g++ 11.4.0; seed %d" % SEED]),
    )
    print("wrote", out, "plies:", len(states), "games:", N_GAMES)


if __name__ == "__main__":
    main()
