"""Generate tests/golden/ugolki_ref.npz from the COMPILED REFERENCE (oracle/_ref/libref_ugolki.so):
seeded games through ClearBoard / GetValidMovesBinary / TryPlay / Play / IsOver / GetObservation.
Half of the games move uniformly at random (cut at 160 plies: random Ugolki never ends), half prefer
moves toward the target corner so that the win counters and terminal states are covered.
Per ply: state (4 u64, layout oracle/ugolki.h) BEFORE the move, 418-bit mask, TryPlay of 8 probe
actions, action, is_over, observations of both players.  Run in the build container only."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import _oracle  # noqa: E402

SEED, N_GAMES = 20240609, 40


def main():
    R = _oracle.ref_ugolki()
    assert R is not None
    rng = np.random.default_rng(SEED)
    bits, flags = np.zeros(418, dtype=np.uint64), np.zeros(418, dtype=np.uint64)
    for a in range(417):
        b, f = C.c_uint64(), C.c_uint64()
        assert R.ref_ug_action(a, C.byref(b), C.byref(f))
        bits[a], flags[a] = b.value, f.value
    states, masks, actions, overs, obs, games, probes, probe_ok = [], [], [], [], [], [], [], []
    finished = 0
    for g in range(N_GAMES):
        biased = g % 2 == 1
        s = np.zeros(4, dtype=np.uint64); R.ref_ug_clear(s)
        while True:
            v = np.zeros(418, dtype=np.int32); R.ref_ug_valid_moves(s, v)
            legal = np.nonzero(v)[0]
            m = np.zeros(7, dtype=np.uint64)
            for a in legal:
                m[a >> 6] |= np.uint64(1) << np.uint64(a & 63)
            o = np.zeros((2, 64), dtype=np.int32)
            R.ref_ug_observation(s, 0, o[0]); R.ref_ug_observation(s, 1, o[1])
            over = R.ref_ug_is_over(s)
            ply = (int(s[3]) >> 8) & 0xFFFF
            pr = rng.integers(0, 417, size=8).astype(np.int16)
            if len(legal):
                pr[0] = legal[rng.integers(len(legal))]
            pr[1] = 416
            po = np.array([R.ref_ug_tryplay(s, int(a)) for a in pr], dtype=np.uint8)
            states.append(s.copy()); masks.append(m); overs.append(over); obs.append(o.astype(np.int8)); games.append(g)
            probes.append(pr); probe_ok.append(po)
            if over or ply >= 500 or (not biased and ply >= 160) or len(legal) == 0:
                actions.append(-1)
                finished += int(over)
                break
            if biased and rng.random() < 0.85:
                want = 0 if (int(s[3]) & 1) == 0 else 1        # black heads for low bits, white for high bits
                pref = [a for a in legal if a != 416 and (int(flags[a]) & 1) == want]
                a = int(rng.choice(pref)) if pref else int(rng.choice(legal))
            else:
                a = int(rng.choice(legal))
            actions.append(a)
            assert R.ref_ug_tryplay(s, a)
            R.ref_ug_play(s, a)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ugolki_ref.npz")
    np.savez_compressed(out, state=np.array(states, dtype=np.uint64), mask=np.array(masks, dtype=np.uint64),
                        action=np.array(actions, dtype=np.int16), over=np.array(overs, dtype=np.uint8),
                        obs=np.array(obs, dtype=np.int8), game=np.array(games, dtype=np.int32),
                        probe=np.array(probes, dtype=np.int16), probe_ok=np.array(probe_ok, dtype=np.uint8),
                        action_bits=bits, action_flags=flags,
                        meta=np.array(["reference: ARVILab/ELF ugolki GameBoard.cc; g++ 11.4.0; seed %d" % SEED]))
    print("wrote", out, "plies:", len(states), "finished games:", finished)


if __name__ == "__main__":
    main()
