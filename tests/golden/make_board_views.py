"""Generates tests/golden/board_views.json: (state words -> 8x8 board) pairs taken from the COMPILED REFERENCE
(oracle/_ref: GetObservation(board, BLACK_PLAYER), i.e. what ClientGameSelfPlay::getBoard returns) along seeded
random playouts of the three checkers-family games.  Run here (needs /root/reference built into oracle/_ref)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import _oracle  # noqa: E402


def walk(clear, valid, play, over, observe, words, nact, seed, every):
    rng = np.random.default_rng(seed)
    out = []
    for g in range(6):
        w = np.zeros(words, dtype=np.uint64)
        clear(w)
        for ply in range(120):
            if ply % every == 0:
                obs = np.zeros(64, dtype=np.int32)
                observe(w, 0, obs)
                out.append({"words": [int(x) for x in w], "board": obs.reshape(8, 8).tolist()})
            if over(w):
                break
            v = np.zeros(nact, dtype=np.int32)
            valid(w, v)
            legal = np.flatnonzero(v)
            if len(legal) == 0:
                break
            play(w, int(rng.choice(legal)))
    return out


def main():
    am, ru, ug = _oracle.ref(), _oracle.ref_russian(), _oracle.ref_ugolki()
    assert am and ru and ug, "build oracle/_ref first (__graft_entry__.build() in the container that has /root/reference)"
    gold = {
        "american": walk(am.ref_am_clear, am.ref_am_valid_moves, am.ref_am_play, am.ref_am_is_over, am.ref_am_observation, 6, 170, 11, 7),
        "russian": walk(ru.ref_ru_clear, ru.ref_ru_valid_moves, ru.ref_ru_play, ru.ref_ru_is_over, ru.ref_ru_observation, 5, 280, 12, 7),
        "ugolki": walk(ug.ref_ug_clear, ug.ref_ug_valid_moves, ug.ref_ug_play, ug.ref_ug_is_over, ug.ref_ug_observation, 4, 418, 13, 9),
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "board_views.json")
    json.dump(gold, open(path, "w"))
    print({k: len(v) for k, v in gold.items()}, os.path.getsize(path))


if __name__ == "__main__":
    main()
