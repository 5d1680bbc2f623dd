"""smoke leg for Go (called by __graft_entry__.smoke): small playout on cuda:0 vs the oracle."""
import numpy as np


def run():
    from elf_amd.boards import BoardBatch, GO19
    from tests._oracle import GoBoards
    n = 256
    bb = BoardBatch(GO19, n)
    bb.playout(120, seed=0x5EED0001)
    got = bb.export_states()
    g = GoBoards(n)
    g.playout(120, seed=0x5EED0001)
    assert np.array_equal(got, g.export_all()), "go playout mismatch vs oracle"
    print("smoke: go19 playout == oracle on %d boards x 120 plies" % n)
