"""CPU-side checks of the C-ABI library: loads, exports every declared symbol, argument checks that
need no GPU.  (No compute calls here.)"""
import ctypes as C
import os
import re

import pytest

from elf_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_built_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    L = _lib.lib()
    assert L.elfhip_abi_version() == 2


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "elfhip.h")).read()
    declared = set(re.findall(r"\b(elfhip_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"elfhip_game_info"}
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "missing export: " + name
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)


def test_game_info_and_errors():
    gi = _lib.game_info(_lib.AMERICAN)
    assert (gi.num_actions, gi.mask_words, gi.state_words, gi.num_planes, gi.board_size, gi.max_ply) == (170, 3, 6, 36, 8, 250)
    L = _lib.lib()
    assert L.elfhip_game_get_info(99, C.byref(_lib.GameInfo())) == -1
    assert b"unknown" in L.elfhip_last_error()
    h = C.c_void_p()
    assert L.elfhip_boards_create(_lib.AMERICAN, 0, 0, C.byref(h)) == -1
    assert L.elfhip_boards_reset(None) == -1
    assert L.elfhip_boards_count(None) == -1


def test_random_state_dict_matches_the_documented_parameter_layout():
    """elf_amd.net.random_state_dict (seeded weights for benchmarks) produces every tensor elfhip_net_load
    expects, in shapes that flatten to elfhip_net_param_count (no GPU needed: counting only)."""
    import ctypes as C
    from elf_amd import _lib
    from elf_amd.net import pack_state_dict, random_state_dict, state_dict_order
    L = _lib.lib()
    for cfg in (dict(planes=36, board=8, dim=32, num_block=2, num_action=170), dict(planes=18, board=19, dim=64, num_block=1, num_action=362)):
        sd = random_state_dict(seed=3, **cfg)
        assert set(sd) == set(state_dict_order(cfg["num_block"]))
        blob = pack_state_dict(sd, cfg["num_block"])
        c = _lib.NetConfig(cfg["planes"], cfg["board"], cfg["dim"], cfg["num_block"], cfg["num_action"], 1e-5, 0.0)
        assert blob.size == L.elfhip_net_param_count(C.byref(c))
        sd2 = random_state_dict(seed=3, **cfg)
        assert all((sd[k] == sd2[k]).all() for k in sd)      # seeded: reproducible
