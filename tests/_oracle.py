"""ctypes bindings to the test-only oracle libraries.

oracle/liboracle.so   - our CPU restatements (built by oracle/Makefile, travels to the GPU box)
oracle/_ref/libref.so - the compiled, unmodified reference board sources + harness (build container
                        only; may be absent elsewhere -> `ref()` returns None)

Only tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")

_oracle = None
_ref = None


def build():
    """(Re)build the oracle libraries; cheap when up to date."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)


def oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.am_clear.argtypes = [_u64p]
        L.am_legal_mask.argtypes = [_u64p, _u64p]
        L.am_play.argtypes = [_u64p, C.c_int]
        L.am_play.restype = C.c_int
        L.am_is_over.argtypes = [_u64p]
        L.am_is_over.restype = C.c_int
        L.am_observation.argtypes = [_u64p, C.c_int, _i8p]
        L.am_action_move.argtypes = [C.c_int]
        L.am_action_move.restype = C.c_int64
        L.am_action_dir.argtypes = [C.c_int]
        L.am_action_dir.restype = C.c_int
        L.am_playout.argtypes = [_u64p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64]
        L.am_playout.restype = C.c_uint64
        L.elf_splitmix64.argtypes = [C.c_uint64]
        L.elf_splitmix64.restype = C.c_uint64
        L.am_game_reset.argtypes = [C.c_void_p]
        L.am_game_terminated.argtypes = [C.c_void_p]
        L.am_game_terminated.restype = C.c_int
        L.am_game_forward.argtypes = [C.c_void_p, C.c_int]
        L.am_game_forward.restype = C.c_int
        L.am_game_evaluate.argtypes = [C.c_void_p]
        L.am_game_evaluate.restype = C.c_float
        L.am_game_features.argtypes = [C.c_void_p, _f32p]
        # --- Go (build-defined rules; PARITY UNPINNED vs ELF, see oracle/go.h)
        L.go_zobrist.argtypes = [C.c_int, C.c_int]
        L.go_zobrist.restype = C.c_uint64
        L.go_reset.argtypes = [C.c_void_p]
        L.go_terminated.argtypes = [C.c_void_p]
        L.go_terminated.restype = C.c_int
        L.go_legal_mask.argtypes = [C.c_void_p, _u64p]
        L.go_forward.argtypes = [C.c_void_p, C.c_int]
        L.go_forward.restype = C.c_int
        L.go_evaluate.argtypes = [C.c_void_p]
        L.go_evaluate.restype = C.c_float
        L.go_tt_score.argtypes = [C.c_void_p]
        L.go_tt_score.restype = C.c_float
        L.go_features.argtypes = [C.c_void_p, _f32p]
        L.go_export.argtypes = [C.c_void_p, _u64p]
        L.go_import.argtypes = [C.c_void_p, _u64p]
        L.go_playout.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64]
        L.go_playout.restype = C.c_uint64
        L.go_sizeof_state.restype = C.c_int
        _oracle = L
    return _oracle


def ref():
    """The compiled reference (None when oracle/_ref was not built, e.g. on the GPU box)."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libref.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.ref_am_clear.argtypes = [_u64p]
        L.ref_am_valid_moves.argtypes = [_u64p, _i32p]
        L.ref_am_play.argtypes = [_u64p, C.c_int]
        L.ref_am_play.restype = C.c_int
        L.ref_am_try_play.argtypes = [_u64p, C.c_int]
        L.ref_am_try_play.restype = C.c_int
        L.ref_am_is_over.argtypes = [_u64p]
        L.ref_am_is_over.restype = C.c_int
        L.ref_am_observation.argtypes = [_u64p, C.c_int, _i32p]
        L.ref_am_action.argtypes = [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        _ref = L
    return _ref


# ---- small numpy helpers shared by tests -------------------------------------------------------
AM_GAME_BYTES = 8 * 6 * 7 + 8  # am_state board + 6 hist + int (padded)


class AmGame:
    """Wrapper over oracle am_game (GameState restatement)."""

    def __init__(self):
        self.buf = C.create_string_buffer(AM_GAME_BYTES)
        oracle().am_game_reset(self.buf)

    def reset(self):
        oracle().am_game_reset(self.buf)

    def board(self):
        return np.frombuffer(self.buf, dtype=np.uint64, count=6).copy()

    def terminated(self):
        return bool(oracle().am_game_terminated(self.buf))

    def forward(self, a):
        return bool(oracle().am_game_forward(self.buf, int(a)))

    def evaluate(self):
        return float(oracle().am_game_evaluate(self.buf))

    def features(self):
        out = np.zeros(36 * 64, dtype=np.float32)
        oracle().am_game_features(self.buf, out)
        return out.reshape(36, 8, 8)


def mask_bits(m, n):
    """u64 words -> list of set bit indices < n"""
    return [i for i in range(n) if (int(m[i >> 6]) >> (i & 63)) & 1]


GO_PASS = 361
GO_NO_POINT = 0xFFFF


class GoBoards:
    """n oracle go_state structs in one buffer."""

    def __init__(self, n=1):
        self.L = oracle()
        self.n = n
        self.sz = self.L.go_sizeof_state()
        self.buf = C.create_string_buffer(self.sz * n)
        for i in range(n):
            self.L.go_reset(self.ptr(i))

    def ptr(self, i=0):
        return C.c_void_p(C.addressof(self.buf) + i * self.sz)

    def reset(self, i=0):
        self.L.go_reset(self.ptr(i))

    def forward(self, a, i=0):
        return bool(self.L.go_forward(self.ptr(i), int(a)))

    def play(self, r, c, i=0):
        return self.forward(r * 19 + c, i)

    def terminated(self, i=0):
        return bool(self.L.go_terminated(self.ptr(i)))

    def mask(self, i=0):
        m = np.zeros(6, dtype=np.uint64)
        self.L.go_legal_mask(self.ptr(i), m)
        return m

    def legal(self, i=0):
        return mask_bits(self.mask(i), 362)

    def export(self, i=0):
        w = np.zeros(22, dtype=np.uint64)
        self.L.go_export(self.ptr(i), w)
        return w

    def export_all(self):
        return np.stack([self.export(i) for i in range(self.n)])

    def load(self, w, i=0):
        self.L.go_import(self.ptr(i), np.ascontiguousarray(w, dtype=np.uint64))

    def features(self, i=0):
        f = np.zeros(18 * 361, dtype=np.float32)
        self.L.go_features(self.ptr(i), f)
        return f.reshape(18, 19, 19)

    def score(self, i=0):
        return float(self.L.go_tt_score(self.ptr(i)))

    def evaluate(self, i=0):
        return float(self.L.go_evaluate(self.ptr(i)))

    def playout(self, nply, seed=0x5EED0001, board_id0=0, step_base=0):
        return self.L.go_playout(self.buf, self.n, board_id0, step_base, nply, seed)


def go_words_to_grid(w):
    """export words -> 19x19 int array (0 empty, 1 black, 2 white)"""
    g = np.zeros((19, 19), dtype=np.int8)
    for r in range(19):
        for c in range(19):
            if (int(w[r]) >> c) & 1:
                g[r, c] = 1
            elif (int(w[r]) >> (32 + c)) & 1:
                g[r, c] = 2
    return g


def go_grid_to_words(g, to_move=0, ply=0):
    w = np.zeros(22, dtype=np.uint64)
    for r in range(19):
        v = 0
        for c in range(19):
            if g[r][c] == 1:
                v |= 1 << c
            elif g[r][c] == 2:
                v |= 1 << (32 + c)
        w[r] = v
    h = 0
    L = oracle()
    for r in range(19):
        for c in range(19):
            if g[r][c]:
                h ^= L.go_zobrist(int(g[r][c]) - 1, r * 19 + c)
    w[19] = h
    w[20] = (to_move & 1) | (ply << 16) | (GO_NO_POINT << 32) | (GO_NO_POINT << 48)
    return w
