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
        # --- Russian checkers
        L.ru_clear.argtypes = [_u64p]
        L.ru_legal_mask.argtypes = [_u64p, _u64p]
        L.ru_play.argtypes = [_u64p, C.c_int]
        L.ru_is_over.argtypes = [_u64p]
        L.ru_is_over.restype = C.c_int
        L.ru_observation.argtypes = [_u64p, C.c_int, _i8p]
        L.ru_action.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ru_terminated.argtypes = [_u64p]
        L.ru_terminated.restype = C.c_int
        L.ru_forward.argtypes = [_u64p, C.c_int]
        L.ru_forward.restype = C.c_int
        L.ru_evaluate.argtypes = [_u64p]
        L.ru_evaluate.restype = C.c_float
        L.ru_features.argtypes = [_u64p, _f32p]
        L.ru_playout.argtypes = [_u64p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64]
        L.ru_playout.restype = C.c_uint64
        # --- Ugolki
        L.ug_clear.argtypes = [_u64p]
        L.ug_legal_mask.argtypes = [_u64p, _u64p]
        L.ug_try_play.argtypes = [_u64p, C.c_int]
        L.ug_try_play.restype = C.c_int
        L.ug_play.argtypes = [_u64p, C.c_int]
        L.ug_is_over.argtypes = [_u64p]
        L.ug_is_over.restype = C.c_int
        L.ug_observation.argtypes = [_u64p, C.c_int, _i8p]
        L.ug_action.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ug_action.restype = C.c_int
        L.ug_terminated.argtypes = [_u64p]
        L.ug_terminated.restype = C.c_int
        L.ug_forward.argtypes = [_u64p, C.c_int]
        L.ug_forward.restype = C.c_int
        L.ug_evaluate.argtypes = [_u64p]
        L.ug_evaluate.restype = C.c_float
        L.ug_features.argtypes = [_u64p, _f32p]
        L.ug_playout.argtypes = [_u64p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64]
        L.ug_playout.restype = C.c_uint64
        # --- MCTS restatement (oracle/mcts.cpp)
        _u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
        L.omcts_create.argtypes = [_i32p, C.c_float, C.c_int, C.c_uint64]
        L.omcts_create.restype = C.c_void_p
        L.omcts_destroy.argtypes = [C.c_void_p]
        L.omcts_clear.argtypes = [C.c_void_p]
        L.omcts_advance.argtypes = [C.c_void_p, C.c_int]
        L.omcts_stats.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")]
        L.omcts_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _i32p, _u16p, _i32p, _f32p, _f32p, _i32p, _f32p, _f32p]
        L.omcts_run.restype = C.c_int
        L.omcts_scripted_policy.argtypes = [_u64p, C.c_uint64, _f32p, _f32p]
        L.ogomcts_create.argtypes = [_i32p, C.c_float, C.c_int, C.c_uint64]
        L.ogomcts_create.restype = C.c_void_p
        L.ogomcts_destroy.argtypes = [C.c_void_p]
        L.ogomcts_clear.argtypes = [C.c_void_p]
        L.ogomcts_advance.argtypes = [C.c_void_p, C.c_int]
        L.ogomcts_stats.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")]
        L.ogomcts_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _i32p, _u16p, _i32p, _f32p, _f32p, _i32p, _f32p, _f32p]
        L.ogomcts_run.restype = C.c_int
        L.ogomcts_scripted_policy.argtypes = [_u64p, C.c_uint64, _f32p, _f32p]
        for pre in ("orumcts_", "ougmcts_"):
            getattr(L, pre + "create").argtypes = [_i32p, C.c_float, C.c_int, C.c_uint64]
            getattr(L, pre + "create").restype = C.c_void_p
            getattr(L, pre + "destroy").argtypes = [C.c_void_p]
            getattr(L, pre + "clear").argtypes = [C.c_void_p]
            getattr(L, pre + "advance").argtypes = [C.c_void_p, C.c_int]
            getattr(L, pre + "stats").argtypes = [C.c_void_p, np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")]
            getattr(L, pre + "run").argtypes = [C.c_void_p, C.c_void_p, C.c_int, _i32p, _u16p, _i32p, _f32p, _f32p, _i32p, _f32p, _f32p]
            getattr(L, pre + "run").restype = C.c_int
            getattr(L, pre + "scripted_policy").argtypes = [_u64p, C.c_uint64, _f32p, _f32p]
        L.omcts_test_find_move.argtypes = [C.c_int, _u16p, _f32p, _f32p, _i32p, _f32p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                           _i32p, C.c_float, _i32p, _f32p]
        L.orng_create.argtypes = [C.c_uint]
        L.orng_create.restype = C.c_void_p
        L.orng_destroy.argtypes = [C.c_void_p]
        L.orng_next_u32.argtypes = [C.c_void_p]
        L.orng_next_u32.restype = C.c_uint
        L.orng_sample_policy.argtypes = [C.c_void_p, C.c_int, _i32p]
        L.omcts_std_sort_desc.argtypes = [C.c_int, _f32p, _u16p]
        L.omcts_std_umap_order.argtypes = [C.c_int, _u16p, _u16p]
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
        if hasattr(L, "ref_am_playout"):
            L.ref_am_playout.argtypes = [_u64p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64]
            L.ref_am_playout.restype = C.c_uint64
        _ref = L
    return _ref


_ref_ru = None


def ref_russian():
    global _ref_ru
    if _ref_ru is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libref_russian.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.ref_ru_clear.argtypes = [_u64p]
        L.ref_ru_valid_moves.argtypes = [_u64p, _i32p]
        L.ref_ru_play.argtypes = [_u64p, C.c_int]
        L.ref_ru_is_over.argtypes = [_u64p]
        L.ref_ru_is_over.restype = C.c_int
        L.ref_ru_observation.argtypes = [_u64p, C.c_int, _i32p]
        L.ref_ru_action.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _ref_ru = L
    return _ref_ru


_ref_ug = None


def ref_ugolki():
    global _ref_ug
    if _ref_ug is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libref_ugolki.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.ref_ug_clear.argtypes = [_u64p]
        L.ref_ug_valid_moves.argtypes = [_u64p, _i32p]
        L.ref_ug_play.argtypes = [_u64p, C.c_int]
        L.ref_ug_tryplay.argtypes = [_u64p, C.c_int]
        L.ref_ug_tryplay.restype = C.c_int
        L.ref_ug_is_over.argtypes = [_u64p]
        L.ref_ug_is_over.restype = C.c_int
        L.ref_ug_observation.argtypes = [_u64p, C.c_int, _i32p]
        L.ref_ug_action.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ref_ug_action.restype = C.c_int
        _ref_ug = L
    return _ref_ug


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


class FlatGame:
    """Russian checkers / Ugolki oracle state (bare u64 words; CheckersState / GameState restated in
    oracle/russian.c, oracle/ugolki.c).  `.buf` is what OracleMcts.run takes."""

    def __init__(self, game):
        self.pre = dict(russian="ru_", ugolki="ug_")[game]
        self.nw = dict(russian=5, ugolki=4)[game]
        self.na = dict(russian=280, ugolki=418)[game]
        self.mw = dict(russian=5, ugolki=7)[game]
        self.w = np.zeros(self.nw, dtype=np.uint64)
        self.reset()

    @property
    def buf(self):
        return self.w.ctypes.data_as(C.c_void_p)

    def _f(self, name):
        return getattr(oracle(), self.pre + name)

    def reset(self):
        self._f("clear")(self.w)

    def board(self):
        return self.w.copy()

    def terminated(self):
        return bool(self._f("terminated")(self.w))

    def forward(self, a):
        return bool(self._f("forward")(self.w, int(a)))

    def evaluate(self):
        return float(self._f("evaluate")(self.w))

    def legal(self):
        m = np.zeros(self.mw, dtype=np.uint64)
        self._f("legal_mask")(self.w, m)
        return mask_bits(m, self.na)

    def features(self):
        planes = 6 if self.pre == "ru_" else 16
        out = np.zeros(planes * 64, dtype=np.float32)
        self._f("features")(self.w, out)
        return out.reshape(planes, 8, 8)


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


class OracleMcts:
    """oracle/mcts.cpp: reference tree search restated (num_threads=1).  game="go": the same search over
    the build-defined Go rules (state pointer = GoBoards.ptr(i))."""

    def __init__(self, num_rollouts=256, rollouts_per_batch=1, virtual_loss=1, use_prior=1, c_puct=1.5,
                 unexplored_q_zero=0, root_unexplored_q_zero=0, persistent_tree=1, pick_method=0,
                 policy_kind=0, policy_seed=0, game="american"):
        self.L = oracle()
        self.pre = dict(american="omcts_", go="ogomcts_", russian="orumcts_", ugolki="ougmcts_")[game]
        self.cap = dict(american=64, go=362, russian=128, ugolki=96)[game]
        iopt = np.array([num_rollouts, rollouts_per_batch, virtual_loss, use_prior, unexplored_q_zero,
                         root_unexplored_q_zero, persistent_tree, pick_method], dtype=np.int32)
        self.h = C.c_void_p(getattr(self.L, self.pre + "create")(iopt, c_puct, policy_kind, policy_seed))

    POLICY_CB = C.CFUNCTYPE(None, C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_float))

    def set_callback(self, fn, state_words, num_actions):
        """policy_kind=2: fn(state_words ndarray[u64]) -> (pi ndarray[f32, num_actions], v float)"""
        def tramp(wp, pip, vp):
            w = np.ctypeslib.as_array(wp, shape=(state_words,)).copy()
            pi, v = fn(w)
            np.ctypeslib.as_array(pip, shape=(num_actions,))[:] = pi
            vp[0] = v
        self._cb = self.POLICY_CB(tramp)   # keep alive
        f = getattr(self.L, self.pre + "set_callback")
        f.argtypes = [C.c_void_p, self.POLICY_CB]
        f(self.h, self._cb)

    def __del__(self):
        if getattr(self, "h", None):
            getattr(self.L, self.pre + "destroy")(self.h)
            self.h = None

    def run(self, game):
        """game: AmGame or a go_state pointer. -> dict(best, edges=[(action, visits, reward, prior)], total_visits, root_value, best_q)"""
        n = np.zeros(1, dtype=np.int32); tv = np.zeros(1, dtype=np.int32)
        rv = np.zeros(1, dtype=np.float32); bq = np.zeros(1, dtype=np.float32)
        cap = self.cap
        acts = np.zeros(cap, dtype=np.uint16); vis = np.zeros(cap, dtype=np.int32)
        rew = np.zeros(cap, dtype=np.float32); pri = np.zeros(cap, dtype=np.float32)
        gp = game.buf if hasattr(game, "buf") else game
        best = getattr(self.L, self.pre + "run")(self.h, gp, cap, n, acts, vis, rew, pri, tv, rv, bq)
        k = int(n[0])
        return dict(best=best, edges=[(int(acts[i]), int(vis[i]), float(rew[i]), float(pri[i])) for i in range(k)],
                    total_visits=int(tv[0]), root_value=float(rv[0]), best_q=float(bq[0]))

    def set_noise(self, eps, alpha, seed):
        f = getattr(self.L, self.pre + "set_noise")
        f.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_uint]
        f(self.h, eps, alpha, seed)

    def set_policy_only(self, on):
        """the next run() calls are MCTSAI_T::actPolicyOnly (root evaluation only, edges ranked by prior)"""
        f = getattr(self.L, self.pre + "set_policy_only")
        f.argtypes = [C.c_void_p, C.c_int]
        f(self.h, int(bool(on)))

    def advance(self, a):
        getattr(self.L, self.pre + "advance")(self.h, int(a))

    def clear(self):
        getattr(self.L, self.pre + "clear")(self.h)

    def stats(self):
        o = np.zeros(4, dtype=np.int64)
        getattr(self.L, self.pre + "stats")(self.h, o)
        return dict(rollouts=int(o[0]), evals=int(o[1]), depth_sum=int(o[2]), edge_sum=int(o[3]))
