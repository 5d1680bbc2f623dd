"""Host-side mirror of the reference's GameState / BoardFeature interface, batched over a slab of
boards that lives in HBM (reference: src_cpp/elfgames/<game>/game/GameState.h:7-100,
BoardFeature.h:35-64).  Method names and meanings follow the reference (`forward`, `checkMove`,
`terminated`, `evaluateGame`, `reset`, `extract`); each works on all boards of the slab at once.

torch is used only for device memory and streams; all computation is in libelfhip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import AMERICAN, RUSSIAN, UGOLKI, GO19, F32, BF16, BITS, check  # noqa: F401


class BoardBatch:
    def __init__(self, game, n_boards, device=0):
        self.L = _lib.lib()
        self.game = game
        self.n = int(n_boards)
        self.device = torch.device("cuda", device)
        self.info = _lib.game_info(game)
        h = C.c_void_p()
        check(self.L.elfhip_boards_create(game, self.n, device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.elfhip_boards_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- GameState API, batched ------------------------------------------------------------------
    def reset(self, which=None):
        """Reset every board, or only those with which[i] != 0 ([n] uint8 cuda tensor)."""
        if which is None:
            check(self.L.elfhip_boards_reset(self.h))
        else:
            assert which.dtype == torch.uint8 and which.is_cuda and which.numel() == self.n
            check(self.L.elfhip_boards_reset_mask(self.h, which.data_ptr()))

    def set_stream(self, stream):
        check(self.L.elfhip_boards_set_stream(self.h, C.c_void_p(stream.cuda_stream if stream is not None else 0)))

    def export_states(self):
        out = np.zeros((self.n, self.info.state_words), dtype=np.uint64)
        check(self.L.elfhip_boards_export(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def import_states(self, states):
        states = np.ascontiguousarray(states, dtype=np.uint64)
        assert states.shape == (self.n, self.info.state_words), states.shape
        check(self.L.elfhip_boards_import(self.h, states.ctypes.data_as(C.c_void_p)))

    def legal_mask(self, with_terminal=False):
        """[n, mask_words] int64 tensor holding u64 bit masks (+ [n] uint8 terminated())."""
        mask = torch.empty((self.n, self.info.mask_words), dtype=torch.int64, device=self.device)
        term = torch.empty((self.n,), dtype=torch.uint8, device=self.device) if with_terminal else None
        check(self.L.elfhip_boards_legal_mask(self.h, mask.data_ptr(), term.data_ptr() if with_terminal else None))
        return (mask, term) if with_terminal else mask

    def terminated(self):
        return self.legal_mask(with_terminal=True)[1]

    def forward(self, actions):
        """actions: [n] int16 cuda tensor (negative = skip). Returns [n] uint8 ok flags."""
        assert actions.dtype == torch.int16 and actions.is_cuda and actions.numel() == self.n
        ok = torch.empty((self.n,), dtype=torch.uint8, device=self.device)
        check(self.L.elfhip_boards_step(self.h, actions.data_ptr(), ok.data_ptr()))
        return ok

    def extract(self, dtype=F32):
        """BoardFeature::extract for all boards -> [n, planes, size, size]."""
        p, s = self.info.num_planes, self.info.board_size
        if dtype == F32:
            out = torch.empty((self.n, p, s, s), dtype=torch.float32, device=self.device)
        elif dtype == BF16:
            out = torch.empty((self.n, p, s, s), dtype=torch.bfloat16, device=self.device)
        else:
            words = 1 if s == 8 else 6
            out = torch.empty((self.n, p, words), dtype=torch.int64, device=self.device)
        check(self.L.elfhip_boards_features(self.h, out.data_ptr(), dtype))
        return out

    def evaluateGame(self):
        v = torch.empty((self.n,), dtype=torch.float32, device=self.device)
        check(self.L.elfhip_boards_evaluate(self.h, v.data_ptr()))
        return v

    def playout(self, nply, seed=0x5EED0001, first_board_id=0, counter=None):
        check(self.L.elfhip_boards_playout(self.h, int(nply), C.c_uint64(seed), C.c_uint64(first_board_id),
                                           counter.data_ptr() if counter is not None else None))


def mask_to_numpy(mask):
    """int64 cuda tensor of u64 words -> numpy uint64"""
    return mask.cpu().numpy().view(np.uint64)
