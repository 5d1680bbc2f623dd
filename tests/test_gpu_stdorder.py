"""The search kernels compute std::unordered_map's iteration order with a whole wave
(stdorder::umap_iteration_order_wave).  Here that device code is compared with the real library
(through oracle/mcts.cpp) at every bucket-count boundary and on random key sets."""
import ctypes as C

import numpy as np
import pytest

from elf_amd import _lib
from tests import _oracle

pytestmark = pytest.mark.gpu


def _pair(keys):
    keys = np.ascontiguousarray(keys, dtype=np.uint16)
    m = len(keys)
    a = np.zeros(m, dtype=np.uint16); b = np.zeros(m, dtype=np.uint16)
    _lib.check(_lib.lib().elfhip_test_umap_order_wave(m, keys.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), 0))
    _oracle.oracle().omcts_std_umap_order(m, keys, b)
    return a, b


def test_wave_umap_order_matches_libstdcxx():
    rng = np.random.default_rng(5)
    for m in [0, 1, 2, 7, 12, 13, 14, 15, 28, 29, 30, 31, 58, 59, 60, 61, 63, 64, 65, 126, 127, 128, 129, 170, 256, 257, 258, 300, 361,
              362, 418, 511, 512]:
        for hi in [m, 170, 280, 362, 418, 65536]:
            if hi < m:
                continue
            for rep in range(3):
                keys = rng.permutation(hi)[:m]
                a, b = _pair(keys)
                assert np.array_equal(a, b), (m, hi, rep)
    # ascending and descending keys (every bucket hit in turn), and keys that all fall in one bucket of each size
    for m in [13, 29, 59, 127, 257, 362]:
        for keys in (np.arange(m), np.arange(m)[::-1], np.arange(m) * 13, np.arange(m) * 29 % 65536, np.arange(m) * 127):
            if len(set(keys.tolist())) != m or keys.max() > 65535:
                continue
            a, b = _pair(keys)
            assert np.array_equal(a, b), m
