"""The HIP search kernels re-implement libstdc++'s std::sort and unordered_map iteration order
(elf_amd/csrc/stdorder.h).  Here the host build of that same code is compared with the real
library (through oracle/mcts.cpp) on tie-heavy and adversarial inputs, and with SURVEY Appendix C."""
import ctypes as C
import json
import os

import numpy as np

from elf_amd import _lib
from tests import _oracle


def _sort_pair(keys):
    keys = np.ascontiguousarray(keys, dtype=np.float32)
    n = len(keys)
    a = np.zeros(n, dtype=np.uint16); b = np.zeros(n, dtype=np.uint16)
    _lib.check(_lib.lib().elfhip_test_sort_order(n, keys.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p)))
    _oracle.oracle().omcts_std_sort_desc(n, keys, b)
    return a, b


def _umap_pair(keys):
    keys = np.ascontiguousarray(keys, dtype=np.uint16)
    m = len(keys)
    a = np.zeros(m, dtype=np.uint16); b = np.zeros(m, dtype=np.uint16)
    _lib.check(_lib.lib().elfhip_test_umap_order(m, keys.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p)))
    _oracle.oracle().omcts_std_umap_order(m, keys, b)
    return a, b


def test_appendix_c_known_answers(golden_dir):
    gold = json.load(open(os.path.join(golden_dir, "mcts_appendix_c.json")))
    a, _ = _sort_pair(np.full(170, 1.0 / 170))
    assert list(a[:40]) == gold["sort_uniform_170_first40"]
    a, _ = _umap_pair(np.arange(20))
    assert list(a) == gold["umap_keys_0_19_iteration"]
    a, _ = _umap_pair(gold["start_insertion_order"])
    assert list(a) == gold["start_iteration_order"]


def test_sort_matches_libstdcxx():
    rng = np.random.default_rng(1)
    for n in [0, 1, 2, 3, 15, 16, 17, 18, 33, 100, 170, 280, 362, 418, 1000]:
        for levels in [1, 2, 3, 8, 64, 1024, 10**9]:
            for rep in range(6):
                keys = rng.integers(0, levels, size=n).astype(np.float32) / np.float32(max(levels, 1))
                a, b = _sort_pair(keys)
                assert np.array_equal(a, b), (n, levels, rep)
    # adversarial shapes: sorted, reversed, organ pipe, median-of-3 killer (forces the heap-sort fallback)
    for n in [170, 362, 512, 4096]:
        base = np.arange(n, dtype=np.float32)
        shapes = [base, base[::-1].copy(), np.concatenate([base[: n // 2], base[: n - n // 2][::-1]])]
        k = n // 2
        killer = np.zeros(n, dtype=np.float32)
        for i in range(k):
            if i % 2 == 0:
                killer[i] = i + 1
            else:
                killer[i] = k + i + (k % 2)
            killer[k + i] = 2 * (i + 1)
        shapes += [killer, -killer]
        for s in shapes:
            a, b = _sort_pair(s)
            assert np.array_equal(a, b), n


def test_umap_order_matches_libstdcxx():
    rng = np.random.default_rng(2)
    for m in [0, 1, 7, 12, 13, 14, 15, 28, 29, 30, 31, 58, 59, 60, 61, 126, 127, 128, 129, 170, 256, 257, 258, 300, 362, 418, 512]:
        for hi in [m, 170, 280, 362, 418, 65536]:
            if hi < m:
                continue
            for rep in range(4):
                keys = rng.permutation(hi)[:m]
                a, b = _umap_pair(keys)
                assert np.array_equal(a, b), (m, hi, rep)
