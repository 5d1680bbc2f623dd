import ctypes as C, sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from elf_amd.boards import BoardBatch, AMERICAN
from elf_amd.mcts import TreeSearchBatch
from elf_amd import _lib
L = _lib.lib()
n = 16384
bb = BoardBatch(AMERICAN, n)
ts = TreeSearchBatch(AMERICAN, n, num_rollouts_per_thread=256, num_rollouts_per_batch=8, virtual_loss=1, c_puct=1.5)
out = (C.c_ulonglong * 8)()
for mv in range(6):
    ts.set_roots(bb)
    L.elfhip_debug_mcts_prof(out, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ts.run_uniform()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    L.elfhip_debug_mcts_prof(out, 0)
    v = list(out); nb = n * 32
    print("move", mv, "ms %.1f" % (dt * 1e3), "cycles per wave-batch(8 rollouts): select %d expand %d backup %d" % (v[0] / nb, v[1] / nb, v[2] / nb))
    r = ts.result()
    ok = bb.forward(r["best"]); ts.treeAdvance(r["best"])
