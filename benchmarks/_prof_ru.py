# in-kernel section cycles of the Russian playout kernel (library built with -DRU_PROF by _build_prof_lib.sh; timing only)
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import torch
from elf_amd.boards import BoardBatch, RUSSIAN
from elf_amd import _lib
if os.environ.get("ELFHIP_PROF_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ELFHIP_PROF_LIB"])
L = _lib.lib()
bb = BoardBatch(RUSSIAN, 65536)
out = (C.c_ulonglong * 16)()
names = {0: "pawn jumps", 1: "king jump pretest", 2: "king jump loop", 3: "pawn steps", 4: "king slides", 5: "moves total", 6: "count+select", 7: "play"}
for phase in range(8):
    L.elfhip_debug_ru_prof(out, 1)
    torch.cuda.synchronize()
    bb.playout(64, seed=0x5EED0001, first_board_id=0)
    torch.cuda.synchronize()
    L.elfhip_debug_ru_prof(out, 0)
    v = list(out)
    ws = 1024 * 64
    print("launch", phase, {names[k]: round(v[k] / ws) for k in names}, "king-jump iters/step %.2f slide groups/step %.2f" % (v[9] / ws, v[10] / ws))
