import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import torch
from elf_amd.boards import BoardBatch, GO19
from elf_amd import _lib
if os.environ.get("ELFHIP_PROF_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ELFHIP_PROF_LIB"])
L = _lib.lib()
bb = BoardBatch(GO19, 65536)
out = (C.c_ulonglong * 32)()
names = {0:"seed+any",1:"capture+release fill",2:"capture removal+hash",3:"group status",4:"legal+select",5:"forward total (incl place)",6:"superko+swap"}
for phase in range(6):
    L.elfhip_debug_go_prof(out, 1)
    torch.cuda.synchronize()
    bb.playout(64, seed=0x5EED0001, first_board_id=0)
    torch.cuda.synchronize()
    L.elfhip_debug_go_prof(out, 0)
    v = list(out)
    wavesteps = 1024*64
    print("launch", phase, {names[k]: round(v[k]/wavesteps) for k in names}, "fills/step %.2f sweeps/fill %.2f" % (v[20]/wavesteps, v[21]/max(1,v[20])),
          "job rates", [round(v[10+j]/wavesteps,2) for j in range(7)])
