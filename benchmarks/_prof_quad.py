# wave cycles per phase of the four-trees-per-wave search kernel (library built with -DQUAD_PROF; timing only)
import ctypes as C, sys, os, json
sys.path.insert(0, os.getcwd())
import torch
torch.cuda.init()
from elf_amd import _lib
if os.environ.get("ELFHIP_PROF_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ELFHIP_PROF_LIB"])
import importlib.util
spec = importlib.util.spec_from_file_location("bench_mcts", os.path.join("benchmarks", "bench_mcts.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
L = _lib.lib()
out = (C.c_ulonglong * 8)()
L.elfhip_debug_quad_prof(out, 1)
r = m.measure(moves=8)
L.elfhip_debug_quad_prof(out, 0)
v = list(out); tot = sum(v[:3])
print(json.dumps({"rollouts_per_sec": r["value"], "select": v[0] / tot, "expand": v[1] / tot, "backup": v[2] / tot}))
