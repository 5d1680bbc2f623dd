"""In-tree build of libelfhip.so (hipcc, gfx950 only).  Used by __graft_entry__.build()."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libelfhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-inline-asm"]   # inline-asm: the LDS-DMA helper names m0 as clobbered


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _newer(s, o) or any(_newer(h, o) for h in hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % s)
    if force or procs or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    build_pybind(force=force or bool(procs), verbose=verbose)
    build_cpp_tests(force=force or bool(procs), verbose=verbose)
    return LIB


def build_cpp_tests(force=False, verbose=False):
    """tests/cpp/*.cc: host C++ programs over include/elfhip_tree_search.hpp (run by the GPU tests)."""
    root = os.path.join(HERE, "..")
    hdrs = glob.glob(os.path.join(root, "include", "*"))
    for src in sorted(glob.glob(os.path.join(root, "tests", "cpp", "*.cc"))):
        out = src[:-3] + ".bin"
        if force or _newer(src, out) or any(_newer(h, out) for h in hdrs):
            cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                   "-I" + os.path.join(root, "include"), src, "-o", out, "-L" + HERE, "-lelfhip", "-L/opt/rocm/lib", "-lamdhip64",
                   "-Wl,-rpath,$ORIGIN/../../elf_amd", "-Wl,-rpath,/opt/rocm/lib"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)


# Python-visible facade modules with the reference's module names (host C++ only, g++):
# _elf and one _elfgames_<game> per game that has search kernels.
PYB = os.path.join(HERE, "pybind")
# _elfgames_go: the same GameContext surface over the build-defined Go 19x19 rules (the fork has no Go module;
# the north star names the OpenGo configuration, BASELINE.json configs[0] / configs[3])
GAME_MODULES = {"_elfgames_american_checkers": 0, "_elfgames_russian_checkers": 1, "_elfgames_ugolki": 2, "_elfgames_go": 3}


def build_pybind(force=False, verbose=False):
    import sysconfig
    import pybind11
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    inc = ["-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"], "-I/opt/rocm/include"]
    base = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-D__HIP_PLATFORM_AMD__", "-Wall"] + inc
    link = ["-L" + HERE, "-lelfhip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
    hdrs = [os.path.join(PYB, "facade.h"), os.path.join(PYB, "board_view.h"), os.path.join(HERE, "..", "include", "elfhip.h")]
    jobs = [("_elf", os.path.join(PYB, "elf_module.cc"), [], [])]
    for name, gid in GAME_MODULES.items():
        jobs.append((name, os.path.join(PYB, "game_module.cc"), ["-DELFGAME_MODULE=" + name, "-DELFGAME_ID=%d" % gid], link))
    procs = []
    for name, src, defs, lk in jobs:
        out = os.path.join(PYB, name + suffix)
        if force or _newer(src, out) or any(_newer(h, out) for h in hdrs):
            cmd = base + defs + [src, "-o", out] + lk
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((name, subprocess.Popen(cmd)))
    for name, p in procs:
        if p.wait() != 0:
            raise RuntimeError("g++ failed on pybind module %s" % name)


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
