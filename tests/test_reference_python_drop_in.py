"""The reference's own Python package against the facade modules (`_elf`, `_elfgames_american_checkers`), in the
build container only: /root/reference never travels to the GPU box, so the test is skipped wherever it is absent.

A child interpreter gets PYTHONPATH = <reference>/src_py : <reference>/scripts/elfgames/american_checkers/py (where
`server_addrs.py` lives, as when selfplay.py is run from that directory) : elf_amd/pybind, and PYTHONDONTWRITEBYTECODE=1
so that nothing is written under the read-only reference.  What is verified, and nothing more:
  import elf (star-import of _elf, PyOptionSpec / PyOptionMap subclassing _elf._options, elf.logging over
  _elf._logging), `from elf import GCWrapper, ContextArgs`, import elfgames.american_checkers.game,
  Loader.get_option_spec(), PyOptionSpec.parse() of selfplay.py-style flags, Loader(option_map)._set_params() - the
  reference's code filling OUR ContextOptions / GameOptions / TSOptions field by field and constructing OUR GameContext
  - and getParams().  GCWrapper itself pins its tensors with .pin_memory() (utils_elf.py:42-50), which needs a GPU:
  that last step runs on the GPU box through tests/_gcwrapper.py's stand-in instead.
`pyzmq` is not installed in this image and elf/zmq_util.py imports it at load time: the child puts an EMPTY module
named zmq into sys.modules first (nothing on this path uses it)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src_py", "elf")), reason="reference not present (GPU box)")

CHILD = r'''
import json, sys, types
sys.modules["zmq"] = types.ModuleType("zmq")
import elf
from elf import GCWrapper, ContextArgs
from elf.options import PyOptionSpec, PyOptionMap
import elf.logging as lg
import _elf
out = {}
out["subclass"] = issubclass(PyOptionSpec, _elf._options.OptionSpec) and issubclass(PyOptionMap, _elf._options.OptionMap)
import elfgames.american_checkers.game as game
spec = game.Loader.get_option_spec()
out["n_options"] = len(spec.getOptionNames())
om = spec.parse(args=["--num_games", "4", "--batchsize", "8", "--mcts_threads", "1", "--mcts_rollout_per_thread", "16",
                      "--mcts_rollout_per_batch", "4", "--mcts_persistent_tree", "--mcts_use_prior", "--mcts_puct", "0.9",
                      "--mcts_virtual_loss", "3", "--mode", "selfplay", "--policy_distri_cutoff", "7"])
out["parsed"] = [om.get("num_games"), om.get("mcts_persistent_tree"), om.get("mcts_pick_method"), om.get("mcts_puct")]
clone = om.clone(); clone.loadOptionDict({"num_games": 9})
out["clone_independent"] = [om.get("num_games"), clone.get("num_games")]
loader = game.Loader(om)
co, GC, params = loader._set_params()
m = co.mcts_options
out["co"] = [co.num_games, co.batchsize, m.num_threads, m.num_rollouts_per_thread, m.num_rollouts_per_batch, m.persistent_tree,
             m.virtual_loss, m.pick_method, m.alg_opt.use_prior, round(m.alg_opt.c_puct, 6)]
out["gc"] = type(GC).__module__ + "." + type(GC).__name__
out["params"] = GC.getParams()
lvl = lg.LoggerLevel.from_str("warning")
out["levels"] = [lvl == lg.LoggerLevel.warn, lg.LoggerLevel.from_str("nope") == lg.LoggerLevel.invalid]
logger = lg.getIndexedLogger("drop-in-test-", "")
logger.info("hello from the reference's elf.logging")
out["logger"] = logger.name()
print("RESULT " + json.dumps(out))
'''


def test_reference_package_imports_and_builds_a_game_context():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",
               PYTHONPATH=os.pathsep.join([os.path.join(REF, "src_py"), os.path.join(REF, "scripts", "elfgames", "american_checkers", "py"),
                                           os.path.join(ROOT, "elf_amd", "pybind")]))
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd="/tmp", capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[7:])
    assert out["subclass"] is True
    assert out["n_options"] >= 50
    assert out["parsed"] == [4, True, "most_visited", 0.9]
    assert out["clone_independent"] == [4, 9]
    assert out["co"] == [4, 8, 1, 16, 4, True, 3, "most_visited", True, 0.9]
    assert out["gc"] == "_elfgames_american_checkers.GameContext"
    assert out["params"] == {"checkers_num_action": 170, "checkers_board_size": 8, "checkers_num_future_actions": 1, "checkers_num_planes": 36}
    assert out["levels"] == [True, True]
    assert out["logger"].startswith("drop-in-test-0")
    assert "hello from the reference's elf.logging" in p.stderr


def test_options_module_without_the_reference():
    """_elf._options / _elf._logging on their own (runs everywhere): the argparse description of each option kind
    (OptionSpec.h:164-226) and OptionMap's JSON round trip."""
    sys.path.insert(0, os.path.join(ROOT, "elf_amd", "pybind"))
    import _elf
    S, M = _elf._options.OptionSpec, _elf._options.OptionMap
    s = S()
    assert s.addIntOption("n", "count", 3) and s.addBoolOption("flag", "a flag", False) and s.addBoolOption("on", "default on", True)
    assert s.addStrOption("name", "required string") and s.addFloatListOption("xs", "floats", [0.5, 1.5])
    assert not s.addIntOption("n", "again", 4)                      # emplace keeps the first definition
    opts = {o["kwargs"]["dest"]: o for o in json.loads(s.getPythonArgparseOptionsAsJSONString())}
    assert opts["n"] == {"args": ["--n"], "kwargs": {"type": "int", "help": "count", "required": False, "dest": "n", "default": 3}}
    assert opts["flag"] == {"args": ["--flag"], "kwargs": {"help": "a flag", "dest": "flag", "action": "store_true"}}
    assert opts["on"] == {"args": ["--no_on"], "kwargs": {"help": "default on", "dest": "on", "action": "store_false"}}
    assert opts["name"]["kwargs"]["required"] is True and "default" not in opts["name"]["kwargs"]
    assert opts["xs"]["kwargs"]["nargs"] == "*" and opts["xs"]["kwargs"]["default"] == [0.5, 1.5] and opts["xs"]["kwargs"]["type"] == "float"
    t = S(); t.addInt64Option("big", "a big one", 1 << 40)
    s.merge(t)
    s.addPrefixSuffixToOptionNames("p_", "0")
    assert sorted(s.getOptionNames()) == ["p_big0", "p_flag0", "p_n0", "p_name0", "p_on0", "p_xs0"]
    m = M(s)
    m.loadJSONString('{"p_n0": 7, "extra": [1, 2]}')
    m.setAsJSONString("p_name0", '"x"')
    assert json.loads(m.getAsJSONString("p_n0")) == 7 and json.loads(m.getJSONString()) == {"p_n0": 7, "extra": [1, 2], "p_name0": "x"}
    with pytest.raises(RuntimeError, match="has not been set"):
        m.getAsJSONString("p_big0")
    m2 = M(m); m2.setAsJSONString("p_n0", "8")
    assert json.loads(m.getAsJSONString("p_n0")) == 7 and sorted(m2.getOptionSpec().getOptionNames()) == sorted(s.getOptionNames())
    lg = _elf._logging
    assert lg.LoggerLevel.from_str("error") == lg.LoggerLevel.err and lg.LoggerLevel.from_str("x") == lg.LoggerLevel.invalid
    a = lg.stderr_logger_mt("unit-test-logger")
    assert a.name() == "unit-test-logger" and a.should_log(lg.LoggerLevel.info) and not a.should_log(lg.LoggerLevel.debug)
    a.set_level(lg.LoggerLevel.err)
    assert not a.should_log(lg.LoggerLevel.warn) and lg.get("unit-test-logger") is not None
    with pytest.raises(RuntimeError):
        lg.stderr_logger_mt("unit-test-logger")
    lg.drop("unit-test-logger")
    assert lg.get("unit-test-logger") is None
