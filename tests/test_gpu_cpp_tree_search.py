"""include/elfhip_tree_search.hpp (the C++ face for code written against elf::ai::tree_search) driven by a
host C++ program (tests/cpp/tree_search_known_answer.cc, built by build()): run -> best action -> forward ->
treeAdvance over 12 moves must reproduce the COMPILED reference's known answers."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "tree_search_known_answer.bin")


def test_cpp_program_is_built():
    assert os.path.exists(BIN), "run __graft_entry__.build()"


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 8])
def test_cpp_tree_search_reproduces_reference_known_answers(golden_dir, batch):
    gold = json.load(open(os.path.join(golden_dir, "mcts_appendix_c.json")))["batch%d" % batch]
    out = subprocess.run([BIN, str(batch)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().split("\n")
    actions, totals = [], []
    for mv in range(12):
        head, edges = lines[mv].split("|")
        a, tot = head.split(":")[1].split()
        actions.append(int(a)); totals.append(int(tot))
        if mv < 3:
            got = [[int(x) for x in e.split(":")] for e in edges.split()]
            assert got == gold["move%d" % mv], mv
    assert actions == gold["first12_actions"] and totals == gold["total_visits"]
    assert lines[12] == "mismatch_throws: 1"
