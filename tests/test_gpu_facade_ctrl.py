"""The facade's request handling and per-colour players on GPU:
  * two models = two AIs with their own trees, white overrides (white_puct, white_mcts_rollout_per_*),
    player_swap (ClientGameSelfPlay::restart, common/ClientGameSelfPlay.cc:155-196);
  * black_use_policy_network_only -> actPolicyOnly (:340-352);
  * ClientGameSelfPlay::OnReceive's decisions for a MsgRequest (:198-247);
  * mode "play": the human side through the "actor_white" batch (:273-305).
Records are compared with oracle games driven by the same (uniform) evaluator."""
import json

import numpy as np
import pytest

from tests import _gcwrapper, _oracle
from tests._oracle import AmGame, OracleMcts

pytestmark = pytest.mark.gpu

BASE = dict(num_rollouts=24, rollouts_per_batch=4, virtual_loss=1, c_puct=1.5)


def _ctx(n_games, batchsize, mode="selfplay", **gopts):
    import _elfgames_american_checkers as ac
    co = ac.ContextOptions()
    co.num_games, co.batchsize = n_games, batchsize
    m = co.mcts_options
    m.num_threads, m.num_rollouts_per_thread, m.num_rollouts_per_batch = 1, BASE["num_rollouts"], BASE["rollouts_per_batch"]
    m.virtual_loss, m.alg_opt.c_puct, m.persistent_tree = 1, BASE["c_puct"], True
    go = ac.GameOptions()
    go.mode, go.policy_distri_cutoff, go.policy_distri_training_for_all = mode, 0, True
    for k, v in gopts.items():
        setattr(go, k, v)
    GC = ac.GameContext(co, go)
    desc = dict(game_end=dict(batchsize=1), game_start=dict(batchsize=1, input=["white_ver", "black_ver"], reply=None),
                actor_white=dict(input=["s"], reply=["pi", "V", "a", "rv"], batchsize=batchsize),
                actor_black=dict(input=["s"], reply=["pi", "V", "a", "rv"], batchsize=batchsize))
    return GC, _gcwrapper.Wrapper(GC, batchsize, desc, num_recv=2)


def _uniform(batch, bs):
    return dict(pi=np.full((bs, 170), np.float32(1.0 / 170.0), dtype=np.float32), V=np.zeros(bs, dtype=np.float32),
                a=np.zeros(bs, dtype=np.int64), rv=np.zeros(bs, dtype=np.int64))


def _quant(edges):
    vis = np.array([e[1] for e in edges], dtype=np.float32)
    p = vis / np.float32(vis.sum())
    pol = np.zeros(170, dtype=np.uint8)
    q = (p / p.max() * np.float32(255)).astype(np.float32)
    for (a, _, _, _), c in zip(edges, q):
        pol[a] = np.uint8(int(c))
    return pol


def _oracle_first_game(ai_opts, policy_only=(False, False)):
    """One game with AI ai_opts[0] for black and ai_opts[-1] for white (own trees), uniform evaluator."""
    g = AmGame()
    ais = [OracleMcts(policy_kind=0, **o) for o in ai_opts]
    moves, vals, pols = [], [], []
    while not g.terminated():
        pl = int(g.board()[4]) & 1
        ai = ais[pl if len(ais) == 2 else 0]
        ai.set_policy_only(policy_only[pl])
        r = ai.run(g)
        moves.append(r["best"]); vals.append(np.float32(r["best_q"]))
        if not policy_only[pl]:
            pols.append(_quant(r["edges"]))
        assert g.forward(r["best"])
        for a in ais:
            a.advance(r["best"])
    return dict(moves=moves, values=vals, policies=pols, reward=g.evaluate())


def _run_until_ends(w, n_ends, counts, limit=400000):
    ends = []
    w.reg_callback("game_end", lambda batch, bs: ends.append(1))
    for _ in range(limit):
        name, _ = w.run()
        counts[name] = counts.get(name, 0) + 1
        if len(ends) >= n_ends:
            return
    raise AssertionError("games did not finish")


def _check_first_games(GC, want, n):
    recs = [r for r in GC.takeRecords() if r["seq"] == 0]
    assert len(recs) == n
    for r in recs:
        assert list(r["moves"]) == want["moves"]
        assert np.array_equal(np.array(r["values"], dtype=np.float32), np.array(want["values"], dtype=np.float32), equal_nan=True)
        assert [bytes(p) for p in r["policies"]] == [p.tobytes() for p in want["policies"]]
        assert r["reward"] == want["reward"]


@pytest.mark.parametrize("swap", [False, True])
def test_two_models_own_trees_overrides_and_player_swap(swap):
    n = 3
    GC, w = _ctx(n, 16, white_puct=2.25, white_mcts_rollout_per_batch=2, white_mcts_rollout_per_thread=16)
    req = dict(vers=dict(black_ver=3, white_ver=7), client_ctrl=dict(client_type=2, num_game_thread_used=-1, player_swap=swap, **{"async": False}))
    assert GC.getClient().onRequestJson(json.dumps(req)) == "UPDATE_MODEL"
    w.reg_callback("actor_black", _uniform)
    w.reg_callback("actor_white", _uniform)
    w.start()
    info = GC.aiInfo()
    white_ai = dict(BASE, c_puct=2.25, rollouts_per_batch=2, num_rollouts=16)
    if swap:   # swap(_ai1, _ai2): the "actor_white" AI (white_ver, overrides) plays black
        assert [(d["actor"], d["version"], d["plays"]) for d in info] == [("actor_white", 7, "black"), ("actor_black", 3, "white")]
        ais = [white_ai, BASE]
    else:
        assert [(d["actor"], d["version"], d["plays"]) for d in info] == [("actor_black", 3, "black"), ("actor_white", 7, "white")]
        ais = [BASE, white_ai]
    assert info[1 if not swap else 0]["c_puct"] == 2.25 and info[1 if not swap else 0]["num_rollouts_per_batch"] == 2
    counts = {}
    _run_until_ends(w, n, counts)
    assert counts["actor_black"] > 0 and counts["actor_white"] > 0
    js = json.loads(GC.dumpRecordsJson(clear=False))
    assert all(j["request"]["client_ctrl"]["player_swap"] is swap and j["request"]["client_ctrl"]["client_type"] == 2 and
               j["result"]["using_models"] == [3, 7] for j in js)
    _check_first_games(GC, _oracle_first_game(ais), n)
    w.stop()


def test_black_policy_network_only():
    n = 3
    GC, w = _ctx(n, 16, black_use_policy_network_only=True)
    GC.getClient().setRequest(1, -1, 1)
    w.reg_callback("actor_black", _uniform)
    w.start()
    counts = {}
    _run_until_ends(w, n, counts)
    assert counts.get("actor_white", 0) == 0
    _check_first_games(GC, _oracle_first_game([BASE], policy_only=(True, False)), n)
    w.stop()


def test_on_receive_decisions():
    GC, w = _ctx(2, 16)
    cl = GC.getClient()
    w.reg_callback("actor_black", _uniform)
    w.reg_callback("actor_white", _uniform)
    starts = []
    w.reg_callback("game_start", lambda batch, bs: starts.append((int(batch["black_ver"][0]), int(batch["white_ver"][0]))))
    w.start()

    def send(black, white, swap=False, asy=False):
        return cl.onRequestJson(json.dumps(dict(vers=dict(black_ver=black, white_ver=white),
                                                client_ctrl=dict(client_type=1, num_game_thread_used=-1, player_swap=swap, **{"async": asy}))))
    assert send(4, -1) == "UPDATE_MODEL"                    # previous request was "wait": restart
    assert [w.run()[0] for _ in range(6)].count("game_start") == 1 and starts == [(4, -1)]
    assert send(4, -1) == "UPDATE_REQUEST_ONLY"             # same versions, same swap
    assert send(5, -1, asy=True) == "UPDATE_MODEL_ASYNC"    # async: new version without a restart
    assert send(5, -1, asy=True) == "UPDATE_REQUEST_ONLY"
    for _ in range(20):
        assert w.run()[0] != "game_start"
    assert send(5, -1, swap=True) == "UPDATE_MODEL"         # player_swap changed: restart
    assert [w.run()[0] for _ in range(6)].count("game_start") == 1 and starts[-1] == (5, -1)
    assert send(-1, -1) == "ONLY_WAIT"                      # the server says wait: nothing is produced
    seen_none = False
    for _ in range(200):                                    # the move in flight finishes, then wait() comes back empty
        if GC.ctx().wait() is None:
            seen_none = True
            break
        GC.ctx().step()
    assert seen_none and GC.ctx().wait() is None
    assert send(6, -1) == "UPDATE_MODEL"                    # leaving the wait state always restarts
    assert w.run()[0] == "game_start" and starts[-1] == (6, -1)
    w.stop()


def test_play_mode_human_through_actor_white():
    """Human plays black (human_plays_for = 0): its boards arrive as "actor_white" batches; the reply's `a` is the
    move.  An illegal move is asked again, -100 leaves the game where it is, -1 ends the game."""
    L = _oracle.oracle()
    n = 2
    GC, w = _ctx(n, 8, mode="play", human_plays_for=0)
    GC.getClient().setRequest(1, -1, 1)
    script = {"phase": 0, "asked": 0, "human_moves": []}

    def human(batch, bs):
        s = batch["s"]
        assert s.shape == (bs, 36, 8, 8) and (s[:, 34].reshape(bs, -1).max(1) == 1).all()   # black to move in every row
        script["asked"] += 1
        a = np.zeros(bs, dtype=np.int64)
        if script["phase"] == 1:
            a[:] = -1                                  # end the game
        elif script["asked"] == 1:
            a[:] = 169                                 # not legal at the start: asked again
        elif script["asked"] == 2:
            a[:] = -100                                # nothing yet
        elif len(script["human_moves"]) < 3:
            mv = int(np.flatnonzero(GC.getGame(0).getValidMoves())[0])
            a[:] = mv                                  # both games are identical: same legal moves
            script["human_moves"].append(mv)
        else:
            a[:] = -100
            script["idle"] = True
        return dict(pi=np.zeros((bs, 170), dtype=np.float32), V=np.zeros(bs, dtype=np.float32), a=a, rv=np.zeros(bs, dtype=np.int64))

    w.reg_callback("actor_white", human)
    w.reg_callback("actor_black", _uniform)
    ends = []
    w.reg_callback("game_end", lambda batch, bs: ends.append(1))
    w.start()
    names = []
    for _ in range(4000):
        names.append(w.run()[0])
        if script.get("idle"):
            break
    assert names[0] == "actor_white" and names[1] == "actor_white" and "actor_black" in names
    assert GC.getGame(0).getStateWords() == GC.getGame(1).getStateWords()
    # replay on the oracle: human moves for black, the AI (one tree, actor_black) for white
    g = AmGame(); ai = OracleMcts(policy_kind=0, **BASE)
    hm = list(script["human_moves"])
    while hm or (int(g.board()[4]) & 1) == 1:
        if (int(g.board()[4]) & 1) == 0:
            mv = hm.pop(0)
        else:
            mv = ai.run(g)["best"]
        assert g.forward(mv)
        ai.advance(mv)
    assert list(np.array(g.board(), dtype=np.uint64)) == GC.getGame(0).getStateWords()
    script["phase"] = 1                                     # now resign: -1 ends both games without a move
    for _ in range(4000):
        w.run()
        if len(ends) >= n:
            break
    assert len(ends) >= n
    assert (GC.getGame(0).getStateWords()[4] >> 8) & 0xFFFF == 1   # restarted
    w.stop()


def test_go_game_context_through_the_same_facade():
    """_elfgames_go: the GameContext / Context.wait-step surface over the build-defined Go rules (BASELINE.json
    configs[0] is Go self-play plumbing).  Two games, uniform actor, 16 rollouts per move: after ten moves the
    boards equal the oracle's (Go rules + search restated in oracle/go.c / oracle/mcts.cpp; parity unpinned vs
    ELF, which has no Go source in this fork)."""
    import _elfgames_go as gomod
    from tests._oracle import GoBoards
    co = gomod.ContextOptions()
    co.num_games, co.batchsize = 2, 8
    m = co.mcts_options
    m.num_threads, m.num_rollouts_per_thread, m.num_rollouts_per_batch, m.virtual_loss, m.alg_opt.c_puct = 1, 16, 4, 1, 1.5
    go = gomod.GameOptions()
    go.mode, go.policy_distri_cutoff = "selfplay", 0
    GC = gomod.GameContext(co, go)
    assert GC.getParams() == dict(num_action=362, board_size=19, num_future_actions=1, num_planes=18)
    desc = dict(game_end=dict(batchsize=1), actor_black=dict(input=["s"], reply=["pi", "V", "a", "rv"], batchsize=8))
    w = _gcwrapper.Wrapper(GC, 8, desc, num_recv=2)

    def actor(batch, bs):
        s = batch["s"]
        assert s.shape == (bs, 18, 19, 19) and np.isin(s, (0.0, 1.0)).all()
        assert (s[:, 16].reshape(bs, -1).max(1) + s[:, 17].reshape(bs, -1).max(1) == 1).all()   # exactly one side-to-move plane
        return dict(pi=np.full((bs, 362), np.float32(1.0 / 362.0), dtype=np.float32), V=np.zeros(bs, dtype=np.float32),
                    a=np.zeros(bs, dtype=np.int64), rv=np.zeros(bs, dtype=np.int64))
    w.reg_callback("actor_black", actor)
    w.start()
    nmoves = 10
    for _ in range(100000):
        w.run()
        if (GC.getGame(0).getStateWords()[20] >> 16) & 0xFFFF >= nmoves:
            break
    ply = (GC.getGame(0).getStateWords()[20] >> 16) & 0xFFFF
    g = GoBoards(1)
    om = OracleMcts(policy_kind=0, game="go", num_rollouts=16, rollouts_per_batch=4, virtual_loss=1, c_puct=1.5)
    for _ in range(ply):
        r = om.run(g.ptr(0))
        assert g.forward(r["best"], 0)
        om.advance(r["best"])
    assert [int(x) for x in g.export(0)] == GC.getGame(0).getStateWords() == GC.getGame(1).getStateWords()
    w.stop()


def test_reply_of_another_model_version_is_refused():
    """MCTSGameActor::post_nn_result (mcts/MCTSGameActor.h:222-229): an AI built for model version v throws when a
    reply carries another version; an async request (required_version = -1) accepts any."""
    GC, w = _ctx(2, 8)
    GC.getClient().setRequest(3, -1, 1)
    w.reg_callback("actor_black", _uniform)
    w.start()
    for _ in range(6):
        w.run()                                  # replies stamped with version 3: accepted
    w.stamp_rv = False                           # now the actor answers with rv = 0
    with pytest.raises(RuntimeError, match="model version 0 and required version 3 are not consistent"):
        for _ in range(50):
            w.run()
    w.stop()
