"""Known answers for one findMove step of the reference's tree search, derived by hand-written numpy
float32 / float64 arithmetic from the reference's source text ALONE:
  EdgeInfo::getScore      elf/ai/tree_search/tree_search_base.h:175-201
  NodeT::findMove / UCT   elf/ai/tree_search/tree_search_node.h:218-244, 385-421
  BestAction::addAction   tree_search_node.h:350-364
It calls nothing under oracle/ and nothing of the product: it is the independent pin for the float semantics
(float divide by int, the one double sqrt and double product rounded once to float, the unfused `p * c_puct + q`,
the int truncation of num_visits + virtual_loss, the first strict maximum, the float sum of unsigned_q in edge
order).  Every case stores its inputs and the expected chosen edge, per-edge scores and new unsignedMeanQ_ as raw
float bits.  Cases tagged "fma" / "sqrtf" / "order" are ones where a fused multiply-add, a float sqrt, or another
summation order would give a different answer, so an implementation with that defect fails them.
Run: python tests/golden/gen_uct.py  ->  tests/golden/uct_known_answers.json"""
import json
import os

import numpy as np

f32 = np.float32
LOWEST = f32(np.finfo(np.float32).min)


def bits(x):
    return int(np.asarray(x, dtype=np.float32).view(np.uint32))


def scores(node, edges, opt, mode="exact"):
    """-> (per-edge score list, best index, new meanQ).  mode: exact | fma | sqrtf | reverse (the defects)."""
    flip = bool(node["flip"])
    mean_q = f32(node["meanQ"])
    if opt["unexplored_q_zero"] or (opt["root_unexplored_q_zero"] and node["depth"] == 0):
        mean_q = f32(0.0)                                            # findMove :233-236
    all_visits = int(node["numVisits"]) + 1                          # UCT :399
    best, best_i = LOWEST, -1
    total_uq, total_visits = f32(0.0), 0
    out = []
    order = range(len(edges)) if mode != "reverse" else range(len(edges) - 1, -1, -1)
    sc = {}
    for i in order:
        prior, reward, nvis, vloss = f32(edges[i][0]), f32(edges[i][1]), int(edges[i][2]), f32(edges[i][3])
        r = reward
        if flip:
            r = f32(-r)
        r = f32(r - vloss)
        nvl = int(f32(f32(nvis) + vloss))                            # int + float -> float, then truncation to int
        if nvl > 0:
            q = f32(r / f32(nvl))
        else:
            q = f32(-mean_q) if flip else mean_q
        uq = f32(reward / f32(nvis)) if nvis > 0 else mean_q
        p1 = f32(prior / f32(1 + nvis))                              # float / int -> float
        if mode == "sqrtf":
            pp = f32(p1 * f32(np.sqrt(f32(all_visits))))
        else:
            pp = f32(np.float64(p1) * np.sqrt(np.float64(all_visits)))   # std::sqrt(int) is double; one rounding on the store
        if opt["use_prior"]:
            if mode == "fma":
                s = f32(np.float64(pp) * np.float64(f32(opt["c_puct"])) + np.float64(q))   # product exact in double
            else:
                s = f32(f32(pp * f32(opt["c_puct"])) + q)
        else:
            s = q
        sc[i] = s
        first_visit = nvl == 0
        if not first_visit:
            total_uq = f32(total_uq + uq)
            total_visits += 1
    for i in range(len(edges)):                                      # first strict maximum in edge order
        if sc[i] > best:
            best, best_i = sc[i], i
        out.append(sc[i])
    new_mq = f32(f32(f32(node["parentQ"]) + total_uq) / f32(total_visits + 1))
    return out, best_i, new_mq


def random_case(rng, n, grid):
    flip = int(rng.integers(2))
    nv_edges = rng.integers(0, 40, size=n) * (rng.random(n) < 0.7)
    edges = []
    for k in range(n):
        nvis = int(nv_edges[k])
        if grid:
            prior = float(rng.integers(1, 64)) / 64.0
            reward = float(rng.integers(-8 * nvis, 8 * nvis + 1)) / 8.0 if nvis else 0.0
        else:
            prior = float(f32(rng.random()))
            reward = float(f32((rng.random() * 2 - 1) * nvis))
        vloss = float(rng.integers(0, 4)) if rng.random() < 0.3 else 0.0
        edges.append([float(f32(prior)), float(f32(reward)), nvis, vloss])
    node = dict(flip=flip, meanQ=float(f32(rng.random() * 2 - 1)), parentQ=float(f32(rng.random() * 2 - 1)),
                numVisits=int(sum(e[2] for e in edges)) + int(rng.integers(0, 3)), depth=int(rng.integers(2)))
    opt = dict(use_prior=int(rng.random() < 0.9), c_puct=float(f32(rng.choice([0.9, 1.1, 1.5, 2.3, 5.0]))),
               unexplored_q_zero=int(rng.random() < 0.15), root_unexplored_q_zero=int(rng.random() < 0.15))
    return node, edges, opt


def from_bits(u):
    return float(np.asarray(u, dtype=np.uint32).view(np.float32))


def directed_case(rng, mode):
    """A node on which the defect `mode` picks another edge than the reference arithmetic does: edge k's reward is
    walked, one float at a time, across the point where its score meets the leader's."""
    for _ in range(200):
        n = int(rng.integers(2, 7))
        node, edges, opt = random_case(rng, n, grid=False)
        opt["use_prior"] = 1
        k = int(rng.integers(1, n))
        if edges[k][2] == 0:
            edges[k][2] = int(rng.integers(1, 30))
        node["numVisits"] = int(sum(e[2] for e in edges))
        nv = edges[k][2]

        def lead(rb):
            edges[k][1] = from_bits(rb)
            sc, best, _ = scores(node, edges, opt)
            return best == k
        lo, hi = bits(f32(-nv)), bits(f32(nv))
        # reward bit patterns are not monotone across the sign: search the non-negative side or the negative side
        side = [(bits(f32(0.0)), bits(f32(nv)))] if not node["flip"] else [(bits(f32(-0.0)), bits(f32(-nv)))]
        for a, b in side:           # along this side the edge's q grows with the bit pattern
            if lead(a) or not lead(b):
                continue
            while b - a > 1:
                m = (a + b) // 2
                if lead(m):
                    b = m
                else:
                    a = m
            for rb in range(max(a - 300, 1), b + 300):
                edges[k][1] = from_bits(rb)
                _, b1, _ = scores(node, edges, opt)
                _, b2, _ = scores(node, edges, opt, mode)
                if b1 != b2:
                    return node, [list(e) for e in edges], opt
    return None


def main():
    rng = np.random.default_rng(20261111)
    cases, want = [], dict(plain=40, big=12, fma=12, sqrtf=12, order=12)
    have = dict.fromkeys(want, 0)
    tries = 0
    directed = []
    for mode in ("fma", "sqrtf"):
        while sum(1 for m, _ in directed if m == mode) < want[mode]:
            c = directed_case(rng, mode)
            if c is None:
                break
            directed.append((mode, c))
    while (directed or any(have[k] < want[k] for k in ("plain", "big", "order"))) and tries < 100000:
        tries += 1
        big = have["big"] < want["big"] and tries % 7 == 0
        n = int(rng.integers(17, 330)) if big else int(rng.integers(1, 17))
        if directed:
            big = False
            _, (node, edges, opt) = directed.pop()
        else:
            node, edges, opt = random_case(rng, n, grid=bool(rng.integers(2)))
        sc, best, mq = scores(node, edges, opt)
        tags = []
        for mode in ("fma", "sqrtf"):
            sc2, best2, _ = scores(node, edges, opt, mode)
            if best2 != best:
                tags.append(mode)
        _, _, mq3 = scores(node, edges, opt, "reverse")
        if bits(mq3) != bits(mq) and have["order"] < want["order"]:
            tags.append("order")
        if not tags:
            tags = ["big"] if big and have["big"] < want["big"] else (["plain"] if have["plain"] < want["plain"] and not big else [])
        if not tags:
            continue
        for t in tags:
            have[t] += 1
        cases.append(dict(tags=tags, node=dict(flip=node["flip"], meanQ=bits(node["meanQ"]), parentQ=bits(node["parentQ"]),
                                               numVisits=node["numVisits"], depth=node["depth"]),
                          opt=dict(use_prior=opt["use_prior"], c_puct=bits(opt["c_puct"]), unexplored_q_zero=opt["unexplored_q_zero"],
                                   root_unexplored_q_zero=opt["root_unexplored_q_zero"]),
                          edges=[[bits(e[0]), bits(e[1]), e[2], bits(e[3])] for e in edges],
                          expect=dict(best=best, meanQ=bits(mq), scores=[bits(s) for s in sc])))
    out = dict(source="numpy float32/float64 restatement of tree_search_base.h:175-201, tree_search_node.h:218-244,350-364,385-421; "
                      "floats are stored as uint32 bit patterns; edges = [prior, reward, num_visits, virtual_loss]",
               numpy=np.__version__, counts=have, cases=cases)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "uct_known_answers.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(have, "tries", tries, "->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
