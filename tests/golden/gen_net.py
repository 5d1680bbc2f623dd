"""Generate tests/golden/net_*.npz: (input, pi, V) of the fp32 PyTorch-CPU restatement of
Model_PolicyValue (oracle/model_ref.py) with seeded random weights (the seed is stored in the file).  The weights themselves are NOT
stored (they are regenerated from the seed in the test); inputs are real feature planes of seeded
random checkers games (oracle) or random {0,1} planes (Go-shaped).  Run: python tests/golden/gen_net.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.model_ref import seeded_model  # noqa: E402
from tests import _oracle  # noqa: E402

CONFIGS = {
    "checkers_4x64": dict(planes=36, board=8, dim=64, num_block=4, num_action=170),
    "checkers_10x128": dict(planes=36, board=8, dim=128, num_block=10, num_action=170),
    "go_2x32": dict(planes=18, board=19, dim=32, num_block=2, num_action=362),
    "go_4x224": dict(planes=18, board=19, dim=224, num_block=4, num_action=362),
    "go_20x224": dict(planes=18, board=19, dim=224, num_block=20, num_action=362),   # BASELINE.json configs[3]
    "checkers_4x64_leaky": dict(planes=36, board=8, dim=64, num_block=4, num_action=170, leaky_relu=True),   # --leaky_relu
    "go_2x32_leaky": dict(planes=18, board=19, dim=32, num_block=2, num_action=362, leaky_relu=True),
}


def checkers_inputs(n, seed):
    rng = np.random.default_rng(seed)
    L = _oracle.oracle()
    out = []
    g = _oracle.AmGame()
    while len(out) < n:
        if g.terminated():
            g.reset()
        m = np.zeros(3, dtype=np.uint64); L.am_legal_mask(g.board(), m)
        legal = _oracle.mask_bits(m, 170)
        g.forward(legal[rng.integers(len(legal))])
        if rng.random() < 0.3:
            out.append(g.features().copy())
    return np.stack(out).astype(np.float32)


def main():
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for name, cfg in CONFIGS.items():
        if only and name not in only:
            continue
        n = 24 if cfg["board"] == 8 else (4 if cfg["num_block"] >= 20 else 6)
        if cfg["board"] == 8:
            x = checkers_inputs(n, 5)
        else:
            x = (np.random.default_rng(7).random((n, cfg["planes"], 19, 19)) < 0.3).astype(np.float32)
        # seeded random weights leave a head's single channel dead (ReLU output 0 everywhere) for many seeds, and a
        # dead head's output does not depend on the trunk at all: take the first seed whose two heads are both alive
        for seed in range(64):
            m = seeded_model(seed=seed, **cfg)
            with torch.no_grad():
                xt = torch.from_numpy(x)
                t = m.resnet(m.init_conv(xt))
                alive = 1.0 if cfg.get("leaky_relu") else min(float((m.pi_final_conv(t) > 0).float().mean()), float((m.value_final_conv(t) > 0).float().mean()))
                o = m(xt)
            if alive >= 0.3:
                break
        else:
            raise SystemExit("no seed with live heads for " + name)
        np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "net_%s.npz" % name),
                            x=np.packbits(x.astype(np.uint8), axis=None), x_shape=np.array(x.shape),
                            pi=o["pi"].numpy(), V=o["V"].numpy(), seed=np.array(seed), cfg=np.array([cfg[k] for k in ("planes", "board", "dim", "num_block", "num_action")]), leaky=np.array(int(bool(cfg.get("leaky_relu")))),
                            meta=np.array(["torch %s CPU fp32; oracle/model_ref.py seeded_model(seed=%d)" % (torch.__version__, seed)]))
        print(name, "seed", seed, x.shape, float(o["pi"].max()), float(o["V"].abs().max()))


if __name__ == "__main__":
    main()
