"""Experiment: Go-shape forward over a big batch in sub-batches (activation working set vs the 256 MB MALL)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from elf_amd.net import PolicyValueNet, random_state_dict
cfg = dict(planes=18, board=19, dim=224, num_block=20, num_action=362)
B = 8192
net = PolicyValueNet(max_batch=B, **cfg)
net.load_state_dict(random_state_dict(seed=0, **cfg))
x = (torch.rand((B, 18, 19, 19), device="cuda") < 0.3).float()
for chunk in (8192, 4096, 2048, 1024, 512, 256):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for c0 in range(0, B, chunk):
            net.forward(x[c0:c0 + chunk])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("chunk %5d: %.1f positions/s" % (chunk, B / dt), flush=True)
