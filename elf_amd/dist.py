"""The only two collectives of the self-play path, over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests):

* broadcast_weights  - trainer rank -> all self-play ranks: one contiguous blob + version id.
  Replaces the reference's save-<ver>.bin on a shared filesystem reloaded by every client
  (scripts/elfgames/american_checkers/py/selfplay.py:81-92,214-236); the version travels with the
  blob because the actor checks reply.version == required_version (MCTSGameActor.h:222-229).
* gather_records     - finished-game slabs of every rank -> trainer rank: an all_gather of counts,
  then padded all_gathers of the fixed-stride arrays.  Replaces JSON over ZMQ DEALER->ROUTER
  (common/DistriClient.h:104-122).
Games themselves never communicate: boards and trees are sharded by game index."""
import torch
import torch.distributed as dist


def shard_games(total_games, rank, world):
    """Contiguous shard [lo, hi) of game ids owned by `rank` (game g -> rank by block)."""
    per, rem = divmod(total_games, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def _host_staged(t, group):
    """gloo moves host memory: a CUDA tensor goes through a host copy there (the one-GPU rehearsals); with RCCL
    ("nccl") the tensor itself is the collective's buffer and nothing leaves HBM."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def broadcast_weights(flat, version, src=0, group=None):
    """flat: 1-D tensor (any dtype) with the same numel on every rank.  Returns (flat, version)."""
    staged = _host_staged(flat, group)
    hdr = torch.tensor([int(version), flat.numel()], dtype=torch.int64, device="cpu" if staged else flat.device)
    dist.broadcast(hdr, src=src, group=group)
    if int(hdr[1].item()) != flat.numel():
        raise RuntimeError("broadcast_weights: blob size mismatch (%d vs %d)" % (int(hdr[1].item()), flat.numel()))
    if staged:
        h = flat.cpu()
        dist.broadcast(h, src=src, group=group)
        flat.copy_(h)
    else:
        dist.broadcast(flat, src=src, group=group)
    return flat, int(hdr[0].item())


def update_model(net, flat, version, src=0, group=None):
    """The self-play side of a model update (replaces selfplay.py:81-92,214-236's reload from the shared filesystem):
    every rank passes its fp32 CUDA buffer of net.param_count() floats - on `src` holding the trainer's new weights
    in pack_state_dict order - and ends up with those weights loaded (elfhip_net_load_device) under `version`."""
    flat, version = broadcast_weights(flat, version, src=src, group=group)
    net.load_device(flat, version)
    return version


def gather_records(slab, dst=0, group=None):
    """slab: dict name -> tensor whose leading dimension is this rank's number of finished games
    (may be 0; trailing shapes identical on all ranks).  Returns on `dst` a dict of concatenated
    tensors (rank order), elsewhere None."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    names = sorted(slab.keys())
    k = slab[names[0]].shape[0] if names else 0
    dev = slab[names[0]].device if names else torch.device("cpu")
    if names and _host_staged(slab[names[0]], group):
        slab = {n: v.cpu() for n, v in slab.items()}
        dev = torch.device("cpu")
    cnt = torch.tensor([k], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    kmax = max(counts) if counts else 0
    out = {}
    for name in names:
        t = slab[name]
        pad = torch.zeros((kmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:k] = t
        # ship as raw bytes: one code path for every dtype (gloo has no int16 collectives)
        raw = pad.contiguous().view(torch.uint8).reshape(-1)
        parts = [torch.empty_like(raw) for _ in range(world)]
        dist.all_gather(parts, raw, group=group)
        if rank == dst:
            typed = [p.view(t.dtype).reshape(pad.shape) for p in parts]
            out[name] = torch.cat([p[:c] for p, c in zip(typed, counts)], dim=0)
    return out if rank == dst else None
