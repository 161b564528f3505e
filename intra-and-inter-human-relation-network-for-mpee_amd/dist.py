"""Data-parallel sharding of IMAGES across ranks (one process per GPU) + the one collective of the path.

Persons interact only within an image (reference interformer.py:294-306: the encoder's batch dim is the image),
so images are the independent unit: rank r runs the full forward on its images with replicated weights and the
per-crop heatmaps are all-gathered once per step (RCCL over xGMI when backend == 'nccl'; gloo in the CPU tests).
The reference itself has no counterpart (its DataParallel/DDP eval does not shard).
"""
import torch
import torch.distributed as dist


def shard_images(length, rank, world):
    """Contiguous image chunks balanced by crop count (cost is proportional to crops).
    -> (image index range [lo, hi), crop offset of image lo)."""
    n = len(length)
    total = sum(length)
    bounds, acc, nxt = [0], 0, 1
    for i, l in enumerate(length):
        acc += l
        # cut after image i when the running crop count passes the next equal share (keep >= 1 image per remaining rank)
        while nxt < world and acc >= total * nxt / world and (n - (i + 1)) >= (world - nxt) and len(bounds) == nxt:
            bounds.append(i + 1)
            nxt += 1
    while len(bounds) < world:
        bounds.append(min(n, bounds[-1] + 1))
    bounds.append(n)
    lo, hi = bounds[rank], bounds[rank + 1]
    return lo, hi, sum(length[:lo])


def gather_heatmaps(local, counts, group=None):
    """local: [S_r, J, h, w] on this rank; counts: crops per rank (list, same on all ranks) -> [sum(counts), J, h, w].

    One all_gather_into_tensor of buffers padded to max(counts) (<= a few MB per rank: latency-, not bandwidth-bound
    on xGMI), then the padding is stripped in rank order, which restores the original crop order for contiguous shards.
    """
    world = dist.get_world_size(group)
    assert len(counts) == world and local.shape[0] == counts[dist.get_rank(group)]
    smax = max(counts)
    pad = local
    if local.shape[0] < smax:
        pad = torch.zeros((smax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
    out = torch.empty((world * smax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    if all(c == smax for c in counts):
        return out
    out = out.view(world, smax, *local.shape[1:])
    return torch.cat([out[r, :c] for r, c in enumerate(counts)], dim=0)
