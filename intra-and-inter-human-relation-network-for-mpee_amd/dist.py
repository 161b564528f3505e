"""Data-parallel sharding of IMAGES across ranks (one process per GPU) + the one collective of the path.

Persons interact only within an image (reference interformer.py:294-306: the encoder's batch dim is the image),
so images are the independent unit: rank r runs the full forward on its images with replicated weights and the
per-crop results are all-gathered once per step (RCCL over xGMI when backend == 'nccl'; gloo in the CPU tests): the decoded
key points [S, J, 3] (gather_keypoints: what validate() keeps of a batch, function.py:190-200 -- 168 B per crop at J = 14) or,
optionally, the heat maps themselves (gather_heatmaps: 172 KB per crop).
The reference itself has no counterpart (its DataParallel/DDP eval does not shard).
"""
import torch
import torch.distributed as dist


def shard_bounds(length, world):
    """Image index cuts [b_0 = 0, b_1, ..., b_world = n] of contiguous chunks balanced by crop count (cost is proportional to
    crops).  Every rank gets at least one image whenever there are at least as many images as ranks (a rank with an empty shard
    would skip the forward and hang the collective); with fewer images than ranks the trailing ranks are empty."""
    n = len(length)
    prefix = [0]
    for l in length:
        prefix.append(prefix[-1] + l)
    total = prefix[-1]
    bounds = [0]
    for k in range(1, world):
        lo = min(bounds[-1] + 1, n)            # a chunk holds at least one image ...
        hi = max(lo, n - (world - k))          # ... and leaves one for every rank behind it (when there are enough images)
        hi = min(hi, n)
        # the image boundary whose crop prefix is nearest to k equal shares (the earlier one on a tie)
        bounds.append(min(range(lo, hi + 1), key=lambda j: (abs(prefix[j] * world - total * k), j)))
    bounds.append(n)
    return bounds


def shard_images(length, rank, world):
    """-> (image index range [lo, hi) of this rank, crop offset of image lo)."""
    bounds = shard_bounds(length, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    return lo, hi, sum(length[:lo])


class PendingGather:
    """Handle of an in-flight heat-map all-gather (see gather_heatmaps_async)."""

    def __init__(self, work, out, counts, smax, shape):
        self.work, self.out, self.counts, self.smax, self.shape = work, out, counts, smax, shape

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if all(c == self.smax for c in self.counts):
            return self.out
        out = self.out.view(len(self.counts), self.smax, *self.shape)
        return torch.cat([out[r, :c] for r, c in enumerate(self.counts)], dim=0)


def gather_heatmaps_async(local, counts, group=None):
    """Start the per-step all-gather without blocking the launch stream: the collective runs on RCCL's own stream over
    xGMI while the NEXT forward's kernels are being issued/executed; call .wait() on the returned handle when the gathered
    heat-maps are needed (bench.py waits one step later, i.e. communication of step k overlaps compute of step k+1)."""
    world = dist.get_world_size(group)
    assert len(counts) == world and local.shape[0] == counts[dist.get_rank(group)]
    smax = max(counts)
    pad = local
    if local.shape[0] < smax:
        pad = torch.zeros((smax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
    out = torch.empty((world * smax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, pad.contiguous(), group=group, async_op=True)
    return PendingGather(work, out, list(counts), smax, tuple(local.shape[1:]))


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


def gather_keypoints(preds, maxvals, counts, group=None, async_op=False):
    """The per-step collective when the decode runs on the device (caller.decode): preds [S_r, J, 2] + maxvals [S_r, J, 1] ->
    [sum(counts), J, 3] (x, y, score) in global crop order -- what validate() stores per batch (all_preds[idx:idx+n, :, 0:2] = preds,
    [..., 2:3] = maxvals, lib/core/function.py:193-195).  ~1000x smaller than the heat maps.  async_op: returns a PendingGather."""
    kp = torch.cat([preds, maxvals], dim=2).contiguous()
    if async_op:
        return gather_heatmaps_async(kp, counts, group)
    return gather_heatmaps(kp, counts, group)


class PostStep:
    """What follows the forward inside a data-parallel step -- device decode + the all-gather -- issued on a side stream so that it runs
    under the NEXT forward instead of between two forwards: the caller's stream only records one event; the side stream waits for it,
    decodes (`decode(y) -> (preds, maxvals)`, or None for the heat-map payload), starts the asynchronous all-gather and, one step later,
    waits for it.  `result()` joins the caller's stream with the last gather and returns its tensor (the host-visible end of a run).
    The forward's output stays alive for the side stream through record_stream."""

    _streams = {}  # device -> the side stream, shared by every PostStep of the process

    def __init__(self, device, counts, decode=None, group=None):
        import torch as _t
        from . import engine as _engine
        key = str(_t.device(device))
        if key not in PostStep._streams:
            # The LAST engine lane's stream, not a stream of its own: a process has four hardware queues (the caller's stream + three
            # lanes), a fifth stream shares one of them, and which one is the runtime's choice.  Measured in fresh processes with a
            # one-rank RCCL group (config 4, collective line / plain line, 8 pairs each on one box): a stream of its own 0.89-0.986
            # (median 0.97), lane 3's stream 0.980-1.004 (median 0.99).  Lane 3 only works in the last stage of a four-lane forward (and
            # not at all in the part-batch forwards): it is idle while the next forward starts, which is when the decode + gather run.
            # Default priority: a HIGH-priority side stream was measured at 0.55 -- an active stream of another priority class wrecks the
            # overlap of the lanes (as stream priorities for the lanes themselves did in round 3); so did GPU_MAX_HW_QUEUES=8 (0.66).
            PostStep._streams[key] = _engine.lane_streams(_t.device(device), 3)[2]
        self.stream = PostStep._streams[key]
        self.counts, self.decode, self.group = list(counts), decode, group
        self.pending = None
        self.ready = _t.cuda.Event()

    def __call__(self, y):
        cur = torch.cuda.current_stream(y.device)
        self.ready.record(cur)
        y.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(self.ready)
            if self.decode is not None:
                preds, maxv = self.decode(y)
                h = gather_keypoints(preds, maxv, self.counts, self.group, async_op=True)
            else:
                h = gather_heatmaps_async(y, self.counts, self.group)
            if self.pending is not None:
                self.pending.wait()     # (the gather of the step before: the SIDE stream waits, not the forward's)
            self.pending = h
        return h

    def result(self):
        if self.pending is None:
            return None
        with torch.cuda.stream(self.stream):
            out = self.pending.wait()
        torch.cuda.current_stream(out.device).wait_stream(self.stream)
        self.pending = None
        return out
