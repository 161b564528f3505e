"""CPU, world_size 2 over gloo: the image sharding + heatmap all-gather used by bench.py --gpus N."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from i2r_amd import dist as i2r_dist


def test_shard_images_partitions_everything():
    for length in ([4] * 8, [1, 6, 2, 3, 1, 1, 5], [3], [2, 2]):
        for world in (1, 2, 4, 8):
            if world > len(length):
                continue
            seen, crops = [], 0
            for r in range(world):
                lo, hi, off = i2r_dist.shard_images(length, r, world)
                assert hi > lo and off == sum(length[:lo])
                seen += list(range(lo, hi))
                crops += sum(length[lo:hi])
            assert seen == list(range(len(length))) and crops == sum(length)


def test_shard_images_never_starves_a_rank():
    """property: with at least as many images as ranks every rank gets >= 1 image, cuts are monotone, and no rank
    carries more than an equal share plus one image's crops"""
    import random
    rnd = random.Random(0)
    for length, world in (([1, 1, 10, 1, 1], 3), ([8, 1, 1, 1], 4)):
        b = i2r_dist.shard_bounds(length, world)
        assert all(b[i] < b[i + 1] for i in range(world)), (length, world, b)
    for _ in range(5000):
        n, world = rnd.randint(1, 40), rnd.randint(1, 8)
        length = [rnd.choice([1, 1, 1, 2, 3, 6, 10]) for _ in range(n)]
        b = i2r_dist.shard_bounds(length, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(world))
        if n >= world:
            assert all(b[i] < b[i + 1] for i in range(world)), (length, world, b)
            assert max(sum(length[b[i]:b[i + 1]]) for i in range(world)) <= sum(length) / world + max(length)
        if world == 2 and n >= 2:  # two part-batches (Engine._split_bounds): the cut is the image boundary nearest to half of the crops
            half = sum(length) / 2.0
            best = min(abs(sum(length[:j]) - half) for j in range(1, n))
            assert abs(sum(length[:b[1]]) - half) == best, (length, b)
    assert i2r_dist.shard_bounds([6, 4, 4, 2, 2, 1, 1, 1, 2, 5, 4, 6, 4, 4, 6, 5], 2) == [0, 10, 16]  # 28 / 29 crops (bench config 3)


def _worker(rank, world, port, length):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(sum(length) * 2 * 3 * 2, dtype=torch.float32).view(sum(length), 2, 3, 2)
        counts = []
        for r in range(world):
            lo, hi, off = i2r_dist.shard_images(length, r, world)
            counts.append(sum(length[lo:hi]))
        lo, hi, off = i2r_dist.shard_images(length, rank, world)
        local = full[off:off + counts[rank]].clone()
        got = i2r_dist.gather_heatmaps(local, counts)
        assert torch.equal(got, full), (rank, got.shape)
        h1 = i2r_dist.gather_heatmaps_async(local, counts)       # two collectives in flight, waited in order (bench.py pattern)
        h2 = i2r_dist.gather_heatmaps_async(local * 2, counts)
        assert torch.equal(h1.wait(), full) and torch.equal(h2.wait(), full * 2)
        # decoded key points: [S_r, J, 2] + [S_r, J, 1] -> [S, J, 3] in global crop order
        S, J = sum(length), 5
        preds = torch.arange(S * J * 2, dtype=torch.float32).view(S, J, 2)
        maxv = -torch.arange(S * J, dtype=torch.float32).view(S, J, 1)
        kp = i2r_dist.gather_keypoints(preds[off:off + counts[rank]], maxv[off:off + counts[rank]], counts)
        assert torch.equal(kp, torch.cat([preds, maxv], 2))
        kp2 = i2r_dist.gather_keypoints(preds[off:off + counts[rank]], maxv[off:off + counts[rank]], counts, async_op=True).wait()
        assert torch.equal(kp2, kp)
    finally:
        dist.destroy_process_group()


def test_gather_heatmaps_world2_gloo():
    for length in ([4, 4], [1, 3, 2], [5]):   # [5]: fewer images than ranks -- rank 1 contributes zero rows
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(_worker, args=(2, port, length), nprocs=2, join=True)
