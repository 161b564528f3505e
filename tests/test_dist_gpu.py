"""-m gpu: the path's one collective on hardware.  A 1-GPU box still runs RCCL: backend 'nccl' with world_size 1 exercises communicator
creation, all_gather_into_tensor on device buffers produced by a real forward, the async handle and the key-point payload (the driver's
8-GPU run is the only place a multi-rank RCCL ring exists; tests/test_dist.py covers world_size 2 on gloo)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from _golden import setup
from i2r_amd import caller, models
from i2r_amd import dist as i2r_dist

pytestmark = pytest.mark.gpu


@pytest.fixture()
def nccl_world1():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        yield
    finally:
        dist.destroy_process_group()


def test_rccl_gather_after_real_forward(nccl_world1):
    cfg, sd, x, m, length, g = setup("w48_l213")
    net = models.interformer_pureMulti.get_pose_net(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    lo, hi, off = i2r_dist.shard_images(length, 0, 1)
    assert (lo, hi, off) == (0, len(length), 0)
    y = net(x.cuda(), m.cuda(), length)
    counts = [sum(length)]
    full = i2r_dist.gather_heatmaps(y, counts)
    assert full.is_cuda and torch.equal(full, y)
    h1 = i2r_dist.gather_heatmaps_async(y, counts)          # collective of step k in flight while step k+1 is issued (bench.py pattern)
    y2 = net(x.cuda(), m.cuda(), length)
    h2 = i2r_dist.gather_heatmaps_async(y2, counts)
    assert torch.equal(h1.wait(), y) and torch.equal(h2.wait(), y2) and torch.equal(y, y2)
    # the default payload: key points decoded on the device, [S, J, 3]
    preds, maxv = caller.decode(y, None, None, cfg.TEST.BLUR_KERNEL, transform_back=False)
    kp = i2r_dist.gather_keypoints(preds, maxv, counts)
    assert kp.shape == (sum(length), cfg.MODEL.NUM_JOINTS, 3)
    assert torch.equal(kp[..., :2], preds) and torch.equal(kp[..., 2:], maxv)
    kp2 = i2r_dist.gather_keypoints(preds, maxv, counts, async_op=True).wait()
    assert torch.equal(kp2, kp)
    t = torch.ones(1, device="cuda")
    dist.all_reduce(t)                                       # bench.py's max-over-ranks timing reduction
    assert t.item() == 1.0
