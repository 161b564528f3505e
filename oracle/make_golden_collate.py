"""Run HERE (the reference is importable only in this container): feeds the reference's own collater (lib/dataset/collater.py,
max_patch = 0 as tools/test.py:139 builds it) with small seeded per-image person lists and stores inputs + outputs as a fixture
for tests/test_input_oracle.py.  Usage: python oracle/make_golden_collate.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import importlib.util  # noqa: E402

# (the dataset package's __init__ pulls in cv2 / json_tricks; the collater module itself needs torch and numpy only)
_spec = importlib.util.spec_from_file_location("ref_collater", "/root/reference/lib/dataset/collater.py")
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
collater = _mod.collater

rng = np.random.RandomState(5)
persons = [2, 1, 3]
batch, flat_in, flat_mask = [], [], []
for i, n in enumerate(persons):
    inp = [torch.from_numpy(rng.randn(3, 6, 4).astype(np.float32)) for _ in range(n)]
    msk = [torch.from_numpy(rng.rand(1, 6, 4).astype(np.float32)) for _ in range(n)]
    tgt = [torch.zeros(2, 3, 2) for _ in range(n)]
    tw = [torch.ones(2, 1) for _ in range(n)]
    meta = dict(image="img%d" % i, filename="", rotation=0, imgnum=[0] * n, joints=[np.zeros((2, 3))] * n, joints_vis=[np.zeros((2, 3))] * n,
                center=[np.zeros(2)] * n, scale=[np.ones(2)] * n, score=[1] * n, box=[[0, 0, 1, 1]] * n)
    batch.append((inp, msk, tgt, tw, meta))
    flat_in += inp
    flat_mask += msk
x, m, _, _, meta = collater(0, "window")(batch)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "collate.npz"), persons=np.asarray(persons),
                    inputs=torch.stack(flat_in).numpy(), masks=torch.stack(flat_mask).numpy(), out_x=x.numpy(), out_m=m.numpy(),
                    out_length=meta["length"].numpy())
print("collate fixture:", x.shape, m.shape, meta["length"].tolist())
