"""CPU: the oracle (oracle/i2r_cpu.py) against the golden vectors produced by the imported reference."""
import numpy as np
import pytest
import torch

import i2r_cpu
from _golden import CASES, VARIANTS, probe_of, setup


def _flatten(collect):
    out = {}
    for k, v in collect.items():
        if isinstance(v, (list, tuple)):
            for i, t in enumerate(v):
                out["%s.%d" % (k, i)] = t
        else:
            out[k] = v
    return out


@pytest.mark.parametrize("tag", sorted(CASES) + sorted(VARIANTS))
def test_oracle_matches_reference_golden(tag):
    torch.set_num_threads(8)
    cfg, sd, x, m, length, g = setup(tag)
    collect = {}
    z = i2r_cpu.forward(sd, cfg, x, m, length, collect)
    outs = z if isinstance(z, dict) else {"multi": z}
    for k, t in outs.items():
        scale = max(1.0, float(np.abs(g["probe_out_" + k]).max()))
        np.testing.assert_allclose(probe_of(t, tag + k), g["probe_out_" + k], rtol=0, atol=2e-5 * scale)
        if "out_" + k in g:
            assert t.shape == g["out_" + k].shape
            assert np.abs(t.numpy() - g["out_" + k]).max() < 2e-5 * max(1.0, float(np.abs(g["out_" + k]).max()))
    # per-stage probes localise any drift
    n = 0
    for k, t in _flatten(collect).items():
        ref = g["stage_" + k]
        np.testing.assert_allclose(probe_of(t, tag + k), ref, rtol=0, atol=2e-5 * max(1.0, float(np.abs(ref).max())))
        n += 1
    assert n >= 10


def test_padding_is_output_equivalent():
    """Reference pads persons to max(length) + key mask; the var-len formulation must give the same crops:
    running an image alone equals running it inside a ragged batch (persons interact only within an image)."""
    cfg, sd, x, m, length, g = setup("w48_l213")
    full = i2r_cpu.forward(sd, cfg, x, m, length)
    o = 0
    for n in length:
        alone = i2r_cpu.forward(sd, cfg, x[o:o + n], m[o:o + n], [n])
        assert (alone - full[o:o + n]).abs().max().item() < 2e-5
        o += n
