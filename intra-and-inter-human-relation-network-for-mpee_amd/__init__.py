"""i2r_amd -- MI355X-native I2R-Net inference hot path (package dir: intra-and-inter-human-relation-network-for-mpee_amd).

The directory name is not a Python identifier; import it through the repo-root shim ``i2r_amd`` (i2r_amd.py),
which registers this package in ``sys.modules`` under that name.

Layout (only what the hot path needs -- see DESIGN.md):
  config.py   yacs-free CfgNode + defaults of the keys the path reads
  synth.py    key-addressed deterministic weights / synthetic crops
  arch.py     declarative layer graph of the three I2R-Net variants (state-dict keys == reference's)
  cabi.py     ctypes binding of the C-ABI library (include/i2r_hip.h); fails loudly when missing
  engine.py   weight packing (BN fold, NHWC/k4 layouts) + op program executed through the C-ABI
  models/     host-side mirror of the reference's ``models`` package (get_pose_net factories)
  dist.py     one-process-per-GPU image sharding + RCCL all-gather of heatmaps
  csrc/       hand-written HIP kernels for gfx950 + the extern "C" boundary
"""
__version__ = "0.1.0"
