"""Generate the golden vectors under tests/golden/ by running THE REFERENCE ITSELF (imported from
/root/reference through oracle/ref_shim.py) on CPU.  Run in the build container only:

    python oracle/make_golden.py

Per workload config it writes
  <tag>_keys.json   state-dict key -> [shape, dtype] manifest of the reference module (drop-in contract)
  <tag>.npz         expected outputs of the reference for the seeded synthetic inputs/weights
                    (inputs and weights are NOT stored: i2r_amd.synth regenerates them bit-identically),
                    input/weight checksums, and per-stage probes (mean, abs-mean, 32 sampled values)
Every case is also cross-checked against the CPU restatement (oracle/i2r_cpu.py) before it is written.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import i2r_amd  # noqa: E402,F401
from i2r_amd import config, synth  # noqa: E402
import i2r_cpu  # noqa: E402
import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

CASES = [
    # tag, config name, length, (H, W), store full output?
    ("w48_l31", "w48_pure_en6", [3, 1], (256, 192), True),
    ("w48_l44", "w48_pure_en6", [4, 4], (256, 192), True),       # BASELINE config 1: the CPU-runnable case the metric is quoted beside
    ("w48_l1", "w48_pure_en6", [1], (256, 192), True),
    ("w48_l213", "w48_pure_en6", [2, 1, 3], (256, 192), True),
    ("tph_l21", "tph_192_p6_b4", [2, 1], (256, 192), True),
    ("hrt_l21", "hrt_192_p4_b4", [2, 1], (256, 192), True),
    ("hrt288_l2", "coco_hrt_288_p2_b4", [2], (384, 288), True),      # 96x72 maps, 24x18 inter-human tokens
    ("tph2s_l12", "coco_tph_192_p4_b4", [1, 2], (256, 192), True),   # interformer_2stage wiring (multiplex deconv, multi-pos)
    ("ochtph_l21", "ochuman_tph_192_p3_b8", [2, 1], (256, 192), True),  # multi-position mode 'res' (resnet18 front end)
    ("bare_l21", "w48_bare_p6", [2, 1], (256, 192), True),           # interformer with MODEL.SINGLEFORMER unset (models/hrnet.py)
]


# Reference-expressible settings no shipped yaml uses (the config keys of the path): base config + KEY VALUE overrides.
# The same table lives in tests/_golden.py (VARIANTS); a variant whose overrides change the parameter inventory gets its own <tag>_keys.json.
VARIANTS = [
    # tag, config name, overrides, length, (H, W)
    ("w48_nh8_l21", "w48_pure_en6", ["MODEL.N_HEAD", 8], [2, 1], (256, 192)),                 # vanilla, 8 heads of 12 dims, conv position embedding
    ("tph_nh4_l11", "tph_192_p6_b4", ["MODEL.N_HEAD", 4], [1, 1], (256, 192)),                # both encoder stacks multi-head (sine table in stage 1)
    ("bare_pre_l21", "w48_bare_p6", ["MODEL.NORMALIZE_BEFORE", True], [2, 1], (256, 192)),    # forward_pre, one head of 96 dims
    ("hrt_pre_nh2_l21", "hrt_192_p4_b4", ["MODEL.NORMALIZE_BEFORE", True, "MODEL.N_HEAD", 2], [2, 1], (256, 192)),  # d = 78: heads of 39 dims
    ("w48_fk3_l21", "w48_pure_en6", ["MODEL.EXTRA.FINAL_CONV_KERNEL", 3], [2, 1], (256, 192)),            # 3x3 final_layer (padding 1)
    ("tph_up_l21", "tph_192_p6_b4", ["MODEL.UPSAMPLE_TYPE", "upconv"], [2, 1], (256, 192)),               # interformer.UpConv (upsample_layer.*)
    ("w48_cv_l21", "w48_pure_en6", ["MODEL.MULTI_POS_EMBEDDING", "cat_vec"], [2, 1], (256, 192)),           # per-person vector ADDED (interformer_pureMulti.py:770)
    ("bare_cv_l21", "w48_bare_p6", ["MODEL.MULTI_POS_EMBEDDING", "cat_vec"], [2, 1], (256, 192)),            # CONCATENATED: 192-wide encoder + fc (interformer.py:296-303)
    ("ochtph_cv_nh2_l21", "ochuman_tph_192_p3_b8", ["MODEL.MULTI_POS_EMBEDDING", "cat_vec", "MODEL.N_HEAD", 2], [2, 1], (256, 192)),  # the same behind a TransPose-H first stage
    ("tph2s_cv_l12", "coco_tph_192_p4_b4", ["MODEL.MULTI_POS_EMBEDDING", "cat_vec", "MODEL.USE_MULTI_POS", True], [1, 2], (256, 192)),  # interformer_2stage: added
    ("tph2s_dt_l12", "coco_tph_192_p4_b4", ["MODEL.DOMAIN_TRANS", True], [1, 2], (256, 192)),               # interformer_2stage.py:413-414
    ("w48_d64_l21", "w48_pure_en6", ["MODEL.DIM_MODEL", 64, "MODEL.DIM_FEEDFORWARD", 128, "MODEL.EXTRA.NUM_DECONV_FILTERS", [64]], [2, 1], (256, 192)),  # other widths
    ("tph_pnone_l11", "tph_192_p6_b4", ["MODEL.POS_EMBEDDING", "none"], [1, 1], (256, 192)),                  # TransPose-H without its position table
    ("w32_l21", "w48_pure_en6", ["MODEL.EXTRA.STAGE2.NUM_CHANNELS", [32, 64], "MODEL.EXTRA.STAGE3.NUM_CHANNELS", [32, 64, 128]], [2, 1], (256, 192)),  # HRNet-W32 widths
    ("w48_m2b2_l21", "w48_pure_en6", ["MODEL.EXTRA.STAGE3.NUM_MODULES", 2, "MODEL.EXTRA.STAGE2.NUM_BLOCKS", [2, 2], "MODEL.EXTRA.STAGE3.NUM_BLOCKS", [2, 3, 1]], [2, 1], (256, 192)),  # other module / block counts
    ("w48_dk3_l21", "w48_pure_en6", ["MODEL.EXTRA.NUM_DECONV_KERNELS", [3], "MODEL.EXTRA.DECONV_WITH_BIAS", True], [2, 1], (256, 192)),  # ConvTranspose2d(3, 2, 1, output_padding 1) with bias
    ("tph_dk2_l21", "tph_192_p6_b4", ["MODEL.EXTRA.NUM_DECONV_KERNELS", [2]], [2, 1], (256, 192)),           # ConvTranspose2d(2, 2, 0): one tap per parity
    ("bare_sine_l213", "w48_bare_p6", ["MODEL.MULTI_POS_EMBEDDING", "sine"], [2, 1, 3], (256, 192)),          # canvas table, 3 persons wide (position_embedding.py:34-61,88-91)
    ("ochtph_sine_l12", "ochuman_tph_192_p3_b8", ["MODEL.MULTI_POS_EMBEDDING", "sine"], [1, 2], (256, 192)),  # the same behind a TransPose-H first stage
    ("bare_win_l213", "w48_bare_p6", ["MODEL.ATTENTION_TYPE", "window", "MODEL.N_HEAD", 2], [2, 1, 3], (256, 192)),    # ONE MHA_ block + the re-viewing of its output (attention.py:991-1031)
    ("tph_win_l12", "tph_192_p6_b4", ["MODEL.ATTENTION_TYPE", "window", "MODEL.N_HEAD", 4], [1, 2], (256, 192)),       # the same without a position embedding, behind TransPose-H
    ("tph2s_up_fk3_l12", "coco_tph_192_p4_b4", ["MODEL.UPSAMPLE_TYPE", "upconv", "MODEL.EXTRA.FINAL_CONV_KERNEL", 3], [1, 2], (256, 192)),  # interformer_2stage.UpConv (upsample_conv.*), both heads 3x3
]


def probe(t, key):
    t = t.detach().float().reshape(-1)
    idx = (synth.uniform01(99, "probe." + key, 32) * t.numel()).astype(np.int64)
    return np.concatenate([[t.mean().item(), t.abs().mean().item()], t[torch.from_numpy(idx)].numpy()]).astype(np.float64)


def flatten_collect(collect):
    out = {}
    for k, v in collect.items():
        if isinstance(v, (list, tuple)):
            for i, t in enumerate(v):
                out["%s.%d" % (k, i)] = t
        else:
            out[k] = v
    return out


def main():
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    nets = {}
    only = set(sys.argv[1:])  # optional: regenerate just these tags
    for tag, cname, opts, length, (H, W), full in [(t, c, None, l, hw, f) for t, c, l, hw, f in CASES] + [(t, c, o, l, hw, True) for t, c, o, l, hw in VARIANTS]:
        if only and tag not in only:
            continue
        cfg = config.load_config(cname, opts)
        nkey = cname if opts is None else tag
        if nkey not in nets:
            net = ref_shim.build_reference_model(cfg)
            spec = synth.spec_of(net)
            sd = synth.make_state_dict(spec)
            net.load_state_dict(sd, strict=True)
            nets[nkey] = (net, sd)
            base = os.path.join(OUT, cname + "_keys.json")
            man = {k: [list(s), d] for k, s, d in spec}
            if opts is None or not os.path.exists(base) or json.load(open(base)) != man:  # (a variant with the base inventory re-uses its manifest)
                with open(os.path.join(OUT, nkey + "_keys.json"), "w") as f:
                    json.dump(man, f, indent=0, sort_keys=True)
        net, sd = nets[nkey]
        x, m, length = synth.make_inputs(length, H, W)
        with torch.no_grad():
            y = net(x, m, length)
        collect = {}
        z = i2r_cpu.forward(sd, cfg, x, m, length, collect)
        outs = y if isinstance(y, dict) else {"multi": y}
        zs = z if isinstance(z, dict) else {"multi": z}
        data = {}
        for k in outs:
            err = (outs[k] - zs[k]).abs().max().item()
            print("%-10s %-7s ref-vs-restatement max-abs %.2e  (|y| mean %.3f)" % (tag, k, err, outs[k].abs().mean().item()))
            assert err < 2e-5 * max(1.0, outs[k].abs().max().item())
            data["probe_out_" + k] = probe(outs[k], tag + k)
            if full:
                data["out_" + k] = outs[k].numpy()
        for k, t in flatten_collect(collect).items():
            data["stage_" + k] = probe(t, tag + k)
        data["length"] = np.asarray(length, dtype=np.int64)
        data["hw"] = np.asarray([H, W], dtype=np.int64)
        data["x_checksum"] = np.asarray([x.double().sum().item(), x.double().abs().sum().item()])
        data["mask_checksum"] = np.asarray([m.double().sum().item()])
        data["w_checksum"] = np.asarray([sum(v.double().abs().sum().item() for v in sd.values() if v.dtype == torch.float32)])
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), **data)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
