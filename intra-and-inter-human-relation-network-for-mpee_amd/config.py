"""Config surface of the hot path: a yacs-free CfgNode work-alike + the defaults the path reads.

The reference drives every model factory with a yacs ``CfgNode`` that is read both by attribute
(``cfg.MODEL.DIM_MODEL``) and by item (``cfg['MODEL']['EXTRA']['STAGE2']``) -- reference
lib/models/interformer_pureMulti.py:423,436-438,457-465 -- and fills it from
lib/config/default.py:17-161 merged with experiments/*.yaml (default.py:164-191).  yacs is not in this
image, so this module provides the same *behaviour* for the keys the inference path touches
(SURVEY.md section 5): nested attr-dict, ``merge_from_file`` / ``merge_from_list`` / ``freeze`` /
``defrost``, and the default values of default.py:36-76 and :129-153.

A real yacs CfgNode (or any object with both access styles) is accepted everywhere a cfg is expected,
so the reference's tools/test.py can pass its own cfg straight into ``models.<NAME>.get_pose_net``.
"""
import ast
import copy
import os

import yaml


class CfgNode(dict):
    """dict with attribute access, nested conversion and a freeze flag (yacs.config.CfgNode subset)."""

    def __init__(self, init=None, new_allowed=False):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        object.__setattr__(self, "_new_allowed", new_allowed)
        for k, v in (init or {}).items():
            dict.__setitem__(self, k, CfgNode(v, new_allowed) if isinstance(v, dict) and not isinstance(v, CfgNode) else v)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if self._frozen:
            raise AttributeError("attempted to set %s on a frozen CfgNode" % name)
        self[name] = value

    def __setitem__(self, key, value):
        if self._frozen:
            raise AttributeError("attempted to set %s on a frozen CfgNode" % key)
        dict.__setitem__(self, key, value)

    def __deepcopy__(self, memo):
        out = CfgNode(new_allowed=self._new_allowed)
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        return out

    def clone(self):
        return copy.deepcopy(self)

    def _set_frozen(self, flag):
        object.__setattr__(self, "_frozen", flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return self._frozen

    def _merge(self, other, path):
        for k, v in other.items():
            here = path + [k]
            if k not in self:
                if self._new_allowed:
                    dict.__setitem__(self, k, CfgNode(v, True) if isinstance(v, dict) else v)
                    continue
                raise KeyError("Non-existent config key: " + ".".join(here))
            cur = self[k]
            if isinstance(cur, CfgNode):
                if not isinstance(v, dict):
                    raise ValueError("config key %s expects a mapping" % ".".join(here))
                cur._merge(v, here)
            else:
                if isinstance(v, str) and cur is not None and not isinstance(cur, str):
                    try:  # yaml reads "(0,)" as a string; yacs literal-evals it
                        v = ast.literal_eval(v)
                    except (ValueError, SyntaxError):
                        pass
                dict.__setitem__(self, k, _coerce(v, cur, ".".join(here)))

    def merge_from_other_cfg(self, other):
        self._merge(other, [])

    def merge_from_file(self, path):
        with open(path, "r") as f:
            data = yaml.safe_load(f) or {}
        self._merge(data, [])

    def merge_from_list(self, opts):
        opts = list(opts or [])
        if len(opts) % 2:
            raise ValueError("override list must be KEY VALUE pairs, got %r" % (opts,))
        for full_key, raw in zip(opts[0::2], opts[1::2]):
            node = self
            parts = full_key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError("Non-existent config key: " + full_key)
                node = node[p]
            leaf = parts[-1]
            if leaf not in node and not node._new_allowed:
                raise KeyError("Non-existent config key: " + full_key)
            val = raw
            if isinstance(raw, str):
                try:
                    val = ast.literal_eval(raw)
                except (ValueError, SyntaxError):
                    val = raw
            dict.__setitem__(node, leaf, _coerce(val, node.get(leaf), full_key) if leaf in node else val)


def _coerce(new, old, key):
    """yacs type rule: the replacement must have the original's type (tuple<->list and None allowed)."""
    if old is None or new is None or type(new) is type(old):
        return new
    if isinstance(old, tuple) and isinstance(new, list):
        return tuple(new)
    if isinstance(old, list) and isinstance(new, tuple):
        return list(new)
    if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
        return float(new)
    if isinstance(old, str) and isinstance(new, str):
        return new
    raise ValueError("type mismatch for config key %s: %r (%s) vs default %r (%s)"
                     % (key, new, type(new).__name__, old, type(old).__name__))


def default_config():
    """Defaults of reference lib/config/default.py (section, key, value identical; line refs inline)."""
    C = CfgNode()
    C.OUTPUT_DIR, C.LOG_DIR, C.DATA_DIR = "", "", ""                     # :19-21
    C.GPUS, C.WORKERS, C.PRINT_FREQ = (0,), 4, 20                        # :22-24
    C.AUTO_RESUME, C.PIN_MEMORY, C.RANK = False, True, 0                 # :25-27
    C.CUDNN = CfgNode(dict(BENCHMARK=True, DETERMINISTIC=False, ENABLED=True))  # :30-33
    C.MODEL = CfgNode(dict(                                               # :36-76
        NAME="interformer", SINGLEFORMER=None, SINGLE_MODEL="", LOSS_WEIGHTS=[0.5, 0.5],
        NORMALIZE_BEFORE=False, END2END=False, BACKBONE_FIX=False, SINGLEFORMER_FIX=False,
        INIT_WEIGHTS=True, PRETRAINED="", NUM_JOINTS=17, TAG_PER_JOINT=True, TARGET_TYPE="gaussian",
        IMAGE_SIZE=[256, 256], HEATMAP_SIZE=[64, 64], TRANS_SIZE=[16, 12], SIGMA=2, HRNET_RES_LAYER=0,
        BOTTLENECK_NUM=0, DIM_MODEL=256, DIM_FEEDFORWARD=512, ENCODER_LAYERS=6, ENCODER_MULTI_LAYERS=4,
        USE_MULTI_POS=True, N_HEAD=8, ATTENTION_ACTIVATION="relu", POS_EMBEDDING="learnable",
        SINGLE_POS_EMBEDDING="sine", INTERMEDIATE_SUP=False, PE_ONLY_AT_BEGIN=False, DOMAIN_TRANS=False,
        INTER_SUPERVISION=True, UPSAMPLE_TYPE="multiplex", MULTI_POS_EMBEDDING="conv",
        ATTENTION_TYPE="default", WINDOW_SIZE=4, MULTI_POS_EMBEDDING_DIM=96))
    dict.__setitem__(C.MODEL, "EXTRA", CfgNode(new_allowed=True))        # :55 (free-form)
    C.LOSS = CfgNode(dict(USE_OHKM=False, TOPK=8, USE_TARGET_WEIGHT=True,
                          USE_DIFFERENT_JOINTS_WEIGHT=False))             # :78-82
    C.DATASET = CfgNode(dict(                                             # :85-104
        ROOT="", DATASET="mpii", TRAIN_SET="train", TEST_SET="valid", DATA_FORMAT="jpg",
        HYBRID_JOINTS_TYPE="", SELECT_DATA=False, MAX_PATCH=7, PATCH_MODE="random", USE_COCOMINI=False,
        FLIP=True, SCALE_FACTOR=0.25, ROT_FACTOR=30, PROB_HALF_BODY=0.0, NUM_JOINTS_HALF_BODY=8,
        COLOR_RGB=False))
    C.TRAIN = CfgNode(dict(                                               # :107-126
        LR_FACTOR=0.1, LR_STEP=[90, 110], LR=0.0001, LR_END=0.00001, OPTIMIZER="adam", MOMENTUM=0.9,
        WD=0.0001, NESTEROV=False, GAMMA1=0.99, GAMMA2=0.0, BEGIN_EPOCH=0, END_EPOCH=140, RESUME=False,
        CHECKPOINT="", BATCH_SIZE_PER_GPU=32, SHUFFLE=True))
    C.TEST = CfgNode(dict(                                                # :129-153
        BLUR_KERNEL=3, BATCH_SIZE_PER_GPU=32, FLIP_TEST=False, POST_PROCESS=False, SHIFT_HEATMAP=False,
        USE_GT_BBOX=False, DETAIL_EVAL=False, IMAGE_THRE=0.1, NMS_THRE=0.6, SOFT_NMS=False, OKS_THRE=0.5,
        IN_VIS_THRE=0.0, COCO_BBOX_FILE="", BBOX_THRE=1.0, MODEL_FILE=""))
    C.DEBUG = CfgNode(dict(DEBUG=False, SAVE_BATCH_IMAGES_GT=False, SAVE_BATCH_IMAGES_PRED=False,
                           SAVE_HEATMAPS_GT=False, SAVE_HEATMAPS_PRED=False))  # :156-161
    return C


CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")

# short names for the shipped workload configs (BASELINE.json configs 1-5) and for the rest of the reference's experiments/*.yaml;
# REFERENCE_YAML maps each name to the reference file it mirrors (checked by tests/test_host.py in the build container)
REFERENCE_YAML = {
    "w48_pure_en6": "crowdpose/interformer_crowdpose_w48_pure_en6", "tph_192_p6_b4": "crowdpose/interformer_crowdpose_tph_192_p6_b4",
    "hrt_192_p4_b4": "crowdpose/interformer_crowdpose_hrt_192_p4_b4", "coco_hrt_288_p2_b4": "coco/interformer_coco_hrt_288_p2_b4",
    "coco_tph_192_p4_b4": "coco/interformer_coco_tph_192_p4_b4", "coco_hrt_192_p2_b12": "coco/interformer_coco_hrt_192_p2_b12",
    "coco_w48_pure_en6": "coco/interformer_coco_w48_pure_en6", "ochuman_tph_192_p3_b8": "OCHuman/interformer_ochuman_tph_192_p3_b8",
    "ochuman_hrt_192_p3_b8": "OCHuman/interformer_ochuman_hrt_192_p3_b8", "ochuman_w48_pure_en6": "OCHuman/interformer_ochuman_w48_pure_en6",
}
NAMED = {
    "w48_pure_en6": "crowdpose_w48_pure_en6.yaml",
    "tph_192_p6_b4": "crowdpose_tph_192_p6_b4.yaml",
    "hrt_192_p4_b4": "crowdpose_hrt_192_p4_b4.yaml",
    "coco_hrt_288_p2_b4": "coco_hrt_288_p2_b4.yaml",
    "coco_tph_192_p4_b4": "coco_tph_192_p4_b4.yaml",
    "ochuman_tph_192_p3_b8": "ochuman_tph_192_p3_b8.yaml",   # MULTI_POS_EMBEDDING 'res' with USE_MULTI_POS true
    "ochuman_hrt_192_p3_b8": "ochuman_hrt_192_p3_b8.yaml",
    "ochuman_w48_pure_en6": "ochuman_w48_pure_en6.yaml",     # vanilla without the multi-position embedding
    "coco_hrt_192_p2_b12": "coco_hrt_192_p2_b12.yaml",
    "coco_w48_pure_en6": "coco_w48_pure_en6.yaml",
    "w48_bare_p6": "crowdpose_w48_bare_p6.yaml",   # interformer without a first stage (bare HRNet backbone)
}


def load_config(path_or_name, opts=None, freeze=True):
    """defaults <- yaml <- ``KEY VALUE`` overrides, like reference default.py:164-191 (update_config).

    ``path_or_name`` is a yaml path (the reference's experiments/*.yaml work unchanged) or one of NAMED.
    """
    path = path_or_name
    if path_or_name in NAMED:
        path = os.path.join(CONFIG_DIR, NAMED[path_or_name])
    cfg = default_config()
    cfg.merge_from_file(path)
    cfg.merge_from_list(opts)
    if freeze:
        cfg.freeze()
    return cfg
