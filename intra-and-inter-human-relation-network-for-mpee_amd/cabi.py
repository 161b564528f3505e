"""ctypes binding of the C-ABI library libi2r_hip.so (declared in include/i2r_hip.h).

This is the ONLY way the product path reaches arithmetic: there is no CPU / eager fallback.  If the
shared library is missing, or the visible GPU is not gfx950, ``lib()`` raises -- loudly.

Build: ``__graft_entry__.build()`` (hipcc --offload-arch=gfx950, in-tree .so).
"""
import ctypes as C
import os

import torch  # noqa: F401  -- must be loaded BEFORE libi2r_hip.so so both share torch's HIP runtime (libamdhip64) instance

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libi2r_hip.so")

MAX_TAPS = 9
ABI_VERSION = 14  # I2R_ABI_VERSION of include/i2r_hip.h
OP_CONV, OP_STEM, OP_MAXPOOL, OP_HEAD, OP_ENC_KV, OP_ENC_LAYER, OP_FORK, OP_JOIN, OP_CONV_GROUP = 1, 2, 3, 4, 5, 6, 7, 8, 9
MAX_GROUP = 4
OP_LAYERNORM, OP_WINATTN, OP_DWCONV, OP_UPSAMPLE = 10, 11, 12, 13
OP_CONV_CHAIN = 14
OP_PE_RES_STEM = 15
OP_HRT_ATTN = 16
OP_HRT_MLP = 17
OP_XSYNC = 18
OP_FUSE_UP = 19
OP_CONV1X1_PAIR = 20
OP_CONV1X1_LP = 21
OP_MH_ATTN = 22
OP_PE_CAT_VEC = 23
OP_ROWS_GATHER, OP_VIEW_SCRAMBLE = 24, 25
OP_RECORD, OP_WAIT = 26, 27  # point-to-point: lane field = lane | slot << 8 (| consumer lanes << 16 for RECORD)
OP_LANE_FLAGS = 28          # args = device int32[64] flag buffer: the sync ops behind it run as device-side signal / wait kernels
SYNC_OPS = (OP_FORK, OP_JOIN, OP_XSYNC, OP_RECORD, OP_WAIT, OP_LANE_FLAGS)  # ops that launch nothing (FORK / JOIN / XSYNC: `lane` is a lane mask)

_fp = C.c_void_p  # device pointers travel as integers
_i32 = C.c_int32


class ConvDesc(C.Structure):
    _fields_ = [
        ("in_", _fp), ("in2", _fp), ("w", _fp), ("bias", _fp), ("res1", _fp), ("res2", _fp), ("res_post", _fp), ("out", _fp),
        ("n_img", _i32), ("in_h", _i32), ("in_w", _i32), ("in_cs", _i32), ("cin", _i32),
        ("conv_h", _i32), ("conv_w", _i32), ("out_h", _i32), ("out_w", _i32), ("out_cs", _i32),
        ("cout", _i32), ("cout_pad", _i32), ("stride", _i32), ("iy0", _i32), ("ix0", _i32), ("ntaps", _i32),
        ("dy", _i32 * MAX_TAPS), ("dx", _i32 * MAX_TAPS),
        ("out_step", _i32), ("out_off_y", _i32), ("out_off_x", _i32), ("rep", _i32), ("relu", _i32),
        ("tile_h", _i32), ("tile_w", _i32), ("ck", _i32), ("wn", _i32), ("mt", _i32), ("dtype", _i32),
        ("in_f16", _i32), ("out_f16", _i32), ("algo", _i32),
    ]


class EncoderDesc(C.Structure):
    _fields_ = [
        ("src", _fp), ("pos", _fp), ("kbuf", _fp), ("vbuf", _fp), ("out", _fp), ("grp_off", _fp),
        ("w_in", _fp), ("b_in", _fp), ("w_out", _fp), ("b_out", _fp), ("ln1_w", _fp), ("ln1_b", _fp),
        ("w1", _fp), ("b1", _fp), ("w2", _fp), ("b2", _fp), ("ln2_w", _fp), ("ln2_b", _fp),
        ("n_tok", _i32), ("n_grp", _i32), ("d", _i32), ("cs", _i32), ("dff_pad", _i32), ("pos_period", _i32),
        ("n_qtiles32", _i32), ("ln_eps", C.c_float), ("dtype", _i32), ("n_qtiles16", _i32), ("n_qtiles64", _i32),
        ("w_in_lp", _fp), ("w_out_lp", _fp), ("w1_lp", _fp), ("w2_lp", _fp),
        ("next_w_in", _fp), ("next_b_in", _fp), ("next_kbuf", _fp), ("next_vbuf", _fp), ("vec_lp", _fp),
        ("split_ws", _fp), ("split_cnt", _fp), ("n_qtiles192", _i32),
    ]


class StemArgs(C.Structure):
    _fields_ = [("in_", _fp), ("w", _fp), ("bias", _fp), ("out", _fp),
                ("n_img", _i32), ("cin", _i32), ("in_h", _i32), ("in_w", _i32), ("cout", _i32), ("out_cs", _i32), ("n_src", _i32),
                ("n_valid", _i32), ("out_dt", _i32)]


class PeResArgs(C.Structure):
    _fields_ = [("in_", _fp), ("w_pre", _fp), ("w7", _fp), ("bias", _fp), ("out", _fp),
                ("n_img", _i32), ("in_h", _i32), ("in_w", _i32), ("cout", _i32), ("out_cs", _i32), ("n_src", _i32), ("n_valid", _i32)]


class PeCatVecArgs(C.Structure):
    _fields_ = [("in_", _fp), ("w", _fp), ("bias", _fp), ("out", _fp),
                ("n_img", _i32), ("in_h", _i32), ("in_w", _i32), ("th", _i32), ("tw", _i32), ("rate", _i32), ("vec", _i32), ("out_cs", _i32),
                ("c0", _i32), ("c_end", _i32), ("n_src", _i32), ("n_valid", _i32)]


class PoolArgs(C.Structure):
    _fields_ = [("in_", _fp), ("out", _fp),
                ("n_img", _i32), ("in_h", _i32), ("in_w", _i32), ("c", _i32), ("in_cs", _i32), ("out_cs", _i32)]


class HeadArgs(C.Structure):
    _fields_ = [("in_", _fp), ("w", _fp), ("bias", _fp), ("out", _fp),
                ("n_img", _i32), ("h", _i32), ("w_", _i32), ("cin", _i32), ("in_cs", _i32), ("cout", _i32)]


class LnArgs(C.Structure):
    _fields_ = [("in_", _fp), ("w", _fp), ("b", _fp), ("out", _fp), ("npix", _i32), ("c", _i32), ("cs", _i32), ("eps", C.c_float),
                ("out_dt", _i32)]


class MhAttnArgs(C.Structure):
    _fields_ = [("qk", _fp), ("v", _fp), ("out", _fp), ("grp_off", _fp),
                ("n_grp", _i32), ("heads", _i32), ("hp", _i32), ("k_off", _i32), ("qk_cs", _i32), ("v_cs", _i32), ("out_cs", _i32),
                ("n_qtiles16", _i32), ("n_qtiles32", _i32), ("n_qtiles64", _i32), ("key_len", _fp)]


class GatherArgs(C.Structure):
    _fields_ = [("src", _fp), ("out", _fp), ("map", _fp), ("n_out", _i32), ("floats_per_crop", _i32)]


class ScrambleArgs(C.Structure):
    _fields_ = [("o", _fp), ("out", _fp), ("person_map", _fp), ("n_out", _i32), ("n_images", _i32), ("max_persons", _i32), ("c", _i32), ("cs", _i32),
                ("hw", _i32)]


class WinAttnArgs(C.Structure):
    _fields_ = [("qkv", _fp), ("bias", _fp), ("out", _fp),
                ("n_img", _i32), ("h", _i32), ("w_", _i32), ("c", _i32), ("cs", _i32), ("heads", _i32)]


class HrtAttnArgs(C.Structure):
    _fields_ = [("x", _fp), ("out", _fp), ("ln_w", _fp), ("ln_b", _fp), ("wqkv", _fp), ("bqkv", _fp), ("wo", _fp), ("bo", _fp),
                ("n_img", _i32), ("h", _i32), ("w_", _i32), ("c", _i32), ("cs", _i32), ("heads", _i32), ("eps", C.c_float), ("dtype", _i32),
                ("variant", _i32)]


class HrtMlpArgs(C.Structure):
    _fields_ = [("x", _fp), ("out", _fp), ("ln_w", _fp), ("ln_b", _fp), ("w1", _fp), ("b1", _fp), ("wdw", _fp), ("bdw", _fp), ("w2", _fp),
                ("b2", _fp), ("n_img", _i32), ("h", _i32), ("w_", _i32), ("c", _i32), ("cs", _i32), ("hidden_pad", _i32), ("eps", C.c_float),
                ("dtype", _i32), ("variant", _i32)]


class DwArgs(C.Structure):
    _fields_ = [("in_", _fp), ("w", _fp), ("bias", _fp), ("out", _fp),
                ("n_img", _i32), ("in_h", _i32), ("in_w", _i32), ("c", _i32), ("cs", _i32), ("stride", _i32), ("act", _i32), ("dt", _i32)]


class UpArgs(C.Structure):
    _fields_ = [("low", _fp), ("res", _fp), ("out", _fp),
                ("n_img", _i32), ("low_h", _i32), ("low_w", _i32), ("scale", _i32), ("c", _i32), ("cs", _i32), ("act", _i32),
                ("low2", _fp), ("low3", _fp), ("scale2", _i32), ("scale3", _i32)]


class FuseUpArgs(C.Structure):
    _fields_ = [("base", _fp), ("t1", _fp), ("t2", _fp), ("out", _fp),
                ("n_img", _i32), ("h", _i32), ("w", _i32), ("cs", _i32), ("s1", _i32), ("s2", _i32), ("act", _i32), ("dt", _i32)]


class Conv1x1PairArgs(C.Structure):
    _fields_ = [("x", _fp), ("w_a", _fp), ("b_a", _fp), ("res", _fp), ("y", _fp), ("w_b", _fp), ("b_b", _fp), ("z", _fp),
                ("n_pix", _i32), ("k_a", _i32), ("ca_out", _i32), ("cb_out", _i32), ("x_cs", _i32), ("y_cs", _i32), ("z_cs", _i32),
                ("relu_a", _i32), ("relu_b", _i32), ("mt", _i32)]


class Conv1x1LpArgs(C.Structure):
    _fields_ = [("x", _fp), ("w", _fp), ("bias", _fp), ("res1", _fp), ("res_post", _fp), ("out", _fp),
                ("n_pix", _i32), ("cin_pad", _i32), ("cout_pad", _i32), ("x_cs", _i32), ("out_cs", _i32), ("act", _i32), ("dtype", _i32),
                ("in_16", _i32), ("out_16", _i32), ("mt", _i32), ("res2", _fp)]


class ConvGroupArgs(C.Structure):
    _fields_ = [("d", C.POINTER(ConvDesc) * MAX_GROUP), ("block_map", _fp), ("n", _i32), ("map_len", _i32)]


class ConvChainArgs(C.Structure):
    _fields_ = [("descs", C.POINTER(C.POINTER(ConvDesc))), ("n_layers", _i32), ("n_members", _i32),
                ("kdesc", _fp), ("item_ofs", _fp), ("items", _fp), ("flags", _fp), ("n_blocks", _i32),
                ("n_flags", _i32), ("kdesc_bytes", _i32), ("capacity", _i32), ("nt", _i32), ("mt", _i32), ("cap", _i32), ("pf", _i32),
                ("lds_bytes", _i32), ("tiles", (_i32 * 4) * MAX_GROUP)]


class ImageRef(C.Structure):   # i2r_image_ref (24 bytes)
    _fields_ = [("img", C.c_void_p), ("ih", _i32), ("iw", _i32), ("row_bytes", _i32), ("reserved", _i32)]


class CropRef(C.Structure):    # i2r_crop_ref (80 bytes)
    _fields_ = [("inv_m", C.c_double * 6), ("box", _i32 * 4), ("image", _i32), ("reserved", _i32 * 3)]


class Op(C.Structure):
    _fields_ = [("kind", _i32), ("lane", _i32), ("args", C.c_void_p)]


# every symbol include/i2r_hip.h declares (tests/test_host.py::test_cabi_library_exports_every_declared_symbol checks the built library exports them all)
EXPORTS = ("i2r_conv", "i2r_conv_grouped", "i2r_conv_kernel_name", "i2r_stem_conv", "i2r_pe_res_stem", "i2r_maxpool3x3s2", "i2r_head", "i2r_layernorm", "i2r_window_attn", "i2r_hrt_attn_block", "i2r_hrt_mlp_block", "i2r_dwconv3x3",
           "i2r_upsample_bilinear_add", "i2r_upsample_bilinear_add_multi", "i2r_fuse_up_add", "i2r_conv1x1_pair", "i2r_conv1x1_lp", "i2r_flip_merge", "i2r_decode", "i2r_crop_affine", "i2r_box_mask", "i2r_crop_affine_cv2", "i2r_box_mask_cv2", "i2r_person_inputs_cv2", "i2r_conv_chain_pack", "i2r_conv_chain", "i2r_encoder_kv", "i2r_encoder_layer", "i2r_mh_attention", "i2r_pe_cat_vec", "i2r_rows_gather", "i2r_view_scramble",
           "i2r_run_program", "i2r_run_program_timed", "i2r_abi_version", "i2r_last_error", "i2r_device_check")

_LIB = None


class I2RError(RuntimeError):
    pass


def load_library(path=LIB_PATH):
    """dlopen the library and set prototypes. No GPU needed (used by the CPU-side ABI test)."""
    if not os.path.exists(path):
        raise I2RError(
            "HIP extension %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU fallback for the product path." % path)
    L = C.CDLL(path)
    L.i2r_conv.argtypes = [C.POINTER(ConvDesc), C.c_void_p]
    L.i2r_conv_grouped.argtypes = [C.POINTER(C.POINTER(ConvDesc)), _i32, _fp, _i32, C.c_void_p]
    L.i2r_conv_chain_pack.argtypes = [C.POINTER(ConvChainArgs), C.c_void_p, C.c_int64]
    L.i2r_conv_chain.argtypes = [C.POINTER(ConvChainArgs), C.c_void_p]
    L.i2r_conv_kernel_name.argtypes = [C.POINTER(C.POINTER(ConvDesc)), _i32, C.c_char_p, _i32]
    L.i2r_stem_conv.argtypes = [_fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, C.c_void_p]
    L.i2r_pe_res_stem.argtypes = [_fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, C.c_void_p]
    L.i2r_flip_merge.argtypes = [_fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, C.c_void_p]
    L.i2r_decode.argtypes = [_fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, C.c_void_p]
    L.i2r_crop_affine.argtypes = [_fp, _i32, _i32, _i32, _i32, _fp, _fp, _fp, _fp, _i32, _i32, _i32, C.c_void_p]
    L.i2r_box_mask.argtypes = [_fp, _i32, _i32, _fp, _i32, _i32, _i32, C.c_void_p]
    L.i2r_crop_affine_cv2.argtypes = [_fp, _i32, _i32, _i32, _i32, _fp, _fp, _fp, _fp, _i32, _i32, _i32, C.c_void_p]
    L.i2r_box_mask_cv2.argtypes = [_fp, _i32, _i32, _fp, _i32, _i32, _i32, C.c_void_p]
    L.i2r_person_inputs_cv2.argtypes = [_fp, _i32, _fp, _i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float), _fp, _fp, _i32, _i32, C.c_void_p]
    L.i2r_fuse_up_add.argtypes = [_fp, _fp, _i32, _fp, _i32, _fp, _i32, _i32, _i32, _i32, _i32, _i32, C.c_void_p]
    L.i2r_conv1x1_pair.argtypes = [C.POINTER(Conv1x1PairArgs), C.c_void_p]
    L.i2r_conv1x1_lp.argtypes = [C.POINTER(Conv1x1LpArgs), C.c_void_p]
    L.i2r_maxpool3x3s2.argtypes = [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, C.c_void_p]
    L.i2r_head.argtypes = [_fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, C.c_void_p]
    L.i2r_layernorm.argtypes = [_fp, _fp, _fp, _fp, _i32, _i32, _i32, C.c_float, _i32, C.c_void_p]
    L.i2r_window_attn.argtypes = [_fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, C.c_void_p]
    L.i2r_hrt_attn_block.argtypes = [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, C.c_float, _i32, _i32, C.c_void_p]
    L.i2r_hrt_mlp_block.argtypes = [_fp] * 10 + [_i32] * 6 + [C.c_float, _i32, _i32, C.c_void_p]
    L.i2r_dwconv3x3.argtypes = [_fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, C.c_void_p]
    L.i2r_upsample_bilinear_add_multi.argtypes = [C.POINTER(UpArgs), C.c_void_p]
    L.i2r_upsample_bilinear_add.argtypes = [_fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, C.c_void_p]
    L.i2r_encoder_kv.argtypes = [C.POINTER(EncoderDesc), C.c_void_p]
    L.i2r_encoder_layer.argtypes = [C.POINTER(EncoderDesc), C.c_void_p]
    L.i2r_rows_gather.argtypes = [_fp, _fp, _fp, _i32, _i32, C.c_void_p]
    L.i2r_view_scramble.argtypes = [_fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, C.c_void_p]
    L.i2r_pe_cat_vec.argtypes = [C.POINTER(PeCatVecArgs), C.c_void_p]
    L.i2r_mh_attention.argtypes = [C.POINTER(MhAttnArgs), C.c_void_p]
    L.i2r_run_program.argtypes = [C.POINTER(Op), _i32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.i2r_run_program_timed.argtypes = [C.POINTER(Op), _i32] + [C.POINTER(C.c_void_p)] * 4
    L.i2r_device_check.argtypes = [_i32, C.POINTER(_i32), C.POINTER(_i32)]
    L.i2r_last_error.restype = C.c_char_p
    for name in EXPORTS:
        if name != "i2r_last_error":
            getattr(L, name).restype = C.c_int
    return L


def lib():
    """The loaded library, verified against the ABI version (and cached)."""
    global _LIB
    if _LIB is None:
        L = load_library()
        if L.i2r_abi_version() != ABI_VERSION:
            raise I2RError("libi2r_hip.so ABI version %d != %d -- rebuild" % (L.i2r_abi_version(), ABI_VERSION))
        _LIB = L
    return _LIB


def check(rc, what=""):
    if rc != 0:
        msg = lib().i2r_last_error()
        raise I2RError("%s failed (rc=%d): %s" % (what or "i2r call", rc, msg.decode() if msg else "?"))


def require_gfx950(device_index=0):
    cu, lds = _i32(0), _i32(0)
    check(lib().i2r_device_check(device_index, C.byref(cu), C.byref(lds)), "i2r_device_check")
    return cu.value, lds.value
