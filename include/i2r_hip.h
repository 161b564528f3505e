/*
 * i2r_hip.h -- C-ABI of the MI355X-native I2R-Net inference hot path (libi2r_hip.so, gfx950 only).
 *
 * The reference (leijue222/Intra-and-Inter-Human-Relation-Network-for-MPEE) has NO FFI on this path: its
 * boundary is the Python module contract  models.<MODEL.NAME>.get_pose_net(cfg, is_train) ->
 * nn.Module,  called as  model(input, pos_mask, length)  (lib/core/function.py:135, tools/test.py:87).
 * Every arithmetic op behind that call is issued by torch.nn modules.  This header declares the
 * entry points a maintainer binds instead (ctypes stub in INTEGRATION.md); each one names the
 * reference construct it replaces.
 *
 * Conventions
 *  - plain pointers + sizes only; all pointers are DEVICE pointers owned by the caller (inputs, outputs,
 *    packed weights, workspace); the library allocates nothing and keeps no global state.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous and
 *    re-entrant per stream.
 *  - return 0 on success, negative I2R_E_* otherwise; i2r_last_error() gives a thread-local message.
 *  - activations are fp32 NHWC ("pixel-major, channel-minor") with an explicit channel stride; the
 *    boundary tensors keep the reference's NCHW fp32 layout (i2r_stem_conv reads NCHW, i2r_head writes NCHW).
 */
#ifndef I2R_HIP_H
#define I2R_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define I2R_ABI_VERSION 14

/* The library is built with -fvisibility=hidden: the entry points declared in this header (marked I2R_API) are its ONLY exported
 * symbols (tests/test_host.py holds the header, the dynamic symbol table and cabi.EXPORTS equal). */
#define I2R_API __attribute__((visibility("default")))

#define I2R_OK 0
#define I2R_E_ARG (-1)      /* bad argument (shape / alignment / unsupported combination) */
#define I2R_E_LAUNCH (-2)   /* HIP launch failure */
#define I2R_E_NODEV (-3)    /* no gfx950 device */

#define I2R_MAX_TAPS 9

/* ------------------------------------------------------------------------------------------------
 * i2r_conv_desc -- one fused  conv (implicit GEMM on fp32 MFMA) [+ folded BN bias] [+ residual(s)]
 * [+ ReLU] [+ nearest-upsample scatter]  launch.
 * Replaces: nn.Conv2d + nn.BatchNorm2d(eval) + nn.ReLU / residual add / nn.Upsample(nearest) /
 * the sum of HighResolutionModule.forward (reference lib/models/interformer_pureMulti.py:50-66,
 * 87-107, 332-410), nn.ConvTranspose2d(k4,s2,p1) split into its 4 output-parity 2x2 convs
 * (interformer_pureMulti.py:648-673, interformer.py:84-122), and nn.Linear on token matrices
 * (a 1x1 conv over [T,1,1,C]).
 *
 *   acc[p, co] = sum_{t < ntaps} sum_{ci < cin}  IN[n, oy*stride + iy0 + dy[t], ox*stride + ix0 + dx[t], ci]
 *                                               * W[t][ci][co]              (IN = in (+ in2 if given))
 *   v = acc + bias[co];  v += res1[dst, co] (if res1);  v += res2[dst, co] (if res2);  relu (if relu);
 *   v += res_post[dst, co] (if res_post: residual added AFTER the ReLU, interformer.py:315)
 *   dst pixels: (oy*out_step + out_off_y + ry, ox*out_step + out_off_x + rx) for ry,rx < rep
 *
 * Packed weight layout ("k4"): float w[ntaps][cin/4][cout_pad][4]  (cin % 16 == 0, cout_pad % 16 == 0,
 * zero-filled beyond cout); bias[cout_pad].
 * ------------------------------------------------------------------------------------------------ */
typedef struct i2r_conv_desc {
    const float* in;     /* [n_img, in_h, in_w, in_cs] */
    const float* in2;    /* optional second input added element-wise while staging (same geometry) */
    const float* w;      /* packed k4 weights */
    const float* bias;   /* [cout_pad] */
    const float* res1;   /* optional, laid out like out */
    const float* res2;   /* optional, laid out like out */
    const float* res_post; /* optional, laid out like out, added after the ReLU */
    float* out;          /* [n_img, out_h, out_w, out_cs] */
    int32_t n_img;
    int32_t in_h, in_w, in_cs;     /* in_cs = channel stride (floats) per input pixel */
    int32_t cin;                   /* multiple of 16, <= in_cs */
    int32_t conv_h, conv_w;        /* logical conv output grid (oy, ox ranges) */
    int32_t out_h, out_w, out_cs;  /* destination tensor geometry */
    int32_t cout, cout_pad;        /* real / padded output channels */
    int32_t stride;                /* 1 or 2 */
    int32_t iy0, ix0;              /* input coordinate of tap offset (0,0) for output (0,0): -pad */
    int32_t ntaps;
    int32_t dy[I2R_MAX_TAPS], dx[I2R_MAX_TAPS];
    int32_t out_step, out_off_y, out_off_x, rep;
    int32_t relu;                  /* activation: 0 none, 1 ReLU, 2 exact-erf GELU */
    int32_t tile_h, tile_w;        /* output tile per workgroup (0 = let the library choose) */
    int32_t ck;                    /* input channels staged in LDS per pass (0 = choose) */
    int32_t wn;                    /* waves along cout in the 4-wave workgroup: 1, 2 or 4 (0 = choose) */
    int32_t mt;                    /* 16-pixel fragments per wave, 1..4 (0 = derive from the tile) */
    int32_t dtype;                 /* MFMA operand type: 0 fp32 (w = k4 fp32), 1 bf16, 2 f16 (w = "k8" [tap][cin32/8][cout_pad][8],
                                      cin zero-padded to a multiple of 32; accumulate fp32) */
    int32_t in_f16, out_f16;       /* dtype != 0 only: activation STORAGE of `in` / of `out`, res1, res2 and res_post -- 0 fp32,
                                      1 = 16 bit of the operand type (bf16 / f16; the pointers then address 2-byte elements, all strides
                                      and offsets stay in elements).  The conv towers of the 16-bit modes keep their maps in 16 bit. */
    int32_t algo;                  /* 0 = direct implicit GEMM.  1 = Winograd F(2x2, 3x3): dtype 0, dense 3x3 taps with iy0 = ix0 = -1, stride 1,
                                      rep = out_step = 1, no in2; `w` then holds the TRANSFORMED weights U = G g G^T in the k4 layout with the 16
                                      Winograd positions (row-major 4x4) in place of the taps: float w[16][cin/4][cout_pad][4].  tile_w selects the
                                      fragment shape (16 Winograd tiles = 64 output pixels): 16, 8 or 4 pixels wide (0 = choose); mt = fragments per
                                      workgroup (1 or 2, 0 = 1; 2 only for cout_pad / 16 a multiple of 3).  2.25x fewer matrix-pipe operations than algo 0 for the same sum. */
} i2r_conv_desc;

I2R_API int i2r_conv(const i2r_conv_desc* d, void* stream);

/* i2r_conv_grouped -- up to I2R_MAX_GROUP independent convolutions in ONE launch ("horizontal fusion"): the
 * parallel branches of a HighResolutionModule (interformer_pureMulti.py:396-397) and the same-depth terms of its
 * fuse sums (:401-408) have no mutual dependencies, and individually the low-resolution ones cannot fill 256 CUs.
 * All descriptors must resolve to the same fragment blocking: cout_pad/16 divisible by the same NT, same mt. */
#define I2R_MAX_GROUP 4
/* block_map (optional, device int32[map_len]): dispatch order of the workgroups, entry = (member << 24) | index of
 * the workgroup within that member; map_len must equal the total workgroup count
 * (sum over members of n_img * ceil(conv_h/tile_h) * ceil(conv_w/tile_w) * cout_blocks). */
I2R_API int i2r_conv_grouped(const i2r_conv_desc* const* descs, int32_t n, const int32_t* block_map, int32_t map_len,
                     void* stream);

/* i2r_conv_chain (EXPERIMENTAL: correct and tested, but measured slower than one i2r_conv_grouped per layer on MI355X at 32 crops;
 * the host side keeps it behind I2R_CONV_CHAIN=1) -- n_layers DEPENDENT stride-1 convolutions (layer l of member g reads layer l-1's output of member g) for up to
 * I2R_MAX_GROUP independent members in ONE persistent launch: the 8 convs of the 4 BasicBlocks of every branch of a
 * HighResolutionModule (interformer_pureMulti.py:392-397).  Instead of a chip-wide barrier per layer, a tile starts as soon as the
 * 3x3 tile neighbourhood of the previous layer has finished (completion counters in `flags`), which removes the per-launch
 * fill / drain and round quantisation.  descs: host array [n_layers * n_members], layer-major; every layer of a member must resolve
 * to the same tiling.  Workspaces are the caller's:
 *   i2r_conv_chain_pack(a, host_buf, bytes)  validates, fills the outputs below and (host_buf != NULL) writes the kernel-side
 *       descriptors into host_buf (a->kdesc_bytes bytes) for the caller to copy to the device (a->kdesc);
 *   items / item_ofs: eight work queues, one per XCD -- the persistent workgroups with (index % 8) == x pop
 *       items[item_ofs[x] .. item_ofs[x+1]) in order; each item = (layer << 26) | (member << 24) | workgroup index within the member
 *       (numbered as i2r_conv_grouped numbers them: cout block fastest, then tile x, tile y, image); every queue must be sorted
 *       by layer and hold ALL items of its images (producers and consumers then share one L2); n_blocks <= a->capacity;
 *   flags: n_flags + 17 int32 (zeroed by i2r_conv_chain itself; word [n_flags] afterwards: 0 ok, 1 = a dependency wait timed out,
 *       2 = workgroups of one residue class (index % 8) ran on different XCDs, i.e. the same-L2 assumption of the schedule is void). */
typedef struct i2r_conv_chain_args {
    const i2r_conv_desc* const* descs;
    int32_t n_layers, n_members;
    const void* kdesc;
    const int32_t* item_ofs;
    const int32_t* items;
    int32_t* flags;
    int32_t n_blocks;
    /* outputs of i2r_conv_chain_pack */
    int32_t n_flags, kdesc_bytes, capacity, nt, mt, cap, pf, lds_bytes;
    int32_t tiles[I2R_MAX_GROUP][4];  /* per member: tiles_y, tiles_x, cout blocks, workgroups per layer */
} i2r_conv_chain_args;
I2R_API int i2r_conv_chain_pack(i2r_conv_chain_args* a, void* host_buf, int64_t host_bytes);
I2R_API int i2r_conv_chain(const i2r_conv_chain_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * i2r_stem_conv -- first 3x3 stride-2 pad-1 conv of a tower (cin = 1..4) + folded BN + ReLU, reading the
 * boundary NCHW fp32 tensor and writing NHWC.
 * Replaces: conv1+bn1+relu (interformer_pureMulti.py:677-679) and PositionEmbeddingImage conv1+bn1+relu
 * (position_embedding.py:99-101).   w: float[9][cin][cout], bias[cout]; cout % 16 == 0.
 * n_src: the input holds n_src images; output images n_src..n_img-1 (n_img == 2*n_src) are computed from the
 * horizontally mirrored input (np.flip(input, 3) of the flip test, lib/core/function.py:145-150); n_src == n_img: none.
 * n_valid (1..n_src): the input tensor really holds n_valid images; output slots n_valid..n_src-1 (capacity padding of a
 * pre-built launch program) are computed from image n_valid-1 and are the caller's to drop.
 * ------------------------------------------------------------------------------------------------ */
I2R_API int i2r_stem_conv(const float* in_nchw, const float* w, const float* bias, float* out_nhwc, int32_t n_img,
                  int32_t cin, int32_t in_h, int32_t in_w, int32_t cout, int32_t out_cs, int32_t n_src, int32_t n_valid,
                  int32_t out_dt /* storage of out: 0 fp32, 1 bf16, 2 f16 (see i2r_conv_desc.in_f16) */, void* stream);

/* i2r_pe_res_stem -- front end of PositionEmbeddingImage mode 'res' (lib/models/position_embedding.py:14-17,93-95):
 * conv_pre (nn.Conv2d(1, 3, 3, padding=1, bias=False)) followed by torchvision resnet18's conv1 (3 -> 64, 7x7, stride 2, pad 3,
 * no bias) + bn1 (eval, folded by the host) + ReLU, reading the boundary NCHW bbox mask [n_src, 1, in_h, in_w] and writing NHWC
 * [n_img, in_h/2, in_w/2, out_cs].  w_pre: float[9][3] (tap-major), w7: float[49][3][64] (tap, cin, cout) with the BN scale
 * folded in, bias[64].  n_src / n_valid as in i2r_stem_conv (mirrored copies for the flip test, capacity padding). */
I2R_API int i2r_pe_res_stem(const float* mask_nchw, const float* w_pre, const float* w7, const float* bias, float* out_nhwc,
                    int32_t n_img, int32_t in_h, int32_t in_w, int32_t cout, int32_t out_cs, int32_t n_src, int32_t n_valid, void* stream);

/* i2r_pe_cat_vec -- PositionEmbeddingImage mode 'cat_vec' (position_embedding.py:19-23,69-87): per person, the boundary NCHW bbox mask
 * [n_src, 1, in_h, in_w] max-pooled `rate` times (MaxPool2d(3, 2, 1)) down to th x tw, flattened, through nn.Linear(th*tw, vec)
 * (w: float[vec][th*tw] as the reference stores it, bias[vec]), the resulting vector repeated over the th*tw tokens of the person:
 * written into channels [c0, c0 + vec) of the NHWC token rows out [n_img, th, tw, out_cs]; channels [c0 + vec, c_end) get zeros.
 * c0 = 0: the additive embedding of interformer_pureMulti.py:757,770 / interformer_2stage.py:398-406 (vec = DIM_MODEL there);
 * c0 = DIM_MODEL: the second half of torch.cat([x, multi_pos], dim=2) in interformer.py:296-298 (the first half is written by
 * the launch that produces x).  n_src / n_valid as in i2r_stem_conv (mirrored copies for the flip test, capacity padding). */
typedef struct i2r_pe_cat_vec_args {
    const float* in; const float* w; const float* bias; float* out;
    int32_t n_img, in_h, in_w, th, tw, rate, vec, out_cs, c0, c_end, n_src, n_valid;
} i2r_pe_cat_vec_args;
I2R_API int i2r_pe_cat_vec(const i2r_pe_cat_vec_args* a, void* stream);

/* i2r_maxpool3x3s2 -- nn.MaxPool2d(kernel_size=3, stride=2, padding=1) on NHWC
 * (interformer.py:162,260-264; position_embedding.py:9,106-109).  c % 4 == 0. */
I2R_API int i2r_maxpool3x3s2(const float* in, float* out, int32_t n_img, int32_t in_h, int32_t in_w, int32_t c,
                     int32_t in_cs, int32_t out_cs, void* stream);

/* i2r_head -- final 1x1 conv (with bias) NHWC -> boundary NCHW heatmaps
 * (final_layer, interformer_pureMulti.py:486-492,776; interformer.py:176-182,317).
 * w: float[cout][cin] (the nn.Conv2d weight as is), bias[cout]. out: [n_img, cout, h, w]. */
I2R_API int i2r_head(const float* in, const float* w, const float* bias, float* out_nchw, int32_t n_img, int32_t h,
             int32_t w_, int32_t cin, int32_t in_cs, int32_t cout, void* stream);

/* i2r_conv_kernel_name -- which instantiation conv_igemm_f32<MT, NT, CAP, PF> a (grouped) launch resolves to, as it
 * appears in rocprofv3 kernel traces (used by bench.py to key its per-kernel roofline numbers). No launch happens. */
I2R_API int i2r_conv_kernel_name(const i2r_conv_desc* const* descs, int32_t n, char* buf, int32_t buflen);

/* ---- after the forward: flip-test merge and keypoint decode -------------------------------- */
/* i2r_flip_merge -- out = (y + flip_back(y_flipped)) * 0.5  with flip_back = reverse W + swap left/right joints
 * (lib/core/function.py:142-162, lib/utils/transforms.py:16-30). joint_map: device int32[joints], the joint whose mirrored
 * heatmap lands in channel j (identity for unpaired joints). NCHW fp32 [n, joints, h, w]. */
I2R_API int i2r_flip_merge(const float* y, const float* y_flipped, const int32_t* joint_map, float* out, int32_t n, int32_t joints,
                   int32_t h, int32_t w, void* stream);

/* i2r_decode -- get_final_preds (lib/core/inference.py:90-112): arg-max (:20-48), Gaussian blur with kernel TEST.BLUR_KERNEL
 * on a zero-bordered copy re-normalised to the original max (:73-87; cv2.GaussianBlur(sigma=0) => sigma =
 * 0.3*((k-1)*0.5-1)+0.8), log(max(.,1e-10)), second-order Taylor refinement (:51-70), inverse crop affine with rot 0
 * (lib/utils/transforms.py:50-101: scale about the centres by (scale[0]*200-1)/(w-1)).
 * heatmaps [n, joints, h, w]; center, scale [n, 2] (unused when transform_back == 0); preds [n, joints, 2]; maxvals [n, joints]. */
I2R_API int i2r_decode(const float* heatmaps, const float* center, const float* scale, float* preds, float* maxvals, int32_t n,
               int32_t joints, int32_t h, int32_t w, int32_t blur_kernel, int32_t transform_back, void* stream);

/* ---- input side (reference lib/dataset/JointsDataset.py:296-333) ------------------------
 * i2r_crop_affine -- cv2.warpAffine(image, trans, IMAGE_SIZE, INTER_LINEAR) + ToTensor + Normalize for the n persons of ONE
 * image: out[p][c][y][x] = (bilinear(img, M_p (x, y, 1)^T) / 255 - mean[c]) * inv_std[c], taps outside the image = 0.
 * img: device uint8 [ih, iw, 3] with row pitch row_bytes (HWC as cv2.imread delivers it; swap_rb = DATASET.COLOR_RGB);
 * inv_trans: device float [n, 6] = the INVERSE (input pixel -> image pixel) of get_affine_transform(center, scale, 0, size)
 * (lib/utils/transforms.py:61-96); out: [n, 3, oh, ow] fp32.  fp32 interpolation, NOT cv2's fixed-point one (parity unpinned:
 * cv2 is not available to pin it; deviation bounded by cv2's 1/32-pixel / 8-bit quantisation). */
I2R_API int i2r_crop_affine(const unsigned char* img, int32_t ih, int32_t iw, int32_t row_bytes, int32_t swap_rb, const float* inv_trans,
                    const float* mean, const float* inv_std, float* out, int32_t n, int32_t oh, int32_t ow, void* stream);

/* i2r_box_mask -- get_position(shape, box, 'single') + rotate_bound(., 0) + cv2.resize(., IMAGE_SIZE) + ToTensor
 * (JointsDataset.py:165-177,323-331): the filled inclusive rectangle boxes[p] = (int(x), int(y), int(x+w), int(y+h)) at image
 * resolution, resized bilinearly (half-pixel centres) to [n, 1, oh, ow], values in [0, 1].  Same parity note as above. */
I2R_API int i2r_box_mask(const int32_t* boxes, int32_t ih, int32_t iw, float* out, int32_t n, int32_t oh, int32_t ow, void* stream);

/* i2r_crop_affine_cv2 / i2r_box_mask_cv2 -- the same two steps in cv2's OWN arithmetic, restated from OpenCV's published algorithm
 * (imgwarp.cpp warpAffine -> remap with the 1/32-pixel fixed-point bilinear table, 15-bit weights, 8-bit result; resize.cpp 8-bit
 * linear resize with 11-bit coefficients; rotate_bound(mask, 0)'s half-pixel shift along odd image dimensions, JointsDataset.py:180-202).
 * inv_m: device double [n, 6] = the INVERSE of get_affine_transform(...) computed in double precision the way cv2.warpAffine does.
 * Default of the host side (input.person_inputs); still "parity unpinned": cv2 is absent from the build image, nothing could be
 * compared with a cv2 output (oracle/input_cpu.py carries the same restatement in numpy). */
I2R_API int i2r_crop_affine_cv2(const unsigned char* img, int32_t ih, int32_t iw, int32_t row_bytes, int32_t swap_rb, const double* inv_m,
                        const float* mean, const float* inv_std, float* out, int32_t n, int32_t oh, int32_t ow, void* stream);
I2R_API int i2r_box_mask_cv2(const int32_t* boxes, int32_t ih, int32_t iw, float* out, int32_t n, int32_t oh, int32_t ow, void* stream);

/* i2r_person_inputs_cv2 -- the input side of a whole validate() BATCH in one launch: for every person crop p of every image,
 * i2r_crop_affine_cv2 and i2r_box_mask_cv2 (same arithmetic, bit-identical results) written straight into the collated tensors
 * x_out [n_crops, 3, oh, ow] and mask_out [n_crops, 1, oh, ow] that collater.__call__ would build (lib/dataset/collater.py:14-26).
 * images / crops: DEVICE tables (the caller uploads both in one pinned, stream-ordered copy); crops[p].image indexes `images`;
 * the crops of an image must be consecutive for the model's `length` list to describe them, the kernel itself does not care.
 * The tables are device memory the host entry point cannot read: a crop whose `image` index lies outside [0, n_images) or whose
 * image entry is empty (null pointer, ih / iw <= 0) gets ZEROS in x_out / mask_out instead of an out-of-bounds read (the call
 * still returns I2R_OK); `reserved` fields are ignored.
 * mean / inv_std: HOST float[3] (passed on as kernel arguments). */
typedef struct {
    const unsigned char* img;      /* device uint8 [ih, iw, 3], channel order as cv2.imread delivers it */
    int32_t ih, iw, row_bytes, reserved;
} i2r_image_ref;                   /* 24 bytes */
typedef struct {
    double inv_m[6];               /* the INVERSE of get_affine_transform(center, scale, 0, size) in double, as cv2.warpAffine derives it */
    int32_t box[4];                /* (int(x), int(y), int(x + w), int(y + h)): the inclusive corners cv2.rectangle fills */
    int32_t image;                 /* index into the image table */
    int32_t reserved[3];
} i2r_crop_ref;                    /* 80 bytes */
I2R_API int i2r_person_inputs_cv2(const i2r_image_ref* images, int32_t n_images, const i2r_crop_ref* crops, int32_t n_crops, int32_t swap_rb,
                          const float* mean, const float* inv_std, float* x_out, float* mask_out, int32_t oh, int32_t ow, void* stream);

/* ---- HRFormer-B glue (reference lib/models/hrformer.py) ---------------------------------------------------- */
/* i2r_layernorm -- nn.LayerNorm(c, eps) over the channels of every pixel/token of an NHWC tensor
 * (GeneralTransformerBlock.norm1/norm2, hrformer.py:1198,1235-1237). w, b: [cs] zero-padded. */
I2R_API int i2r_layernorm(const float* in, const float* w, const float* b, float* out, int32_t npix, int32_t c, int32_t cs,
                  float eps, int32_t out_dt /* storage of out: 0 fp32, 1 bf16, 2 f16 */, void* stream);

/* i2r_window_attn -- the softmax(q k^T) v core of InterlacedPoolAttention / MHA_ over 7x7 windows
 * (hrformer.py:1164-1180, 692-935): centre zero-padding to multiples of 7 (:947-956), window gather (:978-987),
 * heads = c / head_dim, q scaled by head_dim^-0.5 (:780), NO relative-position bias (:883-885), de-pad (:958-964).
 * qkv: [n, h, w, 3*hs] holding the q | k | v projections of the LayerNorm-ed tokens (a 1x1 i2r_conv with the stacked
 * q/k/v_proj weights), each part hs = heads*40 wide: head hh's dim d at channel hh*40 + d, the pad channel(s) of a head
 * exactly zero (zero weight rows), q ALREADY scaled by head_dim^-0.5 (folded into q_proj by the host).
 * bias_qkv: [3*hs] = the projections of a zero token in the same layout (what padded tokens contribute).
 * out: [n, h, w, hs] attention output BEFORE out_proj, same head-padded channel order (out_proj gets zero columns there).
 * 36 < head_dim <= 40 (HRFormer-B: 39).  Runs on the fp32 matrix pipe: one workgroup per (crop, window, head). */
I2R_API int i2r_window_attn(const float* qkv, const float* bias_qkv, float* out, int32_t n_img, int32_t h, int32_t w, int32_t c,
                    int32_t hs, int32_t heads, void* stream);

/* i2r_hrt_attn_block -- 16-bit modes only: the attention half of a GeneralTransformerBlock in ONE launch,
 *     out = x + out_proj(window_attention(q|k|v_proj(LayerNorm(x))))      (hrformer.py:1230-1236; same semantics as
 * i2r_layernorm + i2r_conv(q|k|v) + i2r_window_attn + i2r_conv(out_proj, res1 = x)), one workgroup per 7x7 window, on
 * v_mfma_f32_16x16x32_{bf16,f16} with fp32 LayerNorm / softmax / accumulation; x and out are fp32 NHWC [n, h, w, cs] (may not alias).
 * Built for the four HRFormer-B branches: (c, heads, cs) = (78, 2, 80), (156, 4, 160), (312, 8, 320), (624, 16, 624); head_dim 39 padded
 * to 48.  variant: 0 = the library's choice; 1 = one wave per 16-token tile of the window, K / V^T through LDS (rounds 3-4; 78 / 156
 * only); 2 = one wave per HEAD over all 64 token rows, the whole attention of a head in registers, every weight fragment feeding four
 * matrix instructions (round 5; all four widths).  Same operands, same arithmetic, results agree to fp32 summation order.
 * Operand images (16-bit, fragment-packed for the 32-deep MFMA: element e of lane l of a fragment = M[16*rowblk + (l & 15)][32*kstep +
 * 8*(l >> 4) + e], 16 bytes per lane, 1 KB per fragment):
 *   wqkv  [head][q, k, v][cs/32 rounded up k-steps][3 dim blocks][64 lanes][8]: rows = the head's 39 output dims (+ 9 zero rows) of
 *         q_proj / k_proj / v_proj, columns = input channels (zero beyond cs); q rows (and bias) pre-multiplied by head_dim^-0.5 * log2(e);
 *   bqkv  float [head][3][48];
 *   wo    [cs/16 output blocks][heads*48/32 k-steps][64 lanes][8]: rows = out_proj outputs, columns = (head, dim) with zeros beyond 39,
 *         re-ordered inside every 32-column k-step to the kernel's slot order: slot 8g + 4h + r <- column 16 (2 kstep + h) + 4g + r
 *         (g < 4, h < 2, r < 4): the B operand of that GEMM is two 16-dim accumulator fragments packed side by side;
 *   bo float [cs];   ln_w, ln_b float [cs] zero-padded. */
I2R_API int i2r_hrt_attn_block(const float* x, float* out, const float* ln_w, const float* ln_b, const void* wqkv, const float* bqkv,
                       const void* wo, const float* bo, int32_t n_img, int32_t h, int32_t w, int32_t c, int32_t cs, int32_t heads,
                       float eps, int32_t dtype, int32_t variant, void* stream);

/* i2r_hrt_mlp_block -- 16-bit modes only: the MLP half of a GeneralTransformerBlock in ONE launch,
 *     out = x + GELU(BN3(fc2( GELU(BN2(dw3x3( GELU(BN1(fc1( LayerNorm(x) ))) ))) )))      (hrformer.py:1237, MlpDWBN :1094-1119; same
 * semantics as i2r_layernorm + i2r_conv(fc1, GELU) + i2r_dwconv3x3(GELU) + i2r_conv(fc2, GELU, res_post = x)), one workgroup per 8x6
 * pixel tile (+ halo of 1, recomputed) on v_mfma_f32_16x16x32_{bf16,f16}; the 4C-wide hidden tensor only exists in LDS, 16 channels per
 * wave at a time.  x and out: fp32 NHWC [n, h, w, cs], distinct buffers; pad channels of x must be zero (they are everywhere in this library).
 * (c, cs) = (78, 80), (156, 160) or (312, 320); hidden_pad = 4 cs (zero weights / biases beyond 4c).
 * variant: 0 = the library's choice; 1 = every wave accumulates fc2 over its own hidden blocks for all output blocks, partial sums meet
 * once through LDS (rounds 3-4; 78 / 156); 2 = ten waves per tile, the hidden dimension in rounds of ten block pairs whose packed
 * depth-wise outputs meet in LDS, every wave accumulating ITS output blocks (round 5; 156 / 312).  Same operands and arithmetic.
 * w1: 16-bit 32-deep fragments [hidden_pad/16][cs/32 rounded up][64 lanes][8] of the BN-folded fc1 matrix [hidden, c] (fragment layout as
 * in i2r_hrt_attn_block; columns zero beyond cs), b1 float [hidden_pad]; wdw float [9][hidden_pad] tap-major BN-folded depth-wise
 * weights, bdw [hidden_pad]; w2: fragments [cs/16][hidden_pad/32][64][8] of the BN-folded fc2 matrix [c, hidden] with the hidden columns
 * of every 32-column k-step in slot order (slot 8g + 4h + r <- column 16 (2 kstep + h) + 4g + r), b2 float [cs]. */
I2R_API int i2r_hrt_mlp_block(const float* x, float* out, const float* ln_w, const float* ln_b, const void* w1, const float* b1,
                      const float* wdw, const float* bdw, const void* w2, const float* b2, int32_t n_img, int32_t h, int32_t w,
                      int32_t c, int32_t cs, int32_t hidden_pad, float eps, int32_t dtype, int32_t variant, void* stream);

/* i2r_dwconv3x3 -- depth-wise 3x3 conv, pad 1, stride 1|2, + bias (eval BN folded) + activation (0 none, 1 ReLU,
 * 2 GELU): MlpDWBN.dw3x3+norm2+act2 (hrformer.py:1070-1080,1106-1108) and the DW down-sampling hops of the fuse
 * layers (:1651-1704). w: [9][cs] (tap-major), bias [cs]. */
I2R_API int i2r_dwconv3x3(const float* in, const float* w, const float* bias, float* out, int32_t n_img, int32_t in_h, int32_t in_w,
                  int32_t c, int32_t cs, int32_t stride, int32_t act, int32_t dt /* storage of in and out: 0 fp32, 1 bf16, 2 f16 (stride 1) */,
                  void* stream);

/* i2r_upsample_bilinear_add -- out = act(res + F.interpolate(low, scale_factor=scale, mode='bilinear',
 * align_corners=False)): the up-sampling terms of HighResolutionTransformerModule.forward (hrformer.py:1629-1646,
 * 1718-1730). res may alias out. */
I2R_API int i2r_upsample_bilinear_add(const float* low, const float* res, float* out, int32_t n_img, int32_t low_h, int32_t low_w,
                              int32_t scale, int32_t c, int32_t cs, int32_t act, void* stream);
/* the same with up to three up-sampled terms added in ONE pass, in the order given (a->low2 / a->low3 may be null): the sum of
 * HighResolutionTransformerModule.forward over the lower-resolution branches j > i (hrformer.py:1718-1730) without handing the partial
 * sums through memory; bit-identical to successive i2r_upsample_bilinear_add calls.  All terms share n_img, c, cs; the output is
 * low_h * scale x low_w * scale and every scale divides it.  (i2r_up_args: below, with the program runner's structs.) */
struct i2r_up_args;
I2R_API int i2r_upsample_bilinear_add_multi(const struct i2r_up_args* a, void* stream);

/* i2r_fuse_up_add -- out = act(base + up(t1, s1) [+ up(t2, s2)]), up = nearest-neighbour up-sampling by s (a power of two; t_k is
 * [n, h/s_k, w/s_k, cs]).  The closing step of the HRNet fuse sum for the outputs that receive lower-resolution terms
 * (interformer_pureMulti.py:392-410: y_i = ReLU(x_i + sum_j Upsample(BN(conv1x1(x_j))))): the 1x1 convs write their small maps once
 * and this HBM-bound pass adds them, instead of s x s read-modify-write scatters from the conv epilogues.  The sum is evaluated
 * left to right ((base + t1) + t2).  base may alias out; t2 may be null.  dt: storage type of all tensors (0 fp32, 1 bf16, 2 f16). */
I2R_API int i2r_fuse_up_add(const float* base, const float* t1, int32_t s1, const float* t2, int32_t s2, float* out, int32_t n_img,
                    int32_t h, int32_t w, int32_t cs, int32_t act, int32_t dt, void* stream);

/* i2r_conv1x1_pair -- two chained 1x1 convolutions over NHWC rows in one launch (fp32):
 *     y = act_a(W_a x + b_a [+ res])     written out, [n_pix, y_cs]
 *     z = act_b(W_b y + b_b)             optional (cb_out = 0: only y), [n_pix, z_cs]
 * Replaces conv3 + bn3 + residual + ReLU of one Bottleneck together with conv1 + bn1 + ReLU of the next (reference lib/models/hrnet.py
 * Bottleneck.forward as used by layer1, interformer_pureMulti.py:69-107, :462-476 _make_layer): y stays in registers between the two
 * GEMMs instead of being written and read back.  w_a / w_b are fragment-packed ([cout / 16][cin / 16][64 lanes][4], engine.pack_frag)
 * with eval BatchNorm folded, b_* the folded biases.  res (optional) is laid out like y.  k_a in {64, 128}; ca_out % 32 == 0;
 * cb_out in {0, 64}; channel strides are multiples of 4 floats.  mt: 16-pixel tiles per wave (1, 2, 4; 0 = default). */
typedef struct i2r_conv1x1_pair_args {
    const float* x; const float* w_a; const float* b_a; const float* res; float* y;
    const float* w_b; const float* b_b; float* z;
    int32_t n_pix, k_a, ca_out, cb_out, x_cs, y_cs, z_cs, relu_a, relu_b, mt;
} i2r_conv1x1_pair_args;
I2R_API int i2r_conv1x1_pair(const i2r_conv1x1_pair_args* a, void* stream);

/* i2r_conv1x1_lp -- 1x1 convolution (+ folded BN) over a small number of NHWC pixel rows on the 16-bit matrix pipe, operands straight
 * from global memory, K split over the four waves of a workgroup:
 *     out = act(W x + bias [+ res1] [+ res2]) [+ res_post]          act: 0 none, 1 ReLU, 2 exact-erf GELU
 * Replaces the single 1x1 convs of the unfused HRFormer-B transformer blocks -- q|k|v and out projections (hrformer.py:1164-1180),
 * MlpDWBN fc1 / fc2 (:1094-1119) -- and of the fuse layers (:1629-1704) where the pixel count is a few thousand and i2r_conv is bound
 * by its per-chunk staging latency.  w: fragment-packed 16-bit [cout_pad / 16][ceil(cin_pad / 32)][64 lanes][8] (engine.pack_frag32:
 * lane (li, g) of fragment (f, c) holds W[16 f + li][32 c + 8 g .. + 8), columns beyond cin_pad zero); bias
 * fp32 [cout_pad]; x fp32 (in_16 = 0, packed on load) or stored in the operand type; out, res1, res_post fp32 or 16 bit (out_16).
 * cout_pad / 16 must be a multiple of 3, 4, 5 or 6; cin_pad >= 64.  res1 / res_post may alias out (in-place accumulation of a fuse sum).  mt: 16-pixel tiles per workgroup (1, 2; 0 = chosen from the grid size). */
typedef struct i2r_conv1x1_lp_args {
    const void* x; const void* w; const float* bias; const void* res1; const void* res_post; void* out;
    int32_t n_pix, cin_pad, cout_pad, x_cs, out_cs, act, dtype, in_16, out_16, mt;
    const void* res2;   /* ABI 14: second pre-activation residual, out = act((W x + bias + res1) + res2) [+ res_post]: the last 1x1 conv of an
                           HRFormer down path that closes a fuse sum (running sum + the branch's own map, hrformer.py:1716-1731) */
} i2r_conv1x1_lp_args;
I2R_API int i2r_conv1x1_lp(const i2r_conv1x1_lp_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * i2r_encoder_desc -- one DETR-style post-norm encoder layer over variable-length token groups
 * (persons of one image attend to each other; no padding, no mask tensor).
 * Replaces: TransformerEncoderLayer.forward_post (interformer_pureMulti.py:192-213; attention.py:61-82;
 * transpose_h.py:189-210) incl. nn.MultiheadAttention (1 head), both LayerNorms and the FFN, plus the
 * pad / key_padding_mask / get_valid_output dance (interformer_pureMulti.py:706-750, utils.py:24-37).
 *
 *   q = ((src+pos) Wq^T + bq) * d^-0.5 ; k = (src+pos) Wk^T + bk ; v = src Wv^T + bv
 *   a = softmax_over_group(q k^T) v ;  x = LN1(src + a Wo^T + bo) ;  out = LN2(x + W2 relu(W1 x + b1) + b2)
 *
 * Token matrix layout: [n_tok, cs] fp32 (cs = padded d, multiple of 16).  Group g owns the contiguous
 * token rows [grp_off[g], grp_off[g+1]).  `pos` has per-token rows when pos_stride_tok = cs, or is a
 * table indexed by (token % pos_period) when pos_period > 0 (TransPose-H sine table).
 * Matrices are the reference's [out][in] nn.Linear weights, zero-padded to cs / dff_pad and FRAGMENT-PACKED: every
 * 16 x 16 block is stored as the MFMA A-operand image in lane order,
 *     packed[((rb*KC + c)*64 + l)*4 + r] = W[16*rb + (l & 15)][16*c + 4*(l >> 4) + r],   KC = cols / 16,
 * so that one 64-lane 16-byte load is 1 KB contiguous (a row-major matrix read in operand order costs 4x the
 * texture-addresser time on gfx950).  i2r_encoder_kv fills K / V^T for the first layer of a stack (fp32 mode: one
 * fragment-packed image per 16-token tile of each group, tiles numbered group by group: <= n_tok/16 + n_grp tiles of
 * 16*cs floats each in kbuf and in vbuf); i2r_encoder_layer consumes them.
 * ------------------------------------------------------------------------------------------------ */
typedef struct i2r_encoder_desc {
    const float* src;      /* [n_tok, cs] */
    const float* pos;      /* optional */
    float* kbuf;           /* workspace, max((n_tok/16 + n_grp) * 16 * cs, cs * n_tok_pad) floats (fp32 / 16-bit mode images) */
    float* vbuf;           /* workspace, same size; n_tok_pad = roundup(n_tok, 64) + 64 */
    float* out;            /* [n_tok, cs] (may not alias src) */
    const int32_t* grp_off;/* device int32 [n_grp + 1] */
    const float* w_in;     /* [3*cs, cs]  (q rows, k rows, v rows; padded), fragment-packed like all four matrices */
    const float* b_in;     /* [3*cs] */
    const float* w_out;    /* [cs, cs] */
    const float* b_out;    /* [cs] */
    const float* ln1_w; const float* ln1_b;  /* [cs] */
    const float* w1;       /* [dff_pad, cs] */
    const float* b1;       /* [dff_pad] */
    const float* w2;       /* [cs, dff_pad] */
    const float* b2;       /* [cs] */
    const float* ln2_w; const float* ln2_b;  /* [cs] */
    int32_t n_tok, n_grp;
    int32_t d;             /* real model dim (LayerNorm width, softmax scale d^-0.5) */
    int32_t cs;            /* padded dim: 96 or 80 */
    int32_t dff_pad;       /* padded feed-forward dim: 192 */
    int32_t pos_period;    /* 0: pos row = token; >0: pos row = token % pos_period */
    int32_t n_qtiles32;    /* sum over groups of ceil(group_len / 32) (host knows the lengths): work items of the two-tile kernel */
    float ln_eps;
    /* 16-bit MFMA mode (dtype 1 bf16 / 2 f16; cs 96 or 80, any group offsets): q-proj, QK^T, PV, out-proj and FFN run on
     * v_mfma_f32_16x16x32 with fp32 accumulation; kbuf / vbuf then hold 16-bit data, one image per 32-token block of a group,
     * blocks numbered group by group (n_qtiles32 of them; the same byte budget is enough).  The model dim is padded to
     * csp = 96 = three 32-feature MFMA steps for BOTH widths (d = 78: zero weights beyond feature 77, rows stay 80 floats in HBM).
     * w_*_lp: the matrices above zero-padded to csp / dff_pad, as 16-bit [out][in] with the columns of every 32-block permuted to
     * new position 8g + 4*half + r  <-  column 32c + 16*half + 4g + r  (g < 4, half < 2, r < 4), fragment-packed. */
    int32_t dtype;
    int32_t n_qtiles16, n_qtiles64;   /* like n_qtiles32 for 16- and 64-query tiles (n_qtiles16 = number of K / V fragments) */
    const void* w_in_lp; const void* w_out_lp; const void* w1_lp; const void* w2_lp;
    /* fp32 mode: fuse the K / V projection of the NEXT layer into this launch (the layer output is still in registers):
     * next_w_in / next_b_in = the next layer's padded in_proj (its k and v rows are used), written to next_kbuf / next_vbuf
     * (layouts as kbuf / vbuf, distinct buffers).  All null = no fusion (then the next layer needs i2r_encoder_kv). */
    const float* next_w_in; const float* next_b_in; float* next_kbuf; float* next_vbuf;
    /* 16-bit mode: the per-feature fp32 vectors padded to csp = 96 and concatenated:
     * b_in[3*csp] | b_out[csp] | ln1_w | ln1_b | b1[dff_pad] | b2[csp] | ln2_w | ln2_b   (9*csp + dff_pad floats) */
    const float* vec_lp;
    /* fp32 mode, optional: scratch of the "partial key split".  A launch with between one and two 16-query tiles per CU (256 < n_qtiles16
     * < 512) handles just enough tiles with TWO workgroups (each over half of the group's keys) that every CU carries two workgroups
     * (csrc/i2r_encoder.hip).  split_ws: 256 * 2 * 1792 floats; split_cnt: 256 int32 hand-off counters: i2r_encoder_kv (the first launch
     * of every stack) zeroes them and every layer launch leaves them zero, so a forward that follows a faulted one starts clean.  The
     * hand-off itself is write-through (sc1 payload stores, drained, then an agent-scope atomic; sc1 loads on the other side).
     * SINGLE-STREAM CONTRACT: launches that share a scratch pair -- i.e. all launches of one i2r_run_program list / one Program --
     * must be stream-ordered; replaying the same list concurrently on two streams pairs partials of different forwards.
     * Both null = one workgroup per tile. */
    float* split_ws; int32_t* split_cnt;
    /* 16-bit mode, long groups (n_qtiles64 >= 512): sum over groups of ceil(group_len / 192) = workgroups of the kernel whose four waves
     * share the K / V stream through LDS (192 queries each); 0 = use the one-wave-per-64-queries kernel */
    int32_t n_qtiles192;
} i2r_encoder_desc;

I2R_API int i2r_encoder_kv(const i2r_encoder_desc* d, void* stream);
I2R_API int i2r_encoder_layer(const i2r_encoder_desc* d, void* stream);

/* i2r_mh_attention -- softmax(q k^T) v of nn.MultiheadAttention with ANY head count over the same variable-length token groups
 * (MODEL.N_HEAD > 1: interformer_pureMulti.py:461,473, transpose_h.py:457,467, interformer_2stage.py:221,233, attention.py:1039; the yacs
 * default is 8, lib/config/default.py:63) -- the attention core of the general encoder layer the host composes for configs the fused
 * single-head post-norm kernels above do not cover (N_HEAD > 1, NORMALIZE_BEFORE with MODEL.NAME interformer: forward_pre,
 * attention.py:84-103): q|k and v come from 1x1 i2r_conv launches over the token rows, out-proj / FFN / LayerNorms likewise.
 *   qk  [n_tok, qk_cs] fp32: q of head h at channels [h*hp, h*hp + hp), k of head h at k_off + the same; q ALREADY scaled by
 *       head_dim^-0.5 (folded into the q rows by the host); hp = head_dim rounded up to a multiple of 16 (<= 256), pad dims exactly 0
 *   v   [n_tok, v_cs], out [n_tok, out_cs]: head h at channels [h*hp, h*hp + hp); out's channels behind heads*hp are written as zeros
 *   grp_off device int32 [n_grp + 1]: group g owns token rows [grp_off[g], grp_off[g+1]) (keys of other groups are never seen: the
 *       reference's pad + key_padding_mask, interformer_pureMulti.py:706-750)
 *   n_qtiles16 / 32 / 64 = sum over groups of ceil(len / 16 | 32 | 64) (the host knows the lengths; the kernel picks its query tile by hp)
 * fp32 matrix pipe, one wave per (query tile, head), no workspace. */
typedef struct i2r_mh_attn_args {
    const float* qk; const float* v; float* out; const int32_t* grp_off;
    int32_t n_grp, heads, hp, k_off, qk_cs, v_cs, out_cs, n_qtiles16, n_qtiles32, n_qtiles64;
    const int32_t* key_len;   /* optional device int32 [n_grp]: group g attends to its FIRST key_len[g] rows only; all its rows stay queries
                                 (padded persons under a key_padding_mask, ATTENTION_TYPE window).  Precondition 1 <= key_len[g] <= length of
                                 group g; the kernel clamps to that range (a device array cannot be validated by the host entry point) */
} i2r_mh_attn_args;
I2R_API int i2r_mh_attention(const i2r_mh_attn_args* a, void* stream);

/* MODEL.ATTENTION_TYPE != 'default' (MODEL.NAME interformer: attention.py:991-1031,1046-1062) -- the inter-human "encoder" is ONE
 * GeneralTransformerBlock: a multi-head attention (q / k / v / out projections with bias, MHA_ :494-835; the relative position bias is
 * gathered but its addition is commented out, :780-786) over the (person, y, x) tokens of an image INCLUDING the padded persons' rows
 * (zero features, keys masked), no residual / FFN / norm, whose [L, B, C] output is then RE-VIEWED: permute(0, 2, 1).contiguous()
 * .view(B, C, P, H, W) (:1025-1029).  The two helpers below restate the padding and that view; the projections are i2r_conv launches, the
 * attention i2r_mh_attention with key_len.
 * i2r_rows_gather: out crop i = src crop map[i] (device int32 [n_out]), zeros where map[i] < 0 (padding_tensor, interformer.py:230-249).
 * i2r_view_scramble: o = attention output rows [n_images][max_persons * hw][cs]; out crop s (NHWC [hw, cs], c real channels) = element
 *   (b', p') = (person_map[s] / max_persons, person_map[s] % max_persons) of the re-viewed tensor:
 *   out[s][yx][c'] = o[b][l][cc] with f = ((b' c + c') max_persons + p') hw + yx, b = f % n_images, cc = (f / n_images) % c, l = f / (c n_images). */
I2R_API int i2r_rows_gather(const float* src, float* out, const int32_t* map, int32_t n_out, int32_t floats_per_crop, void* stream);
I2R_API int i2r_view_scramble(const float* o, float* out, const int32_t* person_map, int32_t n_out, int32_t n_images, int32_t max_persons, int32_t c,
                              int32_t cs, int32_t hw, void* stream);
typedef struct i2r_gather_args { const float* src; float* out; const int32_t* map; int32_t n_out, floats_per_crop; } i2r_gather_args;
typedef struct i2r_scramble_args { const float* o; float* out; const int32_t* person_map; int32_t n_out, n_images, max_persons, c, cs, hw; } i2r_scramble_args;

/* ------------------------------------------------------------------------------------------------
 * Program runner: replay a pre-built list of launches from one C call (no per-op host overhead, and
 * capturable into a hipGraph by the caller).  Streams: ops carry a lane id 0..3; lane 0 is `stream`,
 * other lanes are forked/joined with events by I2R_OP_FORK / I2R_OP_JOIN (op.lane = mask of the lanes 1..3 involved).
 * I2R_OP_XSYNC (op.lane = mask of lanes, bit 0 = lane 0): every lane of the mask continues only after everything issued so far on
 * every OTHER lane of the mask (one event per lane, all-to-all waits) -- the barrier between the branch blocks and the fuse
 * layers of an HRFormer module when branch i and fuse output i both live on lane i, so that no lane idles behind lane 0.
 * I2R_OP_RECORD / I2R_OP_WAIT (round 6; op.lane = lane | slot << 8, slot 0..7): RECORD puts event 8 + slot on the lane's stream behind
 * everything issued on it so far; WAIT makes the lane's stream wait for the LAST record of that slot (nothing if it was recorded on the
 * same stream).  Point-to-point, so a lane waits only for what its next launch reads, in the order the producers finish: on MI355X
 * a cross-stream wait costs the waiter ~10 us after the producer ends and the all-to-all form ~20 us for three lanes
 * (tools/probe/xstream_latency*.hip) -- the HRFormer fuse layers now start the terms of the early lanes under the last lane's blocks.
 * A program with these ops needs an `events` array of 16 entries; RECORD's op.lane also carries the consumer lanes (mask << 16).
 * DEVICE-SIDE FORM: behind an I2R_OP_LANE_FLAGS op (op.args = device int32[64], zeroed once by the caller; NULL = stay with events)
 * FORK / JOIN / RECORD / WAIT become one-wave kernels -- the producer's stream sets flags behind its work, the consumer's stream spins
 * on its flag, clears it and ends (~6 us instead of ~20 us per cross-lane hop).  The caller hands the buffer over ONLY when every lane
 * stream is its own hardware queue (a spinning kernel in front of the kernel that signals it would never end; engine.lane_streams
 * probes this) and never replays two programs that share a buffer at the same time; a wait gives up after 50 ms and sets entry 63 of
 * the buffer instead of hanging the GPU.  XSYNC keeps the event form.
 * ------------------------------------------------------------------------------------------------ */
enum {
    I2R_OP_CONV = 1, I2R_OP_STEM = 2, I2R_OP_MAXPOOL = 3, I2R_OP_HEAD = 4,
    I2R_OP_ENC_KV = 5, I2R_OP_ENC_LAYER = 6, I2R_OP_FORK = 7, I2R_OP_JOIN = 8, I2R_OP_CONV_GROUP = 9,
    I2R_OP_LAYERNORM = 10, I2R_OP_WINATTN = 11, I2R_OP_DWCONV = 12, I2R_OP_UPSAMPLE = 13, I2R_OP_CONV_CHAIN = 14,
    I2R_OP_PE_RES_STEM = 15, I2R_OP_HRT_ATTN = 16, I2R_OP_HRT_MLP = 17, I2R_OP_XSYNC = 18, I2R_OP_FUSE_UP = 19,
    I2R_OP_CONV1X1_PAIR = 20, I2R_OP_CONV1X1_LP = 21, I2R_OP_MH_ATTN = 22, I2R_OP_PE_CAT_VEC = 23, I2R_OP_ROWS_GATHER = 24, I2R_OP_VIEW_SCRAMBLE = 25,
    I2R_OP_RECORD = 26, I2R_OP_WAIT = 27, I2R_OP_LANE_FLAGS = 28
};

typedef struct i2r_stem_args {
    const float* in; const float* w; const float* bias; float* out;
    int32_t n_img, cin, in_h, in_w, cout, out_cs, n_src, n_valid, out_dt;
} i2r_stem_args;

typedef struct i2r_pe_res_args {
    const float* in; const float* w_pre; const float* w7; const float* bias; float* out;
    int32_t n_img, in_h, in_w, cout, out_cs, n_src, n_valid;
} i2r_pe_res_args;

typedef struct i2r_pool_args {
    const float* in; float* out;
    int32_t n_img, in_h, in_w, c, in_cs, out_cs;
} i2r_pool_args;

typedef struct i2r_head_args {
    const float* in; const float* w; const float* bias; float* out;
    int32_t n_img, h, w_, cin, in_cs, cout;
} i2r_head_args;

typedef struct i2r_ln_args {
    const float* in; const float* w; const float* b; float* out;
    int32_t npix, c, cs; float eps; int32_t out_dt;
} i2r_ln_args;

typedef struct i2r_winattn_args {
    const float* qkv; const float* bias; float* out;
    int32_t n_img, h, w_, c, cs, heads;
} i2r_winattn_args;

typedef struct i2r_hrt_attn_args {
    const float* x; float* out; const float* ln_w; const float* ln_b; const void* wqkv; const float* bqkv; const void* wo; const float* bo;
    int32_t n_img, h, w_, c, cs, heads; float eps; int32_t dtype, variant;
} i2r_hrt_attn_args;

typedef struct i2r_hrt_mlp_args {
    const float* x; float* out; const float* ln_w; const float* ln_b; const void* w1; const float* b1; const float* wdw; const float* bdw;
    const void* w2; const float* b2;
    int32_t n_img, h, w_, c, cs, hidden_pad; float eps; int32_t dtype, variant;
} i2r_hrt_mlp_args;

typedef struct i2r_dw_args {
    const float* in; const float* w; const float* bias; float* out;
    int32_t n_img, in_h, in_w, c, cs, stride, act, dt;
} i2r_dw_args;

typedef struct i2r_up_args {
    const float* low; const float* res; float* out;
    int32_t n_img, low_h, low_w, scale, c, cs, act;
    /* optional further terms (i2r_upsample_bilinear_add_multi): out = act(((res + up(low, scale)) + up(low2, scale2)) + up(low3, scale3)) */
    const float* low2; const float* low3;
    int32_t scale2, scale3;
} i2r_up_args;

typedef struct i2r_fuse_up_args {
    const float* base; const float* t1; const float* t2; float* out;
    int32_t n_img, h, w, cs, s1, s2, act, dt;
} i2r_fuse_up_args;

typedef struct i2r_conv_group_args {
    const i2r_conv_desc* d[I2R_MAX_GROUP];
    const int32_t* block_map;
    int32_t n;
    int32_t map_len;
} i2r_conv_group_args;

typedef struct i2r_op {
    int32_t kind;
    int32_t lane;          /* stream lane 0..3 (FORK/JOIN: bitmask of lanes to fork to / join from) */
    const void* args;      /* host pointer to the matching *_desc / *_args struct */
} i2r_op;

/* streams: array of 4 hipStream_t (lane 0 = the caller's stream); events: array of >= 8 hipEvent_t (16 when the program holds
 * RECORD / WAIT ops) created by the caller with hipEventDisableTiming.  Both may be NULL when every op uses lane 0. */
I2R_API int i2r_run_program(const i2r_op* ops, int32_t n_ops, void* const* streams, void* const* events);
/* The same replay with every LAUNCH timed through caller-owned events (hipEventCreate with timing enabled; entries of sync ops -- FORK /
 * JOIN / XSYNC -- are ignored): the launch of op i goes out through hipExtLaunchKernelGGL with t1[i] BOUND to the dispatch (its completion
 * signal carries the kernel's end time; no extra packet, the replay is not slowed down) and, where t0[i] is non-NULL, t0[i] as a marker in
 * front of it (~5 us of stream time).  hipEventElapsedTime(t0[i], t1[i]) is the kernel's own duration IN SITU -- with the program's other
 * lanes and any sibling program in flight; for a launch without a start marker the start is the completion of whatever it waited for (its
 * predecessor on the stream, or the lanes a sync op named), i.e. the t1 of those ops.  tools/probe/event_timing.hip on MI355X, kernels of
 * 20 / 50 us: start + stop 20.9 / 50.9 us at 26.3 / 56.5 us per launch of stream time; stop only 21.4 / 51.4 us between consecutive stops
 * at 21.5 / 51.5 us per launch (= the untimed rate); plain hipEventRecord pairs 23.7 / 53.6 us at 29.0 / 58.8.  A rocprofv3 kernel trace
 * of the same forward is NOT comparable where streams overlap: its host-side interception serialises the part-batch programs
 * (DESIGN.md section 5).  This is the measurement hook of bench.py (per-kernel durations from HIP events on the stream the
 * kernel is launched on); the reference has no counterpart (tools/compute_flops.py:21-32 times whole forwards).
 * t0, t1: n_ops entries each; t1[i] NULL = op i is not timed. */
I2R_API int i2r_run_program_timed(const i2r_op* ops, int32_t n_ops, void* const* streams, void* const* events, void* const* t0,
                                  void* const* t1);

I2R_API int i2r_abi_version(void);
I2R_API const char* i2r_last_error(void);
/* device sanity: returns 0 when device `dev` is gfx950, fills cu_count / lds_bytes if non-NULL */
I2R_API int i2r_device_check(int32_t dev, int32_t* cu_count, int32_t* lds_bytes);

#ifdef __cplusplus
}
#endif
#endif /* I2R_HIP_H */
