// Input side of the path: what JointsDataset.__getitem__ does per person with cv2 on the CPU
// (lib/dataset/JointsDataset.py:296-333) -- affine crop of the image to the network input + ToTensor + Normalize, and the
// person's bounding-box mask rasterised at image resolution and resized to the input size.  Byte-in, fp32-out, HBM-bound:
// one thread per output pixel, coalesced NCHW stores, source image rows stay in L2 (a 1-2 MB image feeds all its persons).
//
// NOT bit-identical to cv2 (absent from this image, so the step is "parity unpinned"): cv2.warpAffine / cv2.resize run a
// fixed-point bilinear (coordinates on a 1/32-pixel grid, 8-bit result); this kernel interpolates in fp32 and keeps the result
// unrounded.  The deviation is bounded by cv2's own quantisation: 1/64 pixel in the sample position plus half an intensity level.
#include "i2r_common.h"

namespace {

// out[p][c][y][x] = (bilinear(img, M_p (x, y, 1)) / 255 - mean[c]) * inv_std[c];  taps outside the image read 0 (BORDER_CONSTANT)
__global__ __launch_bounds__(256) void crop_affine_k(const unsigned char* __restrict__ img, int ih, int iw, int row_bytes, int swap_rb,
                                                     const float* __restrict__ inv_trans, const float* __restrict__ mean,
                                                     const float* __restrict__ inv_std, float* __restrict__ out, int n, int oh, int ow) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)n * oh * ow) return;
    const int x = (int)(gid % ow);
    const int y = (int)((gid / ow) % oh);
    const int p = (int)(gid / ((long long)ow * oh));
    const float* m = inv_trans + p * 6;
    const float sx = m[0] * x + m[1] * y + m[2];
    const float sy = m[3] * x + m[4] * y + m[5];
    const float fx0 = floorf(sx), fy0 = floorf(sy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float ax = sx - fx0, ay = sy - fy0;
    const float w00 = (1.f - ax) * (1.f - ay), w01 = ax * (1.f - ay), w10 = (1.f - ax) * ay, w11 = ax * ay;
    const bool vx0 = x0 >= 0 && x0 < iw, vx1 = x0 + 1 >= 0 && x0 + 1 < iw;
    const bool vy0 = y0 >= 0 && y0 < ih, vy1 = y0 + 1 >= 0 && y0 + 1 < ih;
    const unsigned char* r0 = img + (size_t)(vy0 ? y0 : 0) * row_bytes;
    const unsigned char* r1 = img + (size_t)(vy1 ? y0 + 1 : 0) * row_bytes;
    const int c0 = (vx0 ? x0 : 0) * 3, c1 = (vx1 ? x0 + 1 : 0) * 3;
    const float m00 = (vx0 && vy0) ? w00 : 0.f, m01 = (vx1 && vy0) ? w01 : 0.f;
    const float m10 = (vx0 && vy1) ? w10 : 0.f, m11 = (vx1 && vy1) ? w11 : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int sc = swap_rb ? 2 - c : c;  // cv2.imread gives BGR; DATASET.COLOR_RGB converts (JointsDataset.py:223-224)
        const float v = m00 * r0[c0 + sc] + m01 * r0[c1 + sc] + m10 * r1[c0 + sc] + m11 * r1[c1 + sc];
        out[(((size_t)p * 3 + c) * oh + y) * ow + x] = (v * (1.f / 255.f) - mean[c]) * inv_std[c];
    }
}

// out[p][0][y][x] = bilinear resize (half-pixel centres, edge-replicating: cv2.resize INTER_LINEAR geometry) of the image-sized
// binary mask that is 1 inside the inclusive integer rectangle box_p = (x0, y0, x1, y1) clipped to the image (cv2.rectangle, filled)
__global__ __launch_bounds__(256) void box_mask_k(const int* __restrict__ boxes, int ih, int iw, float* __restrict__ out, int n, int oh,
                                                  int ow) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)n * oh * ow) return;
    const int x = (int)(gid % ow);
    const int y = (int)((gid / ow) % oh);
    const int p = (int)(gid / ((long long)ow * oh));
    const int bx0 = boxes[p * 4], by0 = boxes[p * 4 + 1], bx1 = boxes[p * 4 + 2], by1 = boxes[p * 4 + 3];
    const float sx = fmaxf(((float)x + 0.5f) * ((float)iw / (float)ow) - 0.5f, 0.f);
    const float sy = fmaxf(((float)y + 0.5f) * ((float)ih / (float)oh) - 0.5f, 0.f);
    const int x0 = min((int)sx, iw - 1), y0 = min((int)sy, ih - 1);
    const int x1 = min(x0 + 1, iw - 1), y1 = min(y0 + 1, ih - 1);
    const float ax = x0 == iw - 1 ? 0.f : sx - (float)x0, ay = y0 == ih - 1 ? 0.f : sy - (float)y0;
    const float ix0 = (x0 >= bx0 && x0 <= bx1) ? 1.f : 0.f, ix1 = (x1 >= bx0 && x1 <= bx1) ? 1.f : 0.f;
    const float iy0 = (y0 >= by0 && y0 <= by1) ? 1.f : 0.f, iy1 = (y1 >= by0 && y1 <= by1) ? 1.f : 0.f;
    out[((size_t)p * oh + y) * ow + x] = ((1.f - ax) * ix0 + ax * ix1) * ((1.f - ay) * iy0 + ay * iy1);
}

// ---- cv2's own arithmetic (OpenCV imgwarp.cpp / resize.cpp, restated in oracle/input_cpu.py): fixed-point bilinear ----
// weights of initInterTab2D(INTER_LINEAR, fixpt) for the 1/32-pixel fractions (fx, fy): taps (0,0), (0,1), (1,0), (1,1), sum 1 << 15
__device__ __forceinline__ void cv2_tab(int fx, int fy, int (&w)[4]) {
    w[0] = min((32 - fx) * (32 - fy) * 32, 32767);
    w[1] = fx * (32 - fy) * 32;
    w[2] = (32 - fx) * fy * 32;
    w[3] = fx * fy * 32 + ((fx | fy) == 0 ? 1 : 0);
}

// cv2.warpAffine(img, trans, (ow, oh), INTER_LINEAR), BORDER_CONSTANT 0, then ToTensor + Normalize.  m: the INVERSE map in double, as
// cv2 derives it from `trans`; coordinates on the 1/32 grid: X = (rint((m1 y + m2) 1024) + 16 + rint(m0 x 1024)) >> 5.
struct Norm3 { float mean[3], inv_std[3]; };
__device__ __forceinline__ void crop_pixel_cv2(const unsigned char* __restrict__ img, int ih, int iw, int row_bytes, int swap_rb,
                                               const double* __restrict__ m, const Norm3& nm, float* __restrict__ out_p, int x, int y, int oh, int ow) {
    const long long X = (llrint((m[1] * y + m[2]) * 1024.0) + 16 + llrint(m[0] * x * 1024.0)) >> 5;
    const long long Y = (llrint((m[4] * y + m[5]) * 1024.0) + 16 + llrint(m[3] * x * 1024.0)) >> 5;
    const long long sxl = X >> 5, syl = Y >> 5;
    int w[4];
    cv2_tab((int)(X & 31), (int)(Y & 31), w);
    int acc[3] = {1 << 14, 1 << 14, 1 << 14};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long xi = sxl + (k & 1), yi = syl + (k >> 1);
        if (xi < 0 || xi >= iw || yi < 0 || yi >= ih) continue;
        const unsigned char* px = img + (size_t)yi * row_bytes + (size_t)xi * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += w[k] * px[swap_rb ? 2 - c : c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int v = min(max(acc[c] >> 15, 0), 255);
        out_p[((size_t)c * oh + y) * ow + x] = ((float)v * (1.f / 255.f) - nm.mean[c]) * nm.inv_std[c];
    }
}
__global__ __launch_bounds__(256) void crop_affine_cv2_k(const unsigned char* __restrict__ img, int ih, int iw, int row_bytes, int swap_rb,
                                                         const double* __restrict__ inv_m, const float* __restrict__ mean,
                                                         const float* __restrict__ inv_std, float* __restrict__ out, int n, int oh, int ow) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)n * oh * ow) return;
    const int x = (int)(gid % ow);
    const int y = (int)((gid / ow) % oh);
    const int p = (int)(gid / ((long long)ow * oh));
    const Norm3 nm = {{mean[0], mean[1], mean[2]}, {inv_std[0], inv_std[1], inv_std[2]}};
    crop_pixel_cv2(img, ih, iw, row_bytes, swap_rb, inv_m + p * 6, nm, out + (size_t)p * 3 * oh * ow, x, y, oh, ow);
}

// get_position + rotate_bound(., 0) + cv2.resize(., (ow, oh)) + ToTensor, all in cv2's 8-bit arithmetic: the filled rectangle (255) is
// first shifted by half a pixel along every odd image dimension (rotate_bound's nW / 2 - w // 2, a fixed-point warp), then resized
// with 11-bit coefficients: ((b0 (H0 >> 4)) >> 16) + ((b1 (H1 >> 4)) >> 16) + 2 >> 2.
__device__ __forceinline__ float mask_pixel_cv2(const int* __restrict__ box, int ih, int iw, int x, int y, int oh, int ow) {
    const int bx0 = max(box[0], 0), by0 = max(box[1], 0), bx1 = min(box[2], iw - 1), by1 = min(box[3], ih - 1);
    const int ox = iw & 1, oy = ih & 1;  // odd dimension: source coordinate = pixel - 1/2, i.e. taps (pixel - 1, pixel) at fraction 16/32
    int wt[4];
    cv2_tab(16 * ox, 16 * oy, wt);
    auto shifted = [&](int px, int py) -> int {  // rotate_bound(mask, 0) at integer pixel (px, py)
        int acc = 1 << 14;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xi = px - ox + (k & 1), yi = py - oy + (k >> 1);
            if (xi >= bx0 && xi <= bx1 && yi >= by0 && yi <= by1) acc += wt[k] * 255;  // (the rectangle lies inside the image: border taps are 0)
        }
        return acc >> 15;
    };
    auto coef = [](int d, int n_out, int n_in, int& s0, int& s1, int& c0, int& c1, bool zero_at_ends) {
        double f = ((double)d + 0.5) * ((double)n_in / (double)n_out) - 0.5;
        const int s = (int)floor(f);
        const float fr = (float)(f - (double)s);
        c0 = (int)rintf((1.f - fr) * 2048.f);
        c1 = (int)rintf(fr * 2048.f);
        if (zero_at_ends && (s < 0 || s >= n_in - 1)) { c0 = 2048; c1 = 0; }
        s0 = min(max(s, 0), n_in - 1);
        s1 = min(max(s + 1, 0), n_in - 1);
    };
    int x0, x1, a0, a1, y0, y1, b0, b1;
    coef(x, ow, iw, x0, x1, a0, a1, true);
    coef(y, oh, ih, y0, y1, b0, b1, false);
    const int h0 = shifted(x0, y0) * a0 + shifted(x1, y0) * a1;
    const int h1 = shifted(x0, y1) * a0 + shifted(x1, y1) * a1;
    const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
    return (float)min(max(v, 0), 255) * (1.f / 255.f);
}
__global__ __launch_bounds__(256) void box_mask_cv2_k(const int* __restrict__ boxes, int ih, int iw, float* __restrict__ out, int n, int oh,
                                                      int ow) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)n * oh * ow) return;
    const int x = (int)(gid % ow);
    const int y = (int)((gid / ow) % oh);
    const int p = (int)(gid / ((long long)ow * oh));
    out[((size_t)p * oh + y) * ow + x] = mask_pixel_cv2(boxes + p * 4, ih, iw, x, y, oh, ow);
}

// the whole batch of a validate() step in ONE launch: every person crop of every image + its box mask, written straight into the
// collated [S, 3, oh, ow] / [S, 1, oh, ow] tensors (collater.py:14-26); images and crops are described by two device tables
__global__ __launch_bounds__(256) void person_inputs_cv2_k(const i2r_image_ref* __restrict__ images, const i2r_crop_ref* __restrict__ crops,
                                                           int n_images, int swap_rb, Norm3 nm, float* __restrict__ x_out,
                                                           float* __restrict__ m_out, int n, int oh, int ow) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)n * oh * ow) return;
    const int x = (int)(gid % ow);
    const int y = (int)((gid / ow) % oh);
    const int p = (int)(gid / ((long long)ow * oh));
    const i2r_crop_ref* cr = crops + p;
    // the tables live on the device: the host entry point cannot validate them, so a crop whose image index lies outside the image
    // table (or whose image entry is empty) writes zeros instead of reading out of bounds
    if ((unsigned)cr->image >= (unsigned)n_images || images[cr->image].img == nullptr || images[cr->image].ih <= 0 || images[cr->image].iw <= 0) {
        for (int c = 0; c < 3; ++c) x_out[(((size_t)p * 3 + c) * oh + y) * ow + x] = 0.f;
        m_out[((size_t)p * oh + y) * ow + x] = 0.f;
        return;
    }
    const i2r_image_ref im = images[cr->image];
    crop_pixel_cv2(im.img, im.ih, im.iw, im.row_bytes, swap_rb, cr->inv_m, nm, x_out + (size_t)p * 3 * oh * ow, x, y, oh, ow);
    m_out[((size_t)p * oh + y) * ow + x] = mask_pixel_cv2(cr->box, im.ih, im.iw, x, y, oh, ow);
}

}  // namespace

extern "C" int i2r_person_inputs_cv2(const i2r_image_ref* images, int32_t n_images, const i2r_crop_ref* crops, int32_t n_crops, int32_t swap_rb,
                                     const float* mean, const float* inv_std, float* x_out, float* mask_out, int32_t oh, int32_t ow, void* stream) {
    I2R_CHECK_ARG(images && crops && mean && inv_std && x_out && mask_out, "i2r_person_inputs_cv2: null pointer");
    I2R_CHECK_ARG(n_images > 0 && n_crops > 0 && oh > 0 && ow > 0, "i2r_person_inputs_cv2: sizes");
    Norm3 nm;
    for (int c = 0; c < 3; ++c) { nm.mean[c] = mean[c]; nm.inv_std[c] = inv_std[c]; }  // (HOST arrays: they travel as kernel arguments)
    const long long nthr = (long long)n_crops * oh * ow;
    i2r_launch(person_inputs_cv2_k, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, images, crops, n_images, swap_rb,
                       nm, x_out, mask_out, n_crops, oh, ow);
    I2R_CHECK_LAUNCH("i2r_person_inputs_cv2");
    return I2R_OK;
}

extern "C" int i2r_crop_affine_cv2(const unsigned char* img, int32_t ih, int32_t iw, int32_t row_bytes, int32_t swap_rb, const double* inv_m,
                                   const float* mean, const float* inv_std, float* out, int32_t n, int32_t oh, int32_t ow, void* stream) {
    I2R_CHECK_ARG(img && inv_m && mean && inv_std && out, "i2r_crop_affine_cv2: null pointer");
    I2R_CHECK_ARG(ih > 0 && iw > 0 && row_bytes >= 3 * iw && n > 0 && oh > 0 && ow > 0, "i2r_crop_affine_cv2: sizes");
    const long long nthr = (long long)n * oh * ow;
    i2r_launch(crop_affine_cv2_k, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, ih, iw, row_bytes,
                       swap_rb, inv_m, mean, inv_std, out, n, oh, ow);
    I2R_CHECK_LAUNCH("i2r_crop_affine_cv2");
    return I2R_OK;
}

extern "C" int i2r_box_mask_cv2(const int32_t* boxes, int32_t ih, int32_t iw, float* out, int32_t n, int32_t oh, int32_t ow, void* stream) {
    I2R_CHECK_ARG(boxes && out, "i2r_box_mask_cv2: null pointer");
    I2R_CHECK_ARG(ih > 0 && iw > 0 && n > 0 && oh > 0 && ow > 0, "i2r_box_mask_cv2: sizes");
    const long long nthr = (long long)n * oh * ow;
    i2r_launch(box_mask_cv2_k, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes, ih, iw, out, n, oh, ow);
    I2R_CHECK_LAUNCH("i2r_box_mask_cv2");
    return I2R_OK;
}

extern "C" int i2r_crop_affine(const unsigned char* img, int32_t ih, int32_t iw, int32_t row_bytes, int32_t swap_rb,
                               const float* inv_trans, const float* mean, const float* inv_std, float* out, int32_t n, int32_t oh,
                               int32_t ow, void* stream) {
    I2R_CHECK_ARG(img && inv_trans && mean && inv_std && out, "i2r_crop_affine: null pointer");
    I2R_CHECK_ARG(ih > 0 && iw > 0 && row_bytes >= 3 * iw && n > 0 && oh > 0 && ow > 0, "i2r_crop_affine: sizes");
    const long long nthr = (long long)n * oh * ow;
    i2r_launch(crop_affine_k, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, ih, iw, row_bytes,
                       swap_rb, inv_trans, mean, inv_std, out, n, oh, ow);
    I2R_CHECK_LAUNCH("i2r_crop_affine");
    return I2R_OK;
}

extern "C" int i2r_box_mask(const int32_t* boxes, int32_t ih, int32_t iw, float* out, int32_t n, int32_t oh, int32_t ow, void* stream) {
    I2R_CHECK_ARG(boxes && out, "i2r_box_mask: null pointer");
    I2R_CHECK_ARG(ih > 0 && iw > 0 && n > 0 && oh > 0 && ow > 0, "i2r_box_mask: sizes");
    const long long nthr = (long long)n * oh * ow;
    i2r_launch(box_mask_k, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes, ih, iw, out, n, oh, ow);
    I2R_CHECK_LAUNCH("i2r_box_mask");
    return I2R_OK;
}
