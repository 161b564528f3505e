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

}  // namespace

extern "C" int i2r_crop_affine(const unsigned char* img, int32_t ih, int32_t iw, int32_t row_bytes, int32_t swap_rb,
                               const float* inv_trans, const float* mean, const float* inv_std, float* out, int32_t n, int32_t oh,
                               int32_t ow, void* stream) {
    I2R_CHECK_ARG(img && inv_trans && mean && inv_std && out, "i2r_crop_affine: null pointer");
    I2R_CHECK_ARG(ih > 0 && iw > 0 && row_bytes >= 3 * iw && n > 0 && oh > 0 && ow > 0, "i2r_crop_affine: sizes");
    const long long nthr = (long long)n * oh * ow;
    hipLaunchKernelGGL(crop_affine_k, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, ih, iw, row_bytes,
                       swap_rb, inv_trans, mean, inv_std, out, n, oh, ow);
    I2R_CHECK_LAUNCH("i2r_crop_affine");
    return I2R_OK;
}

extern "C" int i2r_box_mask(const int32_t* boxes, int32_t ih, int32_t iw, float* out, int32_t n, int32_t oh, int32_t ow, void* stream) {
    I2R_CHECK_ARG(boxes && out, "i2r_box_mask: null pointer");
    I2R_CHECK_ARG(ih > 0 && iw > 0 && n > 0 && oh > 0 && ow > 0, "i2r_box_mask: sizes");
    const long long nthr = (long long)n * oh * ow;
    hipLaunchKernelGGL(box_mask_k, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes, ih, iw, out, n, oh, ow);
    I2R_CHECK_LAUNCH("i2r_box_mask");
    return I2R_OK;
}
