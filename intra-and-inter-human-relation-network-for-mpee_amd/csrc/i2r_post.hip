// The two steps that follow the forward in the reference's validate() loop (NEXT #1/#2), on device:
//   i2r_flip_merge  flip-test merge: (y + flip_back(y_flipped)) * 0.5        lib/core/function.py:142-162, utils/transforms.py:16-30
//   i2r_decode      heatmap -> keypoints (DarkPose decode)                   lib/core/inference.py:20-112, utils/transforms.py:50-101
// Both are tiny, HBM/LDS-bound kernels; they remove the D2H copy of every heatmap and a Python double loop over S x J.
#include "i2r_common.h"

namespace {

__global__ __launch_bounds__(256) void flip_merge_k(const float* __restrict__ y, const float* __restrict__ yf,
                                                    const int* __restrict__ jmap, float* __restrict__ out, int n, int J, int h,
                                                    int w) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)n * J * h * w) return;
    const int x = (int)(gid % w);
    const long long r = gid / w;
    const int yy = (int)(r % h);
    const int j = (int)((r / h) % J);
    const int s = (int)(r / ((long long)h * J));
    const float b = yf[(((size_t)s * J + jmap[j]) * h + yy) * w + (w - 1 - x)];
    out[gid] = (y[gid] + b) * 0.5f;
}

struct ArgMax {
    float v;
    int i;
};
__device__ __forceinline__ ArgMax amax2(ArgMax a, ArgMax b) {  // np.argmax: first occurrence wins ties
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

// one workgroup per (crop, joint). LDS: A[h*w] (heatmap, later the blurred map), B[h*w] (row-filtered)
__global__ __launch_bounds__(256) void decode_k(const float* __restrict__ hm, const float* __restrict__ center,
                                                const float* __restrict__ scale, float* __restrict__ preds, float* __restrict__ maxvals,
                                                int J, int h, int w, int ksize, int transform_back) {
    extern __shared__ float sm[];
    float* A = sm;
    float* B = sm + h * w;
    __shared__ ArgMax red[4];
    __shared__ float redf[4];
    __shared__ float gk[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sj = blockIdx.x;
    const int hw = h * w;
    const float* src = hm + (size_t)sj * hw;
    const int border = (ksize - 1) / 2;
    if (tid < ksize) {  // cv2.getGaussianKernel(ksize, sigma<=0): sigma = 0.3*((ksize-1)*0.5 - 1) + 0.8, normalised to sum 1
        const double sigma = 0.3 * ((ksize - 1) * 0.5 - 1.0) + 0.8;
        double sum = 0.0;
        for (int k = 0; k < ksize; ++k) sum += exp(-0.5 * (k - border) * (k - border) / (sigma * sigma));
        gk[tid] = (float)(exp(-0.5 * (tid - border) * (tid - border) / (sigma * sigma)) / sum);
    }
    ArgMax am = {-__builtin_inff(), 0x7fffffff};
    for (int i = tid; i < hw; i += 256) {
        const float v = src[i];
        A[i] = v;
        am = amax2(am, ArgMax{v, i});
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = amax2(am, ArgMax{__shfl_xor(am.v, o), __shfl_xor(am.i, o)});
    if (lane == 0) red[wave] = am;
    __syncthreads();
    am = amax2(amax2(red[0], red[1]), amax2(red[2], red[3]));
    const float origin_max = am.v;
    // row filter (zero padding == the reference's explicit zero border of width (ksize-1)/2)
    for (int i = tid; i < hw; i += 256) {
        const int y = i / w, x = i - y * w;
        float acc = 0.f;
        for (int k = 0; k < ksize; ++k) {
            const int xx = x + k - border;
            if (xx >= 0 && xx < w) acc = fmaf(gk[k], A[y * w + xx], acc);
        }
        B[i] = acc;
    }
    __syncthreads();
    float bm = -__builtin_inff();
    for (int i = tid; i < hw; i += 256) {  // column filter
        const int y = i / w, x = i - y * w;
        float acc = 0.f;
        for (int k = 0; k < ksize; ++k) {
            const int yy = y + k - border;
            if (yy >= 0 && yy < h) acc = fmaf(gk[k], B[yy * w + x], acc);
        }
        A[i] = acc;
        bm = fmaxf(bm, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bm = fmaxf(bm, __shfl_xor(bm, o));
    if (lane == 0) redf[wave] = bm;
    __syncthreads();
    if (tid != 0) return;
    bm = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    const float rescale = origin_max / bm;                       // hm *= origin_max / max(blurred)
    auto L = [&](int y, int x) { return logf(fmaxf(A[y * w + x] * rescale, 1e-10f)); };  // log(max(hm, 1e-10))
    float cx = (float)(am.i % w), cy = floorf((float)am.i / (float)w);
    if (!(origin_max > 0.f)) { cx = 0.f; cy = 0.f; }             // pred_mask (inference.py:42-45)
    const int px = (int)cx, py = (int)cy;
    if (1 < px && px < w - 2 && 1 < py && py < h - 2) {         // taylor (inference.py:51-70)
        const float dx = 0.5f * (L(py, px + 1) - L(py, px - 1));
        const float dy = 0.5f * (L(py + 1, px) - L(py - 1, px));
        const float dxx = 0.25f * (L(py, px + 2) - 2.f * L(py, px) + L(py, px - 2));
        const float dxy = 0.25f * (L(py + 1, px + 1) - L(py - 1, px + 1) - L(py + 1, px - 1) + L(py - 1, px - 1));
        const float dyy = 0.25f * (L(py + 2, px) - 2.f * L(py, px) + L(py - 2, px));
        const float det = dxx * dyy - dxy * dxy;
        if (det != 0.f) {  // offset = -H^-1 g
            cx += -(dyy * dx - dxy * dy) / det;
            cy += -(-dxy * dx + dxx * dy) / det;
        }
    }
    if (transform_back) {  // inverse of the crop affine with rot = 0: pure scale about the centres (transforms.py:50-90)
        const int s = sj / J;
        const float r = (scale[s * 2] * 200.f - 1.f) / (float)(w - 1);
        cx = center[s * 2] + (cx - 0.5f * (float)(w - 1)) * r;
        cy = center[s * 2 + 1] + (cy - 0.5f * (float)(h - 1)) * r;
    }
    preds[(size_t)sj * 2] = cx;
    preds[(size_t)sj * 2 + 1] = cy;
    maxvals[sj] = origin_max;
}

}  // namespace

extern "C" int i2r_flip_merge(const float* y, const float* y_flipped, const int32_t* joint_map, float* out, int32_t n, int32_t joints,
                              int32_t h, int32_t w, void* stream) {
    I2R_CHECK_ARG(y && y_flipped && joint_map && out, "i2r_flip_merge: null pointer");
    const long long tot = (long long)n * joints * h * w;
    i2r_launch(flip_merge_k, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, y_flipped, joint_map,
                       out, n, joints, h, w);
    I2R_CHECK_LAUNCH("i2r_flip_merge");
    return I2R_OK;
}

extern "C" int i2r_decode(const float* heatmaps, const float* center, const float* scale, float* preds, float* maxvals, int32_t n,
                          int32_t joints, int32_t h, int32_t w, int32_t blur_kernel, int32_t transform_back, void* stream) {
    I2R_CHECK_ARG(heatmaps && preds && maxvals && (!transform_back || (center && scale)), "i2r_decode: null pointer");
    I2R_CHECK_ARG(blur_kernel >= 1 && blur_kernel <= 31 && (blur_kernel & 1), "i2r_decode: blur kernel %d", blur_kernel);
    const size_t lds = (size_t)2 * h * w * sizeof(float);
    I2R_CHECK_ARG(lds <= 150 * 1024 && w > 1, "i2r_decode: heatmap %dx%d too large", h, w);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decode_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    i2r_launch(decode_k, dim3((unsigned)(n * joints)), dim3(256), lds, (hipStream_t)stream, heatmaps, center, scale, preds,
                       maxvals, joints, h, w, blur_kernel, transform_back);
    I2R_CHECK_LAUNCH("i2r_decode");
    return I2R_OK;
}
