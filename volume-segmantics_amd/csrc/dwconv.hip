// Operators of smp's DeepLabV3+ decoder that are not dense convolutions (segmentation-models-pytorch 0.2.1,
// decoders/deeplabv3/decoder.py), NHWC tensors, HBM-bound sweeps (gfx950):
//   * depthwise 3x3 convolution with dilation d, padding d (SeparableConv2d's first half: nn.Conv2d(c, c, 3, padding=d, dilation=d,
//     groups=c, bias=False)) - forward, data gradient (the same sweep with the taps flipped), weight gradient (two fixed-order stages)
//   * per-sample spatial sums / means (nn.AdaptiveAvgPool2d(1) of ASPPPooling) and the broadcast of a [n][c] row over the map
//     (F.interpolate of a 1x1 map; both are each other's gradients)
//   * element-wise nn.Dropout(p) with the mask recomputed from a counter-based hash (nothing stored: the backward pass applies the
//     same call to the gradient)
#include <algorithm>

#include "common.h"

namespace {

constexpr int kVec = 8;
constexpr int kDwBlocks = 1024;   // partial rows of the depthwise weight gradient (four workgroups per CU: the row loop is latency-bound)

inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// y[n][h][w][c] = sum_t x[n][h + (kh-1) d][w + (kw-1) d][c] * wgt[c][t'], t' = t (forward) or 8 - t (data gradient).
// A lane keeps ONE channel vector (8 channels: its 72 taps live in registers, loaded once) and walks over pixels; blockIdx.y = slab
// of up to 256 channel vectors.  (The first version re-read the taps per output vector - 72 scalar loads 36 bytes apart - and ran
// at 0.3 TB/s; taps staged in LDS as [tap][c]: 0.95 TB/s.)
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const T* __restrict__ x, const float* __restrict__ wgt, T* __restrict__ y, int n, int h, int w,
                                                      int c, int d, int flip) {
    const int cv_all = c / kVec, v0 = blockIdx.y * 256;
    const int cv = min(256, cv_all - v0), ppb = 256 / cv;          // pixels per workgroup and sweep
    const int cg = v0 + threadIdx.x % cv, pl = threadIdx.x / cv;
    if (pl >= ppb) return;
    float wr[9][kVec];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int k = 0; k < kVec; ++k) wr[tap][k] = wgt[((size_t)cg * kVec + k) * 9 + ((flip & 1) ? 8 - tap : tap)];
    const int64_t pixels = (int64_t)n * h * w;
    for (int64_t p = (int64_t)blockIdx.x * ppb + pl; p < pixels; p += (int64_t)gridDim.x * ppb) {
        const int wo = (int)(p % w), ho = (int)(p / w % h);
        const int64_t b = p / w / h;
        float acc[kVec];
#pragma unroll
        for (int k = 0; k < kVec; ++k) acc[k] = 0.f;
        const T* xb = x + (size_t)b * h * w * c + cg * kVec;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int hi = ho + (tap / 3 - 1) * d, wi = wo + (tap % 3 - 1) * d;
            if (hi < 0 || hi >= h || wi < 0 || wi >= w) continue;
            float v[kVec];
            ld8(xb + ((size_t)hi * w + wi) * c, v);
#pragma unroll
            for (int k = 0; k < kVec; ++k) acc[k] += v[k] * wr[tap][k];
        }
        T* yo = y + (size_t)p * c + cg * kVec;
        if (flip & 2) {       // accumulate onto y (a gradient that already holds other consumers' contributions)
            float old[kVec];
            ld8(yo, old);
#pragma unroll
            for (int k = 0; k < kVec; ++k) acc[k] += old[k];
        }
        st8(yo, acc);
    }
}

// weight gradient, stage 1: partial[blk][c][9] = sum over the block's pixels of dy[p][c] * x[p (+) tap][c]
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy, int n, int h, int w, int c,
                                                                    int d, float* __restrict__ partial) {
    extern __shared__ float red[];                 // [row lane][cv][kVec][9]
    // blockIdx.y = slab of up to 256 channels (the last one may be narrower: 304 = 256 + 48); rl row lanes share a slab's vectors
    const int ctot = c, c0 = blockIdx.y * 256;
    const int cs = min(256, ctot - c0), cv = cs / kVec, rl = 256 / cv;
    const int cg = threadIdx.x % cv, r0 = threadIdx.x / cv;
    float s[kVec][9];
#pragma unroll
    for (int k = 0; k < kVec; ++k)
#pragma unroll
        for (int t = 0; t < 9; ++t) s[k][t] = 0.f;
    const int64_t rows = (int64_t)n * h * w;
    for (int64_t r = (int64_t)blockIdx.x * rl + r0; r0 < rl && r < rows; r += (int64_t)gridDim.x * rl) {
        const int wo = (int)(r % w), ho = (int)(r / w % h);
        const int64_t b = r / w / h;
        float g[kVec];
        ld8(dy + (size_t)r * ctot + c0 + cg * kVec, g);
        const T* xb = x + (size_t)b * h * w * ctot + c0 + cg * kVec;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int hi = ho + (tap / 3 - 1) * d, wi = wo + (tap % 3 - 1) * d;
            if (hi < 0 || hi >= h || wi < 0 || wi >= w) continue;
            float v[kVec];
            ld8(xb + ((size_t)hi * w + wi) * ctot, v);
#pragma unroll
            for (int k = 0; k < kVec; ++k) s[k][tap] += g[k] * v[k];
        }
    }
#pragma unroll
    for (int k = 0; k < kVec; ++k)
#pragma unroll
        for (int t = 0; t < 9; ++t) red[((size_t)threadIdx.x * kVec + k) * 9 + t] = s[k][t];
    __syncthreads();
    for (int o = threadIdx.x; o < cv * kVec * 9; o += 256) {      // o = (channel in slab) * 9 + tap
        const int ch = o / 9, t = o % 9;
        float acc = 0.f;
        for (int r = 0; r < rl; ++r) acc += red[((size_t)(r * cv + ch / kVec) * kVec + ch % kVec) * 9 + t];
        partial[((size_t)blockIdx.x * ctot + c0 + ch) * 9 + t] = acc;
    }
}
__global__ __launch_bounds__(64) void dwconv3x3_wgrad_final_kernel(const float* __restrict__ partial, float* __restrict__ dw, int nblk, int total) {
    const int o = blockIdx.x;                      // channel * 9 + tap
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 64) s += partial[(size_t)b * total + o];
    s = wave_sum(s);
    if (threadIdx.x == 0) dw[o] = s;
}

// y[n][c] = scale * sum over hw of x[n][hw][c]   (one block per sample and 256-channel slab)
template <typename T>
__global__ __launch_bounds__(256) void spatial_sum_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t hw, int c, float scale) {
    __shared__ float red[256][kVec];
    const int cs = c < 256 ? c : 256, cv = cs / kVec, rl = 256 / cv, c0 = blockIdx.y * 256;
    const int cg = threadIdx.x % cv, r0 = threadIdx.x / cv;
    const T* xs = x + (size_t)blockIdx.x * hw * c + c0 + cg * kVec;
    float s[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) s[k] = 0.f;
    int64_t r = r0;
    for (; r + 7 * rl < hw; r += 8 * rl) {          // eight independent loads in flight per lane (one sample per workgroup: latency-bound)
        float v[8][kVec];
#pragma unroll
        for (int u = 0; u < 8; ++u) ld8(xs + (size_t)(r + u * rl) * c, v[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < kVec; ++k) s[k] += v[u][k];
    }
    for (; r < hw; r += rl) {
        float v[kVec];
        ld8(xs + (size_t)r * c, v);
#pragma unroll
        for (int k = 0; k < kVec; ++k) s[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < kVec; ++k) red[threadIdx.x][k] = s[k];
    __syncthreads();
    if (threadIdx.x < cs) {
        const int g = threadIdx.x / kVec, k = threadIdx.x % kVec;
        float t = 0.f;
        for (int r = 0; r < rl; ++r) t += red[r * cv + g][k];
        Elem<T>::st(y + (size_t)blockIdx.x * c + c0 + threadIdx.x, t * scale);
    }
}
// y[n][hw][c] (+)= scale * v[n][c]
template <typename T>
__global__ void broadcast_rows_kernel(const T* __restrict__ v, T* __restrict__ y, int n, int64_t hw, int c, float scale, int accumulate) {
    const int cv = c / kVec;
    const int64_t total = (int64_t)n * hw * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % cv);
        const int b = (int)(i / cv / hw);
        float a[kVec], o[kVec];
        ld8(v + (size_t)b * c + cg * kVec, a);
#pragma unroll
        for (int k = 0; k < kVec; ++k) o[k] = a[k] * scale;
        if (accumulate) {
            float old[kVec];
            ld8(y + i * kVec, old);
#pragma unroll
            for (int k = 0; k < kVec; ++k) o[k] += old[k];
        }
        st8(y + i * kVec, o);
    }
}

// A 3x3 convolution at a LARGE dilation r (padding r: DeepLabV3's dense ASPP rates 12 / 24 / 36 on a 32 x 32 map, where most taps
// look at padding) as one 1x1 convolution over 9 c channels: col[n][i][j][tap][c] = x[n][i + (kh - 1) r][j + (kw - 1) r][c] (zero
// beyond the map).  The weights [cout][tap][c] ARE that 1x1 convolution's [cout][9 c] matrix.  inverse = 1: the adjoint,
// dst[n][i][j][c] (+)= sum over taps of src[n][i - (kh - 1) r][j - (kw - 1) r][tap][c] (fp32 sum of <= 9 terms).
template <typename T>
__global__ void dilated_im2col_kernel(const T* __restrict__ src, T* __restrict__ dst, int n, int h, int w, int c, int r, int inverse, int accumulate) {
    const int cv = c / kVec;
    if (!inverse) {
        const int64_t total = (int64_t)n * h * w * 9 * cv;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            int64_t t = i;
            const int cg = t % cv; t /= cv;
            const int tap = t % 9; t /= 9;
            const int ww = t % w; t /= w;
            const int hh = t % h;
            const int64_t nb = t / h;
            const int sh = hh + (tap / 3 - 1) * r, sw = ww + (tap % 3 - 1) * r;
            float v[kVec];
#pragma unroll
            for (int k = 0; k < kVec; ++k) v[k] = 0.f;
            if (sh >= 0 && sh < h && sw >= 0 && sw < w) ld8(src + ((nb * h + sh) * w + sw) * c + cg * kVec, v);
            st8(dst + i * kVec, v);
        }
    } else {
        const int64_t total = (int64_t)n * h * w * cv;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            int64_t t = i;
            const int cg = t % cv; t /= cv;
            const int ww = t % w; t /= w;
            const int hh = t % h;
            const int64_t nb = t / h;
            float v[kVec];
#pragma unroll
            for (int k = 0; k < kVec; ++k) v[k] = 0.f;
            for (int tap = 0; tap < 9; ++tap) {
                const int sh = hh - (tap / 3 - 1) * r, sw = ww - (tap % 3 - 1) * r;
                if (sh < 0 || sh >= h || sw < 0 || sw >= w) continue;
                float g[kVec];
                ld8(src + (((nb * h + sh) * w + sw) * 9 + tap) * c + cg * kVec, g);
#pragma unroll
                for (int k = 0; k < kVec; ++k) v[k] += g[k];
            }
            if (accumulate) {
                float old[kVec];
                ld8(dst + i * kVec, old);
#pragma unroll
                for (int k = 0; k < kVec; ++k) v[k] += old[k];
            }
            st8(dst + i * kVec, v);
        }
    }
}

__device__ __forceinline__ uint32_t dmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// y[i] = keep(i) ? x[i] / (1 - p) : 0, keep drawn per ELEMENT from (seed, *counter + bias, i)
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t nvec, float p, uint32_t seed, const int64_t* __restrict__ counter,
                               int64_t bias) {
    const uint64_t step = (uint64_t)((counter ? *counter : 0) + bias);
    uint32_t h0 = dmix32(seed ^ 0x2545f491U);
    h0 = dmix32(h0 ^ (uint32_t)step);
    h0 = dmix32(h0 ^ (uint32_t)(step >> 32) ^ 0xc2b2ae35U);
    const float inv = 1.f / (1.f - p);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        float v[kVec];
        ld8(x + i * kVec, v);
#pragma unroll
        for (int k = 0; k < kVec; ++k) {
            const uint64_t e = (uint64_t)i * kVec + k;
            const uint32_t hh = dmix32(dmix32(h0 ^ (uint32_t)e) ^ (uint32_t)(e >> 32));
            const float u = (float)(hh >> 8) * (1.f / 16777216.f);
            v[k] = u < p ? 0.f : v[k] * inv;
        }
        st8(y + i * kVec, v);
    }
}

}  // namespace

#define VS_LAUNCH_T(kernel, grid, lds, s, ...)                                                                          \
    do {                                                                                                                \
        VS_FOR_T(dtype, { hipLaunchKernelGGL((kernel<T>), grid, dim3(256), lds, s, __VA_ARGS__); });                   \
        VS_LAUNCH_CHECK();                                                                                              \
    } while (0)

// nn.Conv2d(c, c, 3, padding=dilation, dilation=dilation, groups=c, bias=False) on x [n][h][w][c]; w: fp32 [c][9] (torch's [c][1][3][3]).
// flip: bit 0 = the data gradient (y = dx for x = dy: taps reversed), bit 1 = add to y instead of overwriting it.
extern "C" int vs_dwconv3x3(int dtype, const void* x, const float* w, void* y, int n, int h, int wd, int c, int dilation, int flip, void* stream) {
    VS_REQUIRE(x && w && y && c > 0 && c % kVec == 0 && dilation >= 1, "dwconv3x3: channels must be a multiple of 8, dilation >= 1");
    const int cv = c / kVec, slabs = (cv + 255) / 256, ppb = 256 / std::min(cv, 256);
    const int64_t gx = std::min<int64_t>(((int64_t)n * h * wd + ppb - 1) / ppb, 2048 / slabs);   // few, long-lived workgroups: the taps are loaded once per lane
    VS_LAUNCH_T(dwconv3x3_kernel, dim3((unsigned)gx, slabs), 0, (hipStream_t)stream, (const T*)x, w, (T*)y, n, h, wd, c, dilation, flip);
    return VS_OK;
}
extern "C" size_t vs_dwconv3x3_wgrad_workspace(int c) { return (size_t)kDwBlocks * c * 9 * sizeof(float); }
// dw [c][9] fp32 = sum over pixels of dy * shifted x
extern "C" int vs_dwconv3x3_wgrad(int dtype, const void* x, const void* dy, float* dw, int n, int h, int wd, int c, int dilation, float* workspace,
                                  size_t workspace_bytes, void* stream) {
    const int cs = c < 256 ? c : 256;
    VS_REQUIRE(x && dy && dw && workspace && c > 0 && c % kVec == 0 && dilation >= 1, "dwconv3x3_wgrad: channels must be a multiple of 8 (got %d)", c);
    VS_REQUIRE(workspace_bytes >= vs_dwconv3x3_wgrad_workspace(c), "dwconv3x3_wgrad: workspace too small");
    const int rl = 256 / (cs / kVec);
    const int64_t rows = (int64_t)n * h * wd;
    const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(kDwBlocks, (rows + rl - 1) / rl));
    const size_t lds = (size_t)256 * kVec * 9 * sizeof(float);          // 72 KB
    hipStream_t s = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void*)dwconv3x3_wgrad_partial_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        VS_CHECK_HIP(hipFuncSetAttribute((const void*)dwconv3x3_wgrad_partial_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    VS_LAUNCH_T(dwconv3x3_wgrad_partial_kernel, dim3(nblk, (c + 255) / 256), lds, s, (const T*)x, (const T*)dy, n, h, wd, c, dilation, workspace);
    hipLaunchKernelGGL(dwconv3x3_wgrad_final_kernel, dim3(c * 9), dim3(64), 0, s, workspace, dw, nblk, c * 9);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

// y [n][c] = scale * sum over the hw positions of x [n][hw][c]  (AdaptiveAvgPool2d(1): scale = 1 / hw; the gradient of a broadcast: 1)
extern "C" int vs_spatial_sum(int dtype, const void* x, void* y, int n, int64_t hw, int c, float scale, void* stream) {
    const int cs = c < 256 ? c : 256;
    VS_REQUIRE(x && y && c > 0 && c % kVec == 0, "spatial_sum: unsupported channel count %d", c);
    if (c % cs || 256 % (cs / kVec)) return vs_sample_rowsum(dtype, x, nullptr, y, n, hw, c, scale, stream);   // any other multiple of 8 (csrc/effnet.hip)
    VS_LAUNCH_T(spatial_sum_kernel, dim3(n, c / cs), 0, (hipStream_t)stream, (const T*)x, (T*)y, hw, c, scale);
    return VS_OK;
}
// y [n][hw][c] (+)= scale * v [n][c]  (F.interpolate of a 1x1 map to any size; the gradient of the average pool with scale = 1 / hw)
extern "C" int vs_broadcast_rows(int dtype, const void* v, void* y, int n, int64_t hw, int c, float scale, int accumulate, void* stream) {
    VS_REQUIRE(v && y && c > 0 && c % kVec == 0, "broadcast_rows: channels must be a multiple of 8");
    VS_LAUNCH_T(broadcast_rows_kernel, dim3(grid_for((int64_t)n * hw * (c / kVec))), 0, (hipStream_t)stream, (const T*)v, (T*)y, n, hw, c, scale, accumulate);
    return VS_OK;
}
// nn.Dropout(p), element-wise: y = x * mask / (1 - p), mask = f(seed, *counter + bias, element index) recomputed on every call
// (the backward pass calls it on the gradient with the same seed / counter); elems must be a multiple of 8
extern "C" int vs_dropout(int dtype, const void* x, void* y, int64_t elems, float p, uint32_t seed, const int64_t* counter, int64_t bias,
                          void* stream) {
    VS_REQUIRE(x && y && elems >= 0 && elems % kVec == 0 && p >= 0.f && p < 1.f, "dropout: bad arguments");
    VS_LAUNCH_T(dropout_kernel, dim3(grid_for(elems / kVec)), 0, (hipStream_t)stream, (const T*)x, (T*)y, elems / kVec, p, seed, counter, bias);
    return VS_OK;
}

// x [n][h][w][c] -> col [n][h][w][9][c] (see dilated_im2col_kernel); inverse = 1: src is the column form (a gradient), dst the map
// (accumulate = 1 adds to it)
extern "C" int vs_dilated_im2col(int dtype, const void* src, void* dst, int n, int h, int w, int c, int r, int inverse, int accumulate, void* stream) {
    VS_REQUIRE(src && dst && n > 0 && h > 0 && w > 0 && c > 0 && c % kVec == 0 && r >= 1, "dilated_im2col: channels must be a multiple of 8, rate >= 1");
    const int64_t total = (int64_t)n * h * w * (c / kVec) * (inverse ? 1 : 9);
    VS_LAUNCH_T(dilated_im2col_kernel, dim3(grid_for(total)), 0, (hipStream_t)stream, (const T*)src, (T*)dst, n, h, w, c, r, inverse, accumulate);
    return VS_OK;
}
