// Streaming operators of smp.FPN's decoder (segmentation-models-pytorch 0.2.1, decoders/fpn/decoder.py) on NHWC tensors (gfx950):
//   * nearest-x2 upsampling + add                      - FPNBlock: F.interpolate(x, 2, "nearest") + skip_conv(skip)
//   * GroupNorm(32, C) (+ ReLU), forward and backward  - Conv3x3GNReLU
//   * bilinear x2 upsampling, align_corners=True       - Conv3x3GNReLU(upsample=True), forward and backward (NHWC), and the
//     same at any integer factor on fp32 NCHW logits   - SegmentationHead(upsampling=4): nn.UpsamplingBilinear2d
//   * Dropout2d (whole channels of a sample)           - FPNDecoder.dropout, mask drawn from a counter-based generator
// All HBM-bound sweeps: 16-byte accesses along the channel axis, fixed-order two-stage reductions (no float atomics).  The
// convolutions between them are conv_igemm / conv_wgrad launches (unet.hip strings them together).
#include <algorithm>

#include "common.h"

namespace {

constexpr int kVec = 8;
constexpr int kGnBlocks = 64;   // partial rows per sample of the GroupNorm reductions

inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// out[n][2i+a][2j+b][c] = x[n][i][j][c] + s[n][2i+a][2j+b][c]
template <typename T>
__global__ void upsample2x_add_kernel(const T* __restrict__ x, const T* __restrict__ s, T* __restrict__ y, int n, int h, int w, int c) {
    const int cv = c / kVec;
    const int64_t total = (int64_t)n * 4 * h * w * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int cg = t % cv; t /= cv;
        const int wo = t % (2 * w); t /= (2 * w);
        const int ho = t % (2 * h);
        const int b = t / (2 * h);
        float a[kVec], v[kVec];
        ld8(x + (((size_t)b * h + (ho >> 1)) * w + (wo >> 1)) * c + cg * kVec, a);
        if (s) {
            ld8(s + i * kVec, v);
#pragma unroll
            for (int k = 0; k < kVec; ++k) v[k] += a[k];
            st8(y + i * kVec, v);
        } else {            // plain nearest-x2 upsampling (no addend)
            st8(y + i * kVec, a);
        }
    }
}

// ---- GroupNorm -------------------------------------------------------------------------------------------------------------
// x: [n][hw][c], groups of cpg = c / G consecutive channels, statistics per (sample, group) over hw * cpg elements.
// Sweep 1: block (blk, n) sums its rows per channel (thread = (row lane, 8-channel vector)), the row lanes meet in LDS, channel
// sums fold into group sums: partial[(n * nblk + blk) * G + g] = {sum, sum of squares}.
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, int64_t hw, int c, int G, float* __restrict__ partial) {
    __shared__ float red[2][256 * kVec / 8][8];   // [which][row lane * cv + cg][k]
    const int cv = c / kVec, rl = 256 / cv, cpg = c / G;
    const int cg = threadIdx.x % cv, r0 = threadIdx.x / cv;
    const int n = blockIdx.y, nblk = gridDim.x;
    const T* xs = x + (size_t)n * hw * c;
    float s[kVec], q[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) s[k] = q[k] = 0.f;
    if (r0 < rl)
        for (int64_t r = (int64_t)blockIdx.x * rl + r0; r < hw; r += (int64_t)nblk * rl) {
            float v[kVec];
            ld8(xs + (size_t)r * c + cg * kVec, v);
#pragma unroll
            for (int k = 0; k < kVec; ++k) { s[k] += v[k]; q[k] += v[k] * v[k]; }
        }
#pragma unroll
    for (int k = 0; k < kVec; ++k) { red[0][threadIdx.x][k] = s[k]; red[1][threadIdx.x][k] = q[k]; }
    __syncthreads();
    if (threadIdx.x < G) {
        const int g = threadIdx.x;
        float ts = 0.f, tq = 0.f;
        for (int ch = g * cpg; ch < (g + 1) * cpg; ++ch)
            for (int r = 0; r < rl; ++r) { ts += red[0][r * cv + ch / kVec][ch % kVec]; tq += red[1][r * cv + ch / kVec][ch % kVec]; }
        float* o = partial + ((size_t)(n * nblk + blockIdx.x) * G + g) * 2;
        o[0] = ts; o[1] = tq;
    }
}
// stats[(n * G + g) * 2] = {mean, 1 / sqrt(biased var + eps)}
__global__ void gn_finalize_kernel(const float* __restrict__ partial, int nblk, int G, int total_ng, double m, float eps, float* __restrict__ stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_ng) return;
    const int n = i / G, g = i - n * G;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < nblk; ++b) {
        const float* p = partial + ((size_t)(n * nblk + b) * G + g) * 2;
        s += (double)p[0]; q += (double)p[1];
    }
    const double mean = s / m;
    double var = q / m - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}
template <typename T>
__global__ void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                const float* __restrict__ beta, int relu, T* __restrict__ y, int n, int64_t hw, int c, int G) {
    const int cv = c / kVec, cpg = c / G;
    const int64_t total = (int64_t)n * hw * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % cv);
        const int b = (int)(i / cv / hw);
        float v[kVec];
        ld8(x + i * kVec, v);
#pragma unroll
        for (int k = 0; k < kVec; ++k) {
            const int ch = cg * kVec + k;
            const float* st = stats + ((size_t)b * G + ch / cpg) * 2;
            float o = (v[k] - st[0]) * st[1] * gamma[ch] + beta[ch];
            if (relu) o = fmaxf(o, 0.f);
            v[k] = o;
        }
        st8(y + i * kVec, v);
    }
}
// Backward sweep 1: per block (blk, n) and channel: {sum dym, sum dym * xhat}, dym = dy masked by the ReLU (recomputed from x).
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                           int64_t hw, int c, int G, float* __restrict__ partial) {
    __shared__ float red[2][256][8];
    const int cv = c / kVec, rl = 256 / cv, cpg = c / G;
    const int cg = threadIdx.x % cv, r0 = threadIdx.x / cv;
    const int n = blockIdx.y, nblk = gridDim.x;
    const size_t base = (size_t)n * hw * c;
    float mean[kVec], rstd[kVec], ga[kVec], be[kVec], s[kVec], q[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
        const int ch = cg * kVec + k;
        const float* st = stats + ((size_t)n * G + ch / cpg) * 2;
        mean[k] = st[0]; rstd[k] = st[1]; ga[k] = gamma[ch]; be[k] = beta[ch];
        s[k] = q[k] = 0.f;
    }
    if (r0 < rl)
        for (int64_t r = (int64_t)blockIdx.x * rl + r0; r < hw; r += (int64_t)nblk * rl) {
            float g[kVec], v[kVec];
            ld8(dy + base + (size_t)r * c + cg * kVec, g);
            ld8(x + base + (size_t)r * c + cg * kVec, v);
#pragma unroll
            for (int k = 0; k < kVec; ++k) {
                const float xh = (v[k] - mean[k]) * rstd[k];
                const float d = (relu && xh * ga[k] + be[k] <= 0.f) ? 0.f : g[k];
                s[k] += d; q[k] += d * xh;
            }
        }
#pragma unroll
    for (int k = 0; k < kVec; ++k) { red[0][threadIdx.x][k] = s[k]; red[1][threadIdx.x][k] = q[k]; }
    __syncthreads();
    if (threadIdx.x < c) {
        const int ch = threadIdx.x;
        float ts = 0.f, tq = 0.f;
        for (int r = 0; r < rl; ++r) { ts += red[0][r * cv + ch / kVec][ch % kVec]; tq += red[1][r * cv + ch / kVec][ch % kVec]; }
        float* o = partial + ((size_t)(n * nblk + blockIdx.x) * c + ch) * 2;
        o[0] = ts; o[1] = tq;
    }
}
// dgamma / dbeta per channel (over samples and blocks, fixed order) and, per (sample, group), the two means the second sweep
// needs: gstat[(n * G + g) * 2] = {sum_c gamma * sum dym, sum_c gamma * sum dym * xhat} / (hw * cpg).  One wave per output:
// blocks [0, c) reduce a channel over n * nblk partial rows, blocks [c, c + n * G) a (sample, group) over nblk * cpg entries.
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(64) void gn_bwd_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ gamma, int n, int nblk,
                                                            int c, int G, double m, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ gstat) {
    const int cpg = c / G, lane = threadIdx.x;
    double s = 0.0, q = 0.0;
    if ((int)blockIdx.x < c) {
        const int ch = blockIdx.x;
        for (int b = lane; b < n * nblk; b += 64) { s += (double)partial[((size_t)b * c + ch) * 2]; q += (double)partial[((size_t)b * c + ch) * 2 + 1]; }
        s = wave_sum_d(s); q = wave_sum_d(q);
        if (lane == 0) { dbeta[ch] = (float)s; dgamma[ch] = (float)q; }
        return;
    }
    const int i = blockIdx.x - c, b = i / G, g = i - b * G;
    for (int e = lane; e < nblk * cpg; e += 64) {
        const int k = e / cpg, ch = g * cpg + e % cpg;
        const float* p = partial + ((size_t)(b * nblk + k) * c + ch) * 2;
        s += (double)gamma[ch] * p[0]; q += (double)gamma[ch] * p[1];
    }
    s = wave_sum_d(s); q = wave_sum_d(q);
    if (lane == 0) { gstat[2 * i] = (float)(s / m); gstat[2 * i + 1] = (float)(q / m); }
}
template <typename T>
__global__ void gn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ stats,
                                    const float* __restrict__ gstat, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    int relu, T* __restrict__ dx, int n, int64_t hw, int c, int G) {
    const int cv = c / kVec, cpg = c / G;
    const int64_t total = (int64_t)n * hw * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % cv);
        const int b = (int)(i / cv / hw);
        float g[kVec], v[kVec];
        ld8(dy + i * kVec, g);
        ld8(x + i * kVec, v);
#pragma unroll
        for (int k = 0; k < kVec; ++k) {
            const int ch = cg * kVec + k;
            const size_t sg = ((size_t)b * G + ch / cpg) * 2;
            const float xh = (v[k] - stats[sg]) * stats[sg + 1];
            const float d = (relu && xh * gamma[ch] + beta[ch] <= 0.f) ? 0.f : g[k];
            v[k] = stats[sg + 1] * (gamma[ch] * d - gstat[sg] - xh * gstat[sg + 1]);
        }
        st8(dx + i * kVec, v);
    }
}

// ---- bilinear upsampling, align_corners=True (torch's upsample_bilinear2d arithmetic: source index = dst * (in-1)/(out-1) in
// fp32, neighbours i and i + (i < in - 1), weights 1 - frac / frac) -----------------------------------------------------------
struct Lerp { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lerp lerp_of(int o, int in, float ratio) {
    const float r = ratio * (float)o;
    Lerp l;
    l.i0 = (int)r;
    l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0);
    l.w1 = r - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}
__host__ __device__ inline float ac_ratio(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

template <typename T>
__global__ void bilinear_up_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c, int f) {
    const int cv = c / kVec, ho_n = h * f, wo_n = w * f;
    const float rh = ac_ratio(h, ho_n), rw = ac_ratio(w, wo_n);
    const int64_t total = (int64_t)n * ho_n * wo_n * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int cg = t % cv; t /= cv;
        const int wo = t % wo_n; t /= wo_n;
        const int ho = t % ho_n;
        const int b = t / ho_n;
        const Lerp lh = lerp_of(ho, h, rh), lw = lerp_of(wo, w, rw);
        const T* xb = x + (size_t)b * h * w * c + cg * kVec;
        float a[kVec], bb[kVec], cc[kVec], d[kVec], o[kVec];
        ld8(xb + ((size_t)lh.i0 * w + lw.i0) * c, a);
        ld8(xb + ((size_t)lh.i0 * w + lw.i1) * c, bb);
        ld8(xb + ((size_t)lh.i1 * w + lw.i0) * c, cc);
        ld8(xb + ((size_t)lh.i1 * w + lw.i1) * c, d);
#pragma unroll
        for (int k = 0; k < kVec; ++k) o[k] = lh.w0 * (lw.w0 * a[k] + lw.w1 * bb[k]) + lh.w1 * (lw.w0 * cc[k] + lw.w1 * d[k]);
        st8(y + i * kVec, o);
    }
}
// weight of output index o on input index i along one axis (0 when o does not touch i)
__device__ __forceinline__ float lerp_weight(int o, int i, int in, float ratio) {
    const Lerp l = lerp_of(o, in, ratio);
    return (l.i0 == i ? l.w0 : 0.f) + (l.i1 == i ? l.w1 : 0.f);
}
// the adjoint as a gather: every input pixel collects the output pixels that read it (a window of about 2f per axis)
template <typename T>
__global__ void bilinear_up_nhwc_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int n, int h, int w, int c, int f, int accumulate) {
    const int cv = c / kVec, ho_n = h * f, wo_n = w * f;
    const float rh = ac_ratio(h, ho_n), rw = ac_ratio(w, wo_n);
    const int64_t total = (int64_t)n * h * w * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int cg = t % cv; t /= cv;
        const int wi = t % w; t /= w;
        const int hi = t % h;
        const int b = t / h;
        // outputs o with o * ratio in (i - 1, i + 1): a generous window, the exact membership test is lerp_weight's
        const int h_lo = rh > 0.f ? max(0, (int)((hi - 1) / rh) - 1) : 0, h_hi = rh > 0.f ? min(ho_n - 1, (int)((hi + 1) / rh) + 1) : ho_n - 1;
        const int w_lo = rw > 0.f ? max(0, (int)((wi - 1) / rw) - 1) : 0, w_hi = rw > 0.f ? min(wo_n - 1, (int)((wi + 1) / rw) + 1) : wo_n - 1;
        float acc[kVec];
#pragma unroll
        for (int k = 0; k < kVec; ++k) acc[k] = 0.f;
        if (accumulate) ld8(dx + i * kVec, acc);
        const T* gb = dy + (size_t)b * ho_n * wo_n * c + cg * kVec;
        for (int ho = h_lo; ho <= h_hi; ++ho) {
            const float wh = lerp_weight(ho, hi, h, rh);
            if (wh == 0.f) continue;
            for (int wo = w_lo; wo <= w_hi; ++wo) {
                const float ww = lerp_weight(wo, wi, w, rw);
                if (ww == 0.f) continue;
                float g[kVec];
                ld8(gb + ((size_t)ho * wo_n + wo) * c, g);
#pragma unroll
                for (int k = 0; k < kVec; ++k) acc[k] += wh * ww * g[k];
            }
        }
        st8(dx + i * kVec, acc);
    }
}
// fp32 planes [n * k][h][w] -> [n * k][h f][w f] (the segmentation head's UpsamplingBilinear2d) and its adjoint
__global__ void bilinear_up_planes_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int h, int w, int f) {
    const int ho_n = h * f, wo_n = w * f;
    const float rh = ac_ratio(h, ho_n), rw = ac_ratio(w, wo_n);
    const int64_t total = (int64_t)planes * ho_n * wo_n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int wo = (int)(i % wo_n), ho = (int)(i / wo_n % ho_n);
        const int64_t p = i / wo_n / ho_n;
        const Lerp lh = lerp_of(ho, h, rh), lw = lerp_of(wo, w, rw);
        const float* xp = x + p * h * w;
        y[i] = lh.w0 * (lw.w0 * xp[lh.i0 * w + lw.i0] + lw.w1 * xp[lh.i0 * w + lw.i1]) +
               lh.w1 * (lw.w0 * xp[lh.i1 * w + lw.i0] + lw.w1 * xp[lh.i1 * w + lw.i1]);
    }
}
__global__ void bilinear_up_planes_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int planes, int h, int w, int f) {
    const int ho_n = h * f, wo_n = w * f;
    const float rh = ac_ratio(h, ho_n), rw = ac_ratio(w, wo_n);
    const int64_t total = (int64_t)planes * h * w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int wi = (int)(i % w), hi = (int)(i / w % h);
        const int64_t p = i / w / h;
        const int h_lo = rh > 0.f ? max(0, (int)((hi - 1) / rh) - 1) : 0, h_hi = rh > 0.f ? min(ho_n - 1, (int)((hi + 1) / rh) + 1) : ho_n - 1;
        const int w_lo = rw > 0.f ? max(0, (int)((wi - 1) / rw) - 1) : 0, w_hi = rw > 0.f ? min(wo_n - 1, (int)((wi + 1) / rw) + 1) : wo_n - 1;
        const float* gp = dy + p * ho_n * wo_n;
        float acc = 0.f;
        for (int ho = h_lo; ho <= h_hi; ++ho) {
            const float wh = lerp_weight(ho, hi, h, rh);
            if (wh == 0.f) continue;
            for (int wo = w_lo; wo <= w_hi; ++wo) {
                const float ww = lerp_weight(wo, wi, w, rw);
                if (ww != 0.f) acc += wh * ww * gp[(size_t)ho * wo_n + wo];
            }
        }
        dx[i] = acc;
    }
}

// ---- Dropout2d ---------------------------------------------------------------------------------------------------------------
// mask[n * c + k] = 0 or 1 / (1 - p): one draw per (sample, channel) from a counter-based generator keyed by (seed, *counter +
// bias, index) - the counter lives in device memory so a replayed graph draws a new mask every step.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__global__ void dropout2d_mask_kernel(float* __restrict__ mask, int total, float p, uint32_t seed, const int64_t* __restrict__ counter, int64_t bias) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint64_t step = (uint64_t)((counter ? *counter : 0) + bias);
    uint32_t h = mix32(seed ^ 0x9e3779b9U);
    h = mix32(h ^ (uint32_t)step);
    h = mix32(h ^ (uint32_t)(step >> 32) ^ 0x85ebca6bU);
    h = mix32(h ^ (uint32_t)i);
    const float u = (float)(h >> 8) * (1.f / 16777216.f);
    mask[i] = u < p ? 0.f : 1.f / (1.f - p);
}
template <typename T>
__global__ void channel_scale_kernel(const T* __restrict__ x, const float* __restrict__ mask, T* __restrict__ y, int n, int64_t hw, int c) {
    const int cv = c / kVec;
    const int64_t total = (int64_t)n * hw * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % cv);
        const int b = (int)(i / cv / hw);
        float v[kVec];
        ld8(x + i * kVec, v);
        const float* m = mask + (size_t)b * c + cg * kVec;
#pragma unroll
        for (int k = 0; k < kVec; ++k) v[k] *= m[k];
        st8(y + i * kVec, v);
    }
}

bool gn_shape_ok(int c, int G) {
    const int cv = c / kVec;
    return c > 0 && c % kVec == 0 && G > 0 && c % G == 0 && c <= 256 && 256 % cv == 0 && G <= 256;
}

}  // namespace

#define VS_LAUNCH_T(kernel, grid, s, ...)                                                                  \
    do {                                                                                                    \
        VS_FOR_T(dtype, { hipLaunchKernelGGL((kernel<T>), grid, dim3(256), 0, s, __VA_ARGS__); });          \
        VS_LAUNCH_CHECK();                                                                                  \
    } while (0)

extern "C" int vs_upsample2x_add(int dtype, const void* x, const void* skip, void* y, int n, int h, int w, int c, void* stream) {
    VS_REQUIRE(x && y && c > 0 && c % kVec == 0, "upsample2x_add: channels must be a multiple of 8");      // skip may be NULL: plain nearest x2
    const int64_t total = (int64_t)n * 4 * h * w * (c / kVec);
    VS_LAUNCH_T(upsample2x_add_kernel, dim3(grid_for(total)), (hipStream_t)stream, (const T*)x, (const T*)skip, (T*)y, n, h, w, c);
    return VS_OK;
}

// GroupNorm(G, c) over x [n][hw][c]: stats [n][G][2] = {mean, 1/sqrt(var + eps)} (fp32), then y = (x - mean) * rstd * gamma + beta
// (+ ReLU).  workspace: vs_gn_workspace(n, c) bytes.
extern "C" size_t vs_gn_workspace(int n, int c) { return (size_t)n * kGnBlocks * c * 2 * sizeof(float); }
extern "C" int vs_gn_fwd(int dtype, const void* x, const float* gamma, const float* beta, int relu, void* y, float* stats, int n,
                         int64_t hw, int c, int groups, float eps, float* workspace, size_t workspace_bytes, void* stream) {
    VS_REQUIRE(x && gamma && beta && y && stats && workspace, "gn_fwd: null pointer");
    VS_REQUIRE(gn_shape_ok(c, groups), "gn_fwd: unsupported channels %d / groups %d", c, groups);
    VS_REQUIRE(workspace_bytes >= vs_gn_workspace(n, c), "gn_fwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int rl = 256 / (c / kVec);
    const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(kGnBlocks, (hw + rl - 1) / rl));
    VS_LAUNCH_T(gn_partial_kernel, dim3(nblk, n), s, (const T*)x, hw, c, groups, workspace);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(cdiv(n * groups, 256)), dim3(256), 0, s, workspace, nblk, groups, n * groups,
                       (double)hw * (c / groups), eps, stats);
    VS_LAUNCH_CHECK();
    VS_LAUNCH_T(gn_apply_kernel, dim3(grid_for((int64_t)n * hw * (c / kVec))), s, (const T*)x, stats, gamma, beta, relu, (T*)y, n, hw, c, groups);
    return VS_OK;
}
// dx, dgamma[c], dbeta[c] from dy (gradient w.r.t. the output, masked here by the ReLU when relu = 1), x and the forward's stats
extern "C" int vs_gn_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, const float* beta, int relu,
                         void* dx, float* dgamma, float* dbeta, int n, int64_t hw, int c, int groups, float* workspace,
                         size_t workspace_bytes, void* stream) {
    VS_REQUIRE(dy && x && stats && gamma && beta && dx && dgamma && dbeta && workspace, "gn_bwd: null pointer");
    VS_REQUIRE(gn_shape_ok(c, groups), "gn_bwd: unsupported channels %d / groups %d", c, groups);
    VS_REQUIRE(workspace_bytes >= vs_gn_workspace(n, c) + (size_t)n * groups * 2 * sizeof(float), "gn_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int rl = 256 / (c / kVec);
    const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(kGnBlocks, (hw + rl - 1) / rl));
    float* gstat = workspace + (size_t)n * kGnBlocks * c * 2;
    VS_LAUNCH_T(gn_bwd_partial_kernel, dim3(nblk, n), s, (const T*)dy, (const T*)x, stats, gamma, beta, relu, hw, c, groups, workspace);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(c + n * groups), dim3(64), 0, s, workspace, gamma, n, nblk, c,
                       groups, (double)hw * (c / groups), dgamma, dbeta, gstat);
    VS_LAUNCH_CHECK();
    VS_LAUNCH_T(gn_bwd_apply_kernel, dim3(grid_for((int64_t)n * hw * (c / kVec))), s, (const T*)dy, (const T*)x, stats, gstat, gamma, beta,
                relu, (T*)dx, n, hw, c, groups);
    return VS_OK;
}
extern "C" size_t vs_gn_bwd_workspace(int n, int c, int groups) { return vs_gn_workspace(n, c) + (size_t)n * groups * 2 * sizeof(float); }

// F.interpolate(x, scale_factor=f, mode="bilinear", align_corners=True) on NHWC tensors, and its adjoint (dx overwritten, or added
// to when accumulate = 1)
extern "C" int vs_bilinear_up(int dtype, const void* x, void* y, int n, int h, int w, int c, int factor, void* stream) {
    VS_REQUIRE(x && y && c > 0 && c % kVec == 0 && factor >= 1, "bilinear_up: channels must be a multiple of 8");
    const int64_t total = (int64_t)n * h * factor * w * factor * (c / kVec);
    VS_LAUNCH_T(bilinear_up_nhwc_kernel, dim3(grid_for(total)), (hipStream_t)stream, (const T*)x, (T*)y, n, h, w, c, factor);
    return VS_OK;
}
extern "C" int vs_bilinear_up_bwd(int dtype, const void* dy, void* dx, int n, int h, int w, int c, int factor, int accumulate, void* stream) {
    VS_REQUIRE(dy && dx && c > 0 && c % kVec == 0 && factor >= 1, "bilinear_up_bwd: channels must be a multiple of 8");
    const int64_t total = (int64_t)n * h * w * (c / kVec);
    VS_LAUNCH_T(bilinear_up_nhwc_bwd_kernel, dim3(grid_for(total)), (hipStream_t)stream, (const T*)dy, (T*)dx, n, h, w, c, factor, accumulate);
    return VS_OK;
}
// nn.UpsamplingBilinear2d(scale_factor=f) on fp32 planes [planes][h][w] (NCHW logits) and its adjoint
extern "C" int vs_bilinear_up_planes(const float* x, float* y, int planes, int h, int w, int factor, void* stream) {
    VS_REQUIRE(x && y && planes > 0 && factor >= 1, "bilinear_up_planes: bad arguments");
    hipLaunchKernelGGL(bilinear_up_planes_kernel, dim3(grid_for((int64_t)planes * h * factor * w * factor)), dim3(256), 0, (hipStream_t)stream,
                       x, y, planes, h, w, factor);
    VS_LAUNCH_CHECK();
    return VS_OK;
}
extern "C" int vs_bilinear_up_planes_bwd(const float* dy, float* dx, int planes, int h, int w, int factor, void* stream) {
    VS_REQUIRE(dy && dx && planes > 0 && factor >= 1, "bilinear_up_planes_bwd: bad arguments");
    hipLaunchKernelGGL(bilinear_up_planes_bwd_kernel, dim3(grid_for((int64_t)planes * h * w)), dim3(256), 0, (hipStream_t)stream, dy, dx,
                       planes, h, w, factor);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

// nn.Dropout2d(p): mask [n * c] = 0 or 1 / (1 - p) per (sample, channel), drawn from (seed, *counter + bias) - counter: device
// int64, may be NULL; vs_channel_scale applies a mask to x [n][hw][c] (forward and, on the gradient, backward)
extern "C" int vs_dropout2d_mask(float* mask, int n, int c, float p, uint32_t seed, const int64_t* counter, int64_t bias, void* stream) {
    VS_REQUIRE(mask && n > 0 && c > 0 && p >= 0.f && p < 1.f, "dropout2d_mask: bad arguments");
    hipLaunchKernelGGL(dropout2d_mask_kernel, dim3(cdiv(n * c, 256)), dim3(256), 0, (hipStream_t)stream, mask, n * c, p, seed, counter, bias);
    VS_LAUNCH_CHECK();
    return VS_OK;
}
extern "C" int vs_channel_scale(int dtype, const void* x, const float* mask, void* y, int n, int64_t hw, int c, void* stream) {
    VS_REQUIRE(x && mask && y && c > 0 && c % kVec == 0, "channel_scale: channels must be a multiple of 8");
    VS_LAUNCH_T(channel_scale_kernel, dim3(grid_for((int64_t)n * hw * (c / kVec))), (hipStream_t)stream, (const T*)x, mask, (T*)y, n, hw, c);
    return VS_OK;
}
