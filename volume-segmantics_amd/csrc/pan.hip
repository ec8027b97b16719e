// The Feature Pyramid Attention block of smp.PAN (segmentation-models-pytorch 0.2.1, decoders/pan/decoder.py: FPABlock) on NHWC tensors
// (gfx950).  Its pyramid works on SINGLE-CHANNEL maps of 1/2, 1/4 and 1/8 of the bottleneck's resolution - a few thousand values per
// batch - through 7x7 / 5x5 / 3x3 convolutions, train-mode BatchNorm (statistics over the whole batch), 2x2 max-pools and bilinear
// (align_corners) upsamplings.  Kernels:
//   * maxpool2x2 (NHWC, all channels) forward / backward
//   * the 7x7 convolution from C channels to the first single-channel map: forward, data gradient, weight gradient
//   * vs_fpa_pyramid_fwd / _bwd: everything between that map and the block's attention plane in ONE workgroup (fp32, every
//     intermediate kept in a caller-provided arena for the backward pass; __syncthreads between the stages; fixed-order sums)
//   * the combination out = plane * mid + pooled branch, and its gradients; an element-wise sigmoid for the GAU gates
#include <algorithm>

#include "common.h"

namespace {

constexpr int kVec = 8;
inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// ---- nn.MaxPool2d(2, 2) on x [n][h][w][c] -> y [n][h/2][w/2][c]; backward: the gradient goes to the FIRST maximum in scan order ----
template <typename T>
__global__ void maxpool2_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c) {
    const int cv = c / kVec, ho = h / 2, wo = w / 2;
    const int64_t total = (int64_t)n * ho * wo * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int cg = t % cv; t /= cv;
        const int j = t % wo; t /= wo;
        const int r = t % ho;
        const int64_t b = t / ho;
        float m[kVec];
#pragma unroll
        for (int k = 0; k < kVec; ++k) m[k] = -3.0e38f;
        for (int q = 0; q < 4; ++q) {
            float v[kVec];
            ld8(x + (((size_t)b * h + 2 * r + (q >> 1)) * w + 2 * j + (q & 1)) * c + cg * kVec, v);
#pragma unroll
            for (int k = 0; k < kVec; ++k) m[k] = fmaxf(m[k], v[k]);
        }
        st8(y + i * kVec, m);
    }
}
template <typename T>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int n, int h, int w, int c, int accumulate) {
    const int cv = c / kVec, ho = h / 2, wo = w / 2;
    const int64_t total = (int64_t)n * ho * wo * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int cg = t % cv; t /= cv;
        const int j = t % wo; t /= wo;
        const int r = t % ho;
        const int64_t b = t / ho;
        float v[4][kVec], g[kVec];
        int arg[kVec];
        float m[kVec];
        ld8(dy + i * kVec, g);
#pragma unroll
        for (int k = 0; k < kVec; ++k) { m[k] = -3.0e38f; arg[k] = 0; }
        for (int q = 0; q < 4; ++q) {
            ld8(x + (((size_t)b * h + 2 * r + (q >> 1)) * w + 2 * j + (q & 1)) * c + cg * kVec, v[q]);
#pragma unroll
            for (int k = 0; k < kVec; ++k) if (v[q][k] > m[k]) { m[k] = v[q][k]; arg[k] = q; }
        }
        for (int q = 0; q < 4; ++q) {
            T* d = dx + (((size_t)b * h + 2 * r + (q >> 1)) * w + 2 * j + (q & 1)) * c + cg * kVec;
            float o[kVec];
#pragma unroll
            for (int k = 0; k < kVec; ++k) o[k] = arg[k] == q ? g[k] : 0.f;
            if (accumulate) {
                float old[kVec];
                ld8(d, old);
#pragma unroll
                for (int k = 0; k < kVec; ++k) o[k] += old[k];
            }
            st8(d, o);
        }
    }
}

// ---- k x k convolution (padding k / 2) from x [n][h][w][c] (T) to ONE output channel: z [n][h][w] fp32 = bias + sum x * wgt[kh][kw][c] ----
template <typename T>
__global__ __launch_bounds__(256) void conv_to_plane_kernel(const T* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias,
                                                          float* __restrict__ z, int n, int h, int w, int c, int k) {
    const int lane = threadIdx.x & 63;
    const int64_t px = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);        // one wave per output pixel
    if (px >= (int64_t)n * h * w) return;
    const int wo = (int)(px % w), ho = (int)(px / w % h);
    const int64_t b = px / w / h;
    const int cv = c / kVec, pad = k / 2;
    float acc = 0.f;
    for (int tap = 0; tap < k * k; ++tap) {
        const int hi = ho + tap / k - pad, wi = wo + tap % k - pad;
        if (hi < 0 || hi >= h || wi < 0 || wi >= w) continue;
        for (int cg = lane; cg < cv; cg += 64) {
            float v[kVec];
            ld8(x + (((size_t)b * h + hi) * w + wi) * c + cg * kVec, v);
            const float* wp = wgt + (size_t)tap * c + cg * kVec;
#pragma unroll
            for (int q = 0; q < kVec; ++q) acc += v[q] * wp[q];
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) z[px] = acc + bias[0];
}
// dx [n][h][w][c] = sum_taps dz[shifted] * wgt[tap][c]
template <typename T>
__global__ void conv_to_plane_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ wgt, T* __restrict__ dx, int n, int h, int w, int c, int k) {
    const int cv = c / kVec, pad = k / 2;
    const int64_t total = (int64_t)n * h * w * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int cg = t % cv; t /= cv;
        const int wi = t % w; t /= w;
        const int hi = t % h;
        const int64_t b = t / h;
        float acc[kVec];
#pragma unroll
        for (int q = 0; q < kVec; ++q) acc[q] = 0.f;
        for (int tap = 0; tap < k * k; ++tap) {
            const int ho = hi - (tap / k - pad), wo = wi - (tap % k - pad);
            if (ho < 0 || ho >= h || wo < 0 || wo >= w) continue;
            const float g = dz[((size_t)b * h + ho) * w + wo];
            const float* wp = wgt + (size_t)tap * c + cg * kVec;
#pragma unroll
            for (int q = 0; q < kVec; ++q) acc[q] += g * wp[q];
        }
        st8(dx + i * kVec, acc);
    }
}
// dw[tap][c] = sum over pixels of dz * x[shifted]; db = sum dz  (one thread per (tap, 8-channel vector), sequential over the pixels)
template <typename T>
__global__ void conv_to_plane_wgrad_kernel(const T* __restrict__ x, const float* __restrict__ dz, float* __restrict__ dw, float* __restrict__ db,
                                           int n, int h, int w, int c, int k) {
    const int cv = c / kVec, pad = k / 2;
    const int total = k * k * cv;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        float s = 0.f;
        for (int64_t p = 0; p < (int64_t)n * h * w; ++p) s += dz[p];
        db[0] = s;
    }
    if (i >= total) return;
    const int cg = i % cv, tap = i / cv;
    float acc[kVec];
#pragma unroll
    for (int q = 0; q < kVec; ++q) acc[q] = 0.f;
    for (int b = 0; b < n; ++b)
        for (int ho = 0; ho < h; ++ho) {
            const int hi = ho + tap / k - pad;
            if (hi < 0 || hi >= h) continue;
            for (int wo = 0; wo < w; ++wo) {
                const int wi = wo + tap % k - pad;
                if (wi < 0 || wi >= w) continue;
                const float g = dz[((size_t)b * h + ho) * w + wo];
                float v[kVec];
                ld8(x + (((size_t)b * h + hi) * w + wi) * c + cg * kVec, v);
#pragma unroll
                for (int q = 0; q < kVec; ++q) acc[q] += g * v[q];
            }
        }
#pragma unroll
    for (int q = 0; q < kVec; ++q) dw[(size_t)tap * c + cg * kVec + q] = acc[q];
}

// ---- the single-channel pyramid, one workgroup ------------------------------------------------------------------------------------
constexpr int kPT = 1024;     // threads of the pyramid workgroup
struct FpaParams {            // parameters of the six ConvBnRelu(1 or C -> 1) layers behind the first convolution, in smp's order of use
    // index: 0 = down1's BatchNorm only (its convolution is conv_to_plane), 1 = down2 (5x5), 2 = down3.1 (3x3), 3 = down3.2 (3x3),
    // 4 = conv2 (5x5), 5 = conv1 (7x7)
    const float* w[6]; const float* b[6]; const float* gamma[6]; const float* beta[6];
    float* rmean[6]; float* rvar[6];                          // running statistics (updated in training)
    float* dw[6]; float* db[6]; float* dgamma[6]; float* dbeta[6];   // gradients (backward only)
};
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < kPT / 64; ++i) s += red[i];
    __syncthreads();
    return s;
}
__device__ void p_conv(const float* in, float* out, const float* wgt, const float* bias, int n, int h, int w, int k) {
    const int pad = k / 2;
    for (int i = threadIdx.x; i < n * h * w; i += kPT) {
        const int wo = i % w, ho = i / w % h, b = i / w / h;
        float acc = bias[0];
        for (int t = 0; t < k * k; ++t) {
            const int hi = ho + t / k - pad, wi = wo + t % k - pad;
            if (hi >= 0 && hi < h && wi >= 0 && wi < w) acc += in[(b * h + hi) * w + wi] * wgt[t];
        }
        out[i] = acc;
    }
    __syncthreads();
}
// dout -> din (+= when acc), dw[k*k], db
__device__ void p_conv_bwd(const float* dout, const float* in, const float* wgt, float* din, float* dw, float* db, int n, int h, int w, int k, float* red) {
    const int pad = k / 2;
    for (int i = threadIdx.x; i < n * h * w; i += kPT) {
        const int wi = i % w, hi = i / w % h, b = i / w / h;
        float acc = 0.f;
        for (int t = 0; t < k * k; ++t) {
            const int ho = hi - (t / k - pad), wo = wi - (t % k - pad);
            if (ho >= 0 && ho < h && wo >= 0 && wo < w) acc += dout[(b * h + ho) * w + wo] * wgt[t];
        }
        din[i] = acc;
    }
    for (int t = 0; t < k * k; ++t) {
        float s = 0.f;
        for (int i = threadIdx.x; i < n * h * w; i += kPT) {
            const int wo = i % w, ho = i / w % h, b = i / w / h;
            const int hi = ho + t / k - pad, wi = wo + t % k - pad;
            if (hi >= 0 && hi < h && wi >= 0 && wi < w) s += dout[i] * in[(b * h + hi) * w + wi];
        }
        s = block_sum(s, red);
        if (threadIdx.x == 0) dw[t] = s;
    }
    float s = 0.f;
    for (int i = threadIdx.x; i < n * h * w; i += kPT) s += dout[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) db[0] = s;
    __syncthreads();
}
// train: batch statistics (biased variance) + running update; eval: running statistics.  a = relu(gamma * (z - mean) * rstd + beta)
__device__ void p_bn_relu(const float* z, float* a, const float* gamma, const float* beta, float* rmean, float* rvar, float* stat, int m,
                          int training, float* red) {
    float mean, rstd;
    if (training) {
        float s = 0.f;
        for (int i = threadIdx.x; i < m; i += kPT) s += z[i];
        mean = block_sum(s, red) / (float)m;
        float q = 0.f;
        for (int i = threadIdx.x; i < m; i += kPT) { const float d = z[i] - mean; q += d * d; }
        const float var = block_sum(q, red) / (float)m;
        rstd = rsqrtf(var + 1e-5f);
        if (threadIdx.x == 0) {
            stat[0] = mean; stat[1] = rstd;
            rmean[0] = 0.9f * rmean[0] + 0.1f * mean;
            rvar[0] = 0.9f * rvar[0] + 0.1f * (m > 1 ? var * (float)m / (float)(m - 1) : var);
        }
    } else {
        mean = rmean[0]; rstd = rsqrtf(rvar[0] + 1e-5f);
    }
    const float g = gamma[0], be = beta[0];
    for (int i = threadIdx.x; i < m; i += kPT) a[i] = fmaxf(g * (z[i] - mean) * rstd + be, 0.f);
    __syncthreads();
}
// da (gradient w.r.t. the post-ReLU output) -> dz; dgamma, dbeta
__device__ void p_bn_relu_bwd(const float* da, const float* z, const float* gamma, const float* beta, const float* stat, float* dz, float* dgamma,
                              float* dbeta, int m, float* red) {
    const float mean = stat[0], rstd = stat[1], g = gamma[0], be = beta[0];
    float s = 0.f, q = 0.f;
    for (int i = threadIdx.x; i < m; i += kPT) {
        const float xh = (z[i] - mean) * rstd;
        const float d = (g * xh + be > 0.f) ? da[i] : 0.f;
        s += d; q += d * xh;
    }
    s = block_sum(s, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0) { dbeta[0] = s; dgamma[0] = q; }
    for (int i = threadIdx.x; i < m; i += kPT) {
        const float xh = (z[i] - mean) * rstd;
        const float d = (g * xh + be > 0.f) ? da[i] : 0.f;
        dz[i] = g * rstd * (d - s / (float)m - xh * q / (float)m);
    }
    __syncthreads();
}
__device__ void p_pool(const float* in, float* out, int n, int h, int w) {      // (h, w) = input size
    const int ho = h / 2, wo = w / 2;
    for (int i = threadIdx.x; i < n * ho * wo; i += kPT) {
        const int j = i % wo, r = i / wo % ho, b = i / wo / ho;
        const float* p = in + (b * h + 2 * r) * w + 2 * j;
        out[i] = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[w], p[w + 1]));
    }
    __syncthreads();
}
__device__ void p_pool_bwd(const float* dout, const float* in, float* din, int n, int h, int w, int accumulate) {
    const int ho = h / 2, wo = w / 2;
    for (int i = threadIdx.x; i < n * ho * wo; i += kPT) {
        const int j = i % wo, r = i / wo % ho, b = i / wo / ho;
        const int base = (b * h + 2 * r) * w + 2 * j;
        const int off[4] = {0, 1, w, w + 1};
        int arg = 0; float m = in[base];
        for (int q = 1; q < 4; ++q) if (in[base + off[q]] > m) { m = in[base + off[q]]; arg = q; }
        for (int q = 0; q < 4; ++q) {
            const float v = q == arg ? dout[i] : 0.f;
            din[base + off[q]] = accumulate ? din[base + off[q]] + v : v;
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void p_lerp(int o, int in, float ratio, int& i0, int& i1, float& w0, float& w1) {
    const float r = ratio * (float)o;
    i0 = (int)r; i1 = i0 + (i0 < in - 1 ? 1 : 0); w1 = r - (float)i0; w0 = 1.f - w1;
}
// bilinear, align_corners=True, (h, w) -> (2h, 2w); out = up(in) (+ add when add != nullptr)
__device__ void p_up2(const float* in, const float* add, float* out, int n, int h, int w) {
    const int ho = 2 * h, wo = 2 * w;
    const float rh = ho > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.f, rw = wo > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.f;
    for (int i = threadIdx.x; i < n * ho * wo; i += kPT) {
        const int x = i % wo, y = i / wo % ho, b = i / wo / ho;
        int y0, y1, x0, x1; float a0, a1, c0, c1;
        p_lerp(y, h, rh, y0, y1, a0, a1); p_lerp(x, w, rw, x0, x1, c0, c1);
        const float* p = in + b * h * w;
        const float v = a0 * (c0 * p[y0 * w + x0] + c1 * p[y0 * w + x1]) + a1 * (c0 * p[y1 * w + x0] + c1 * p[y1 * w + x1]);
        out[i] = add ? v + add[i] : v;
    }
    __syncthreads();
}
__device__ void p_up2_bwd(const float* dout, float* din, int n, int h, int w) {      // (h, w) = the small size; gather form
    const int ho = 2 * h, wo = 2 * w;
    const float rh = ho > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.f, rw = wo > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.f;
    for (int i = threadIdx.x; i < n * h * w; i += kPT) {
        const int xi = i % w, yi = i / w % h, b = i / w / h;
        float acc = 0.f;
        for (int y = max(0, 2 * yi - 3); y <= min(ho - 1, 2 * yi + 4); ++y) {
            int y0, y1; float a0, a1;
            p_lerp(y, h, rh, y0, y1, a0, a1);
            const float wy = (y0 == yi ? a0 : 0.f) + (y1 == yi ? a1 : 0.f);
            if (wy == 0.f) continue;
            for (int x = max(0, 2 * xi - 3); x <= min(wo - 1, 2 * xi + 4); ++x) {
                int x0, x1; float c0, c1;
                p_lerp(x, w, rw, x0, x1, c0, c1);
                const float wx = (x0 == xi ? c0 : 0.f) + (x1 == xi ? c1 : 0.f);
                if (wx != 0.f) acc += wy * wx * dout[(b * ho + y) * wo + x];
            }
        }
        din[i] = acc;
    }
    __syncthreads();
}

// arena layout (floats): every intermediate of the forward pass, then the backward pass's gradient planes
struct FpaArena {
    // sizes: m1 = n h1 w1 (1/2), m2 (1/4), m3 (1/8), m0 = n h w (full)
    int m0, m1, m2, m3;
    __host__ __device__ int z1() const { return 0; }                 // input: down1's convolution output (m1) - written by the caller
    __host__ __device__ int a1() const { return m1; }                // x1 = relu(bn(z1))
    __host__ __device__ int p2() const { return 2 * m1; }            // pool(x1)                       (m2)
    __host__ __device__ int z2() const { return 2 * m1 + m2; }       // conv5(p2)
    __host__ __device__ int a2() const { return 2 * m1 + 2 * m2; }   // x2
    __host__ __device__ int p3() const { return 2 * m1 + 3 * m2; }   // pool(x2)                       (m3)
    __host__ __device__ int z3() const { return 2 * m1 + 3 * m2 + m3; }
    __host__ __device__ int a3() const { return 2 * m1 + 3 * m2 + 2 * m3; }
    __host__ __device__ int z4() const { return 2 * m1 + 3 * m2 + 3 * m3; }
    __host__ __device__ int a4() const { return 2 * m1 + 3 * m2 + 4 * m3; }      // x3
    __host__ __device__ int z5() const { return 2 * m1 + 3 * m2 + 5 * m3; }      // conv2(x2)            (m2)
    __host__ __device__ int a5() const { return 2 * m1 + 4 * m2 + 5 * m3; }
    __host__ __device__ int s2() const { return 2 * m1 + 5 * m2 + 5 * m3; }      // a5 + up(x3)          (m2)
    __host__ __device__ int z6() const { return 2 * m1 + 6 * m2 + 5 * m3; }      // conv1(x1)            (m1)
    __host__ __device__ int a6() const { return 3 * m1 + 6 * m2 + 5 * m3; }
    __host__ __device__ int s1() const { return 4 * m1 + 6 * m2 + 5 * m3; }      // a6 + up(s2)          (m1)
    __host__ __device__ int stats() const { return 5 * m1 + 6 * m2 + 5 * m3; }   // 6 x {mean, rstd}
    __host__ __device__ int grads() const { return stats() + 16; }               // backward scratch: 3 x m1 + 3 x m2 + 3 x m3
    __host__ __device__ int total() const { return grads() + 3 * m1 + 3 * m2 + 3 * m3; }
};

// forward: arena.z1 holds down1's convolution output; plane [n][h][w] receives up(s1)
__global__ __launch_bounds__(kPT) void fpa_pyramid_fwd_kernel(float* __restrict__ A, float* __restrict__ plane, FpaParams P, FpaArena L, int n, int h,
                                                             int w, int training) {
    __shared__ float red[kPT / 64];
    const int h1 = h / 2, w1 = w / 2, h2 = h / 4, w2 = w / 4, h3 = h / 8, w3 = w / 8;
    float* st = A + L.stats();
    p_bn_relu(A + L.z1(), A + L.a1(), P.gamma[0], P.beta[0], P.rmean[0], P.rvar[0], st + 0, L.m1, training, red);
    p_pool(A + L.a1(), A + L.p2(), n, h1, w1);
    p_conv(A + L.p2(), A + L.z2(), P.w[1], P.b[1], n, h2, w2, 5);
    p_bn_relu(A + L.z2(), A + L.a2(), P.gamma[1], P.beta[1], P.rmean[1], P.rvar[1], st + 2, L.m2, training, red);
    p_pool(A + L.a2(), A + L.p3(), n, h2, w2);
    p_conv(A + L.p3(), A + L.z3(), P.w[2], P.b[2], n, h3, w3, 3);
    p_bn_relu(A + L.z3(), A + L.a3(), P.gamma[2], P.beta[2], P.rmean[2], P.rvar[2], st + 4, L.m3, training, red);
    p_conv(A + L.a3(), A + L.z4(), P.w[3], P.b[3], n, h3, w3, 3);
    p_bn_relu(A + L.z4(), A + L.a4(), P.gamma[3], P.beta[3], P.rmean[3], P.rvar[3], st + 6, L.m3, training, red);
    p_conv(A + L.a2(), A + L.z5(), P.w[4], P.b[4], n, h2, w2, 5);
    p_bn_relu(A + L.z5(), A + L.a5(), P.gamma[4], P.beta[4], P.rmean[4], P.rvar[4], st + 8, L.m2, training, red);
    p_up2(A + L.a4(), A + L.a5(), A + L.s2(), n, h3, w3);            // x = conv2(x2) + up(x3)
    p_conv(A + L.a1(), A + L.z6(), P.w[5], P.b[5], n, h1, w1, 7);
    p_bn_relu(A + L.z6(), A + L.a6(), P.gamma[5], P.beta[5], P.rmean[5], P.rvar[5], st + 10, L.m1, training, red);
    p_up2(A + L.s2(), A + L.a6(), A + L.s1(), n, h2, w2);            // x = up(x) + conv1(x1)
    p_up2(A + L.s1(), nullptr, plane, n, h1, w1);
}
// backward: dplane [n][h][w] -> arena.grads()[0 .. m1) = gradient w.r.t. z1 (down1's convolution output); parameter gradients through P
__global__ __launch_bounds__(kPT) void fpa_pyramid_bwd_kernel(float* __restrict__ A, const float* __restrict__ dplane, FpaParams P, FpaArena L, int n,
                                                             int h, int w) {
    __shared__ float red[kPT / 64];
    const int h1 = h / 2, w1 = w / 2, h2 = h / 4, w2 = w / 4, h3 = h / 8, w3 = w / 8;
    const float* st = A + L.stats();
    float* G = A + L.grads();
    float *g1a = G, *g1b = G + L.m1, *g1c = G + 2 * L.m1;                      // three m1-sized planes
    float *g2a = G + 3 * L.m1, *g2b = g2a + L.m2, *g2c = g2a + 2 * L.m2;        // three m2-sized
    float *g3a = g2a + 3 * L.m2, *g3b = g3a + L.m3, *g3c = g3a + 2 * L.m3;      // three m3-sized
    p_up2_bwd(dplane, g1a, n, h1, w1);                                        // d s1
    // s1 = a6 + up(s2): d a6 = d s1; d s2 = up^T(d s1)
    p_up2_bwd(g1a, g2a, n, h2, w2);                                           // d s2
    p_bn_relu_bwd(g1a, A + L.z6(), P.gamma[5], P.beta[5], st + 10, g1b, P.dgamma[5], P.dbeta[5], L.m1, red);     // d z6
    p_conv_bwd(g1b, A + L.a1(), P.w[5], g1c, P.dw[5], P.db[5], n, h1, w1, 7, red);                             // g1c = d a1 (from conv1)
    // s2 = a5 + up(a4): d a5 = d s2; d a4 = up^T(d s2)
    p_up2_bwd(g2a, g3a, n, h3, w3);                                           // d a4
    p_bn_relu_bwd(g2a, A + L.z5(), P.gamma[4], P.beta[4], st + 8, g2b, P.dgamma[4], P.dbeta[4], L.m2, red);      // d z5
    p_conv_bwd(g2b, A + L.a2(), P.w[4], g2c, P.dw[4], P.db[4], n, h2, w2, 5, red);                             // g2c = d a2 (from conv2)
    p_bn_relu_bwd(g3a, A + L.z4(), P.gamma[3], P.beta[3], st + 6, g3b, P.dgamma[3], P.dbeta[3], L.m3, red);      // d z4
    p_conv_bwd(g3b, A + L.a3(), P.w[3], g3c, P.dw[3], P.db[3], n, h3, w3, 3, red);                             // d a3
    p_bn_relu_bwd(g3c, A + L.z3(), P.gamma[2], P.beta[2], st + 4, g3b, P.dgamma[2], P.dbeta[2], L.m3, red);      // d z3
    p_conv_bwd(g3b, A + L.p3(), P.w[2], g3a, P.dw[2], P.db[2], n, h3, w3, 3, red);                             // d p3
    p_pool_bwd(g3a, A + L.a2(), g2c, n, h2, w2, 1);                                                            // d a2 += pool^T(d p3)
    p_bn_relu_bwd(g2c, A + L.z2(), P.gamma[1], P.beta[1], st + 2, g2b, P.dgamma[1], P.dbeta[1], L.m2, red);      // d z2
    p_conv_bwd(g2b, A + L.p2(), P.w[1], g2a, P.dw[1], P.db[1], n, h2, w2, 5, red);                             // d p2
    p_pool_bwd(g2a, A + L.a1(), g1c, n, h1, w1, 1);                                                            // d a1 += pool^T(d p2)
    p_bn_relu_bwd(g1c, A + L.z1(), P.gamma[0], P.beta[0], st + 0, g1a, P.dgamma[0], P.dbeta[0], L.m1, red);      // d z1 -> G[0 .. m1)
}

// out[n][p][c] = plane[n][p] * mid[n][p][c] + b1[n][c]
template <typename T>
__global__ void fpa_combine_kernel(const float* __restrict__ plane, const T* __restrict__ mid, const T* __restrict__ b1, T* __restrict__ out, int n,
                                   int64_t hw, int c) {
    const int cv = c / kVec;
    const int64_t total = (int64_t)n * hw * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % cv);
        const int64_t px = i / cv;
        const int b = (int)(px / hw);
        float m[kVec], g[kVec];
        ld8(mid + i * kVec, m);
        ld8(b1 + (size_t)b * c + cg * kVec, g);
        const float pl = plane[px];
#pragma unroll
        for (int k = 0; k < kVec; ++k) m[k] = pl * m[k] + g[k];
        st8(out + i * kVec, m);
    }
}
// dmid = dy * plane;  dplane[n][p] = sum_c dy * mid   (one wave per pixel)
template <typename T>
__global__ __launch_bounds__(256) void fpa_combine_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ plane, const T* __restrict__ mid,
                                                            T* __restrict__ dmid, float* __restrict__ dplane, int64_t npx, int c) {
    const int lane = threadIdx.x & 63;
    const int64_t px = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (px >= npx) return;
    const int cv = c / kVec;
    const float pl = plane[px];
    float acc = 0.f;
    for (int cg = lane; cg < cv; cg += 64) {
        float g[kVec], m[kVec], o[kVec];
        ld8(dy + (size_t)px * c + cg * kVec, g);
        ld8(mid + (size_t)px * c + cg * kVec, m);
#pragma unroll
        for (int k = 0; k < kVec; ++k) { acc += g[k] * m[k]; o[k] = g[k] * pl; }
        st8(dmid + (size_t)px * c + cg * kVec, o);
    }
    acc = wave_sum(acc);
    if (lane == 0) dplane[px] = acc;
}

template <typename T>
__global__ void sigmoid_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        float v[kVec];
        ld8(x + i * kVec, v);
#pragma unroll
        for (int k = 0; k < kVec; ++k) v[k] = 1.f / (1.f + __expf(-v[k]));
        st8(y + i * kVec, v);
    }
}
template <typename T>
__global__ void sigmoid_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        float g[kVec], s[kVec];
        ld8(dy + i * kVec, g); ld8(y + i * kVec, s);
#pragma unroll
        for (int k = 0; k < kVec; ++k) g[k] *= s[k] * (1.f - s[k]);
        st8(dx + i * kVec, g);
    }
}
// shift[c] += scale[c] * bias[c]: folds a convolution's own bias into its evaluation-mode BatchNorm constants
__global__ void fold_bias_kernel(const float* __restrict__ scale, const float* __restrict__ bias, float* __restrict__ shift, int c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) shift[i] += scale[i] * bias[i];
}

}  // namespace

#define VS_LAUNCH_T(kernel, grid, s, ...)                                                                                \
    do {                                                                                                                  \
        VS_FOR_T(dtype, { hipLaunchKernelGGL((kernel<T>), grid, dim3(256), 0, s, __VA_ARGS__); });                     \
        VS_LAUNCH_CHECK();                                                                                                \
    } while (0)

extern "C" int vs_maxpool2x2(int dtype, const void* x, void* y, int n, int h, int w, int c, void* stream) {
    VS_REQUIRE(x && y && c % kVec == 0 && !(h & 1) && !(w & 1), "maxpool2x2: even dims, channels a multiple of 8");
    VS_LAUNCH_T(maxpool2_kernel, dim3(grid_for((int64_t)n * (h / 2) * (w / 2) * (c / kVec))), (hipStream_t)stream, (const T*)x, (T*)y, n, h, w, c);
    return VS_OK;
}
extern "C" int vs_maxpool2x2_bwd(int dtype, const void* x, const void* dy, void* dx, int n, int h, int w, int c, int accumulate, void* stream) {
    VS_REQUIRE(x && dy && dx && c % kVec == 0 && !(h & 1) && !(w & 1), "maxpool2x2_bwd: even dims, channels a multiple of 8");
    VS_LAUNCH_T(maxpool2_bwd_kernel, dim3(grid_for((int64_t)n * (h / 2) * (w / 2) * (c / kVec))), (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dx,
                n, h, w, c, accumulate);
    return VS_OK;
}
// k x k convolution (k odd, padding k / 2) from c channels to ONE: z [n][h][w] fp32; w fp32 [k*k][c] (torch's [1][c][k][k] in this library's
// channels-last storage), bias [1]
extern "C" int vs_conv_to_plane(int dtype, const void* x, const float* w, const float* bias, float* z, int n, int h, int wd, int c, int k, void* stream) {
    VS_REQUIRE(x && w && bias && z && c % kVec == 0 && (k & 1), "conv_to_plane: bad arguments");
    VS_LAUNCH_T(conv_to_plane_kernel, dim3((unsigned)(((int64_t)n * h * wd + 3) / 4)), (hipStream_t)stream, (const T*)x, w, bias, z, n, h, wd, c, k);
    return VS_OK;
}
extern "C" int vs_conv_to_plane_bwd(int dtype, const void* x, const float* w, const float* dz, void* dx, float* dw, float* db, int n, int h, int wd,
                                    int c, int k, void* stream) {
    VS_REQUIRE(x && w && dz && dx && dw && db && c % kVec == 0 && (k & 1), "conv_to_plane_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VS_LAUNCH_T(conv_to_plane_dgrad_kernel, dim3(grid_for((int64_t)n * h * wd * (c / kVec))), s, dz, w, (T*)dx, n, h, wd, c, k);
    VS_LAUNCH_T(conv_to_plane_wgrad_kernel, dim3(cdiv(k * k * (c / kVec), 256)), s, (const T*)x, dz, dw, db, n, h, wd, c, k);
    return VS_OK;
}

// the single-channel pyramid of FPABlock between down1's convolution output z1 [n][h/2][w/2] (written by the caller at the start of the
// arena) and the attention plane [n][h][w].  params: 6 layers x {conv weight, conv bias, BN gamma, beta, running mean, running var}
// (layer 0: the BatchNorm of down1 only - weight / bias pointers unused); grads (bwd): 6 x {dw, db, dgamma, dbeta}.  h, w multiples of 8.
extern "C" size_t vs_fpa_arena_floats(int n, int h, int w) {
    FpaArena L{n * h * w, n * (h / 2) * (w / 2), n * (h / 4) * (w / 4), n * (h / 8) * (w / 8)};
    return (size_t)L.total();
}
static FpaParams fpa_params(float* const* params, float* const* grads) {
    FpaParams P{};
    for (int i = 0; i < 6; ++i) {
        P.w[i] = params[6 * i + 0]; P.b[i] = params[6 * i + 1]; P.gamma[i] = params[6 * i + 2]; P.beta[i] = params[6 * i + 3];
        P.rmean[i] = params[6 * i + 4]; P.rvar[i] = params[6 * i + 5];
        if (grads) { P.dw[i] = grads[4 * i + 0]; P.db[i] = grads[4 * i + 1]; P.dgamma[i] = grads[4 * i + 2]; P.dbeta[i] = grads[4 * i + 3]; }
    }
    return P;
}
extern "C" int vs_fpa_pyramid_fwd(float* arena, float* plane, float* const* params, int n, int h, int w, int training, void* stream) {
    VS_REQUIRE(arena && plane && params && h >= 8 && w >= 8 && !(h & 7) && !(w & 7), "fpa_pyramid_fwd: the bottleneck must be a multiple of 8 wide and high");
    FpaArena L{n * h * w, n * (h / 2) * (w / 2), n * (h / 4) * (w / 4), n * (h / 8) * (w / 8)};
    hipLaunchKernelGGL(fpa_pyramid_fwd_kernel, dim3(1), dim3(kPT), 0, (hipStream_t)stream, arena, plane, fpa_params(params, nullptr), L, n, h, w, training);
    VS_LAUNCH_CHECK();
    return VS_OK;
}
// dplane -> arena[vs_fpa_dz1_offset(..)] (gradient w.r.t. z1) and the parameter gradients
extern "C" size_t vs_fpa_dz1_offset(int n, int h, int w) {
    FpaArena L{n * h * w, n * (h / 2) * (w / 2), n * (h / 4) * (w / 4), n * (h / 8) * (w / 8)};
    return (size_t)L.grads();
}
extern "C" int vs_fpa_pyramid_bwd(float* arena, const float* dplane, float* const* params, float* const* grads, int n, int h, int w, void* stream) {
    VS_REQUIRE(arena && dplane && params && grads && !(h & 7) && !(w & 7), "fpa_pyramid_bwd: bad arguments");
    FpaArena L{n * h * w, n * (h / 2) * (w / 2), n * (h / 4) * (w / 4), n * (h / 8) * (w / 8)};
    hipLaunchKernelGGL(fpa_pyramid_bwd_kernel, dim3(1), dim3(kPT), 0, (hipStream_t)stream, arena, dplane, fpa_params(params, grads), L, n, h, w);
    VS_LAUNCH_CHECK();
    return VS_OK;
}
extern "C" int vs_fpa_combine(int dtype, const float* plane, const void* mid, const void* b1, void* out, int n, int64_t hw, int c, void* stream) {
    VS_REQUIRE(plane && mid && b1 && out && c % kVec == 0, "fpa_combine: bad arguments");
    VS_LAUNCH_T(fpa_combine_kernel, dim3(grid_for((int64_t)n * hw * (c / kVec))), (hipStream_t)stream, plane, (const T*)mid, (const T*)b1, (T*)out, n, hw, c);
    return VS_OK;
}
extern "C" int vs_fpa_combine_bwd(int dtype, const void* dy, const float* plane, const void* mid, void* dmid, float* dplane, int n, int64_t hw, int c,
                                  void* stream) {
    VS_REQUIRE(dy && plane && mid && dmid && dplane && c % kVec == 0, "fpa_combine_bwd: bad arguments");
    VS_LAUNCH_T(fpa_combine_bwd_kernel, dim3((unsigned)(((int64_t)n * hw + 3) / 4)), (hipStream_t)stream, (const T*)dy, plane, (const T*)mid, (T*)dmid,
                dplane, (int64_t)n * hw, c);
    return VS_OK;
}
extern "C" int vs_sigmoid(int dtype, const void* x, void* y, int64_t elems, void* stream) {
    VS_REQUIRE(x && y && elems % kVec == 0, "sigmoid: element count must be a multiple of 8");
    VS_LAUNCH_T(sigmoid_kernel, dim3(grid_for(elems / kVec)), (hipStream_t)stream, (const T*)x, (T*)y, elems / kVec);
    return VS_OK;
}
extern "C" int vs_sigmoid_bwd(int dtype, const void* dy, const void* y, void* dx, int64_t elems, void* stream) {
    VS_REQUIRE(dy && y && dx && elems % kVec == 0, "sigmoid_bwd: element count must be a multiple of 8");
    VS_LAUNCH_T(sigmoid_bwd_kernel, dim3(grid_for(elems / kVec)), (hipStream_t)stream, (const T*)dy, (const T*)y, (T*)dx, elems / kVec);
    return VS_OK;
}
extern "C" int vs_bn_fold_bias(const float* scale, const float* bias, float* shift, int c, void* stream) {
    VS_REQUIRE(scale && bias && shift && c > 0, "bn_fold_bias: bad arguments");
    hipLaunchKernelGGL(fold_bias_kernel, dim3(cdiv(c, 256)), dim3(256), 0, (hipStream_t)stream, scale, bias, shift, c);
    VS_LAUNCH_CHECK();
    return VS_OK;
}
