// Operators of smp's EfficientNet encoders (segmentation-models-pytorch 0.2.1 encoders/efficientnet.py over efficientnet-pytorch
// 0.6.3: model.py MBConvBlock / EfficientNet, utils.py Conv2dStaticSamePadding / drop_connect) that the ResNet kernels do not cover,
// NHWC tensors, HBM-bound sweeps (gfx950):
//   * BatchNorm2d over ANY channel count that is a multiple of 8 (the expanded widths 144 .. 2688), eps / momentum as arguments
//     (1e-3 / 0.01 there), followed by nothing, ReLU or swish (x * sigmoid(x)); the backward pass recomputes the activation's
//     derivative from the pre-norm tensor
//   * depthwise k x k convolution (3 / 5), stride 1 / 2, TF-style "same" static padding (pad_lo rows / columns in front, the rest
//     behind): forward, data gradient, weight gradient; with a single-channel fp32 input broadcast over the channels the same
//     kernels ARE the stem nn.Conv2d(1, cout, 3, stride=2)
//   * drop_connect (one Bernoulli draw per sample, the surviving samples scaled by 1 / keep) fused with the residual sum
// Everything here is a fixed-order reduction: same input, same bits.
#include <algorithm>

#include "common.h"

namespace {

constexpr int kVec = 8;
constexpr int kBnBlocks = 512;     // partial rows of the BatchNorm reductions
constexpr int kDwBlocks = 512;     // partial rows of the depthwise weight gradient (its pixel loop is latency-bound: many workgroups)

inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v / (1.f + __expf(-v));
    return v;
}
__device__ __forceinline__ float act_grad(float v, int act) {   // d act(v) / dv
    if (act == 1) return v > 0.f ? 1.f : 0.f;
    if (act == 2) { const float s = 1.f / (1.f + __expf(-v)); return s * (1.f + v * (1.f - s)); }
    return 1.f;
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// A workgroup of 256 lanes covers one slab of up to 256 channel vectors (blockIdx.y) and 256 / cv rows per sweep.
struct Slab {
    int cv, rpb, cvi, rl, ch0;   // vectors in this slab, rows per sweep, this lane's vector / row lane, first channel of the lane
    bool on;
};
__device__ __forceinline__ Slab slab_of(int c) {
    Slab s;
    const int cv_all = c / kVec, v0 = blockIdx.y * 256;
    s.cv = min(256, cv_all - v0);
    s.rpb = 256 / s.cv;
    s.cvi = threadIdx.x % s.cv;
    s.rl = threadIdx.x / s.cv;
    s.ch0 = (v0 + s.cvi) * kVec;
    s.on = s.rl < s.rpb;
    return s;
}

// ---- BatchNorm ----------------------------------------------------------------------------------------------------------
// partial[blk][2][c]: per workgroup sums of (x - K) and (x - K)^2, K = the tensor's first row (no cancellation when |mean| >> std)
template <typename T>
__global__ __launch_bounds__(256) void bn2_stats_partial(const T* __restrict__ x, int64_t rows, int c, float* __restrict__ partial) {
    __shared__ float red[2][256][kVec + 1];
    const Slab sl = slab_of(c);
    float s[kVec], q[kVec], sh[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) s[k] = q[k] = 0.f;
    ld8(x + sl.ch0, sh);
    if (sl.on) {
        for (int64_t r = (int64_t)blockIdx.x * sl.rpb + sl.rl; r < rows; r += (int64_t)gridDim.x * sl.rpb) {
            float v[kVec];
            ld8(x + r * c + sl.ch0, v);
#pragma unroll
            for (int k = 0; k < kVec; ++k) { const float d = v[k] - sh[k]; s[k] += d; q[k] += d * d; }
        }
    }
#pragma unroll
    for (int k = 0; k < kVec; ++k) { red[0][threadIdx.x][k] = s[k]; red[1][threadIdx.x][k] = q[k]; }
    __syncthreads();
    for (int ch = threadIdx.x; ch < sl.cv * kVec; ch += 256) {
        const int g = ch / kVec, k = ch % kVec;
        float a = 0.f, b = 0.f;
        for (int j = 0; j < sl.rpb; ++j) { a += red[0][j * sl.cv + g][k]; b += red[1][j * sl.cv + g][k]; }
        const int cc = blockIdx.y * 256 * kVec + ch;
        partial[((size_t)blockIdx.x * 2 + 0) * c + cc] = a;
        partial[((size_t)blockIdx.x * 2 + 1) * c + cc] = b;
    }
}
// one wave per channel: fp64 sums of the partial rows in a fixed order
template <typename T>
__global__ __launch_bounds__(64) void bn2_stats_finalize(const float* __restrict__ partial, const T* __restrict__ x, int nblocks, int c, int64_t rows,
                                                       float eps, float momentum, float* mean, float* invstd, float* running_mean, float* running_var) {
    const int ch = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) {
        s += (double)partial[((size_t)b * 2 + 0) * c + ch];
        q += (double)partial[((size_t)b * 2 + 1) * c + ch];
    }
    s = wave_sum_f64(s);
    q = wave_sum_f64(q);
    if (threadIdx.x != 0) return;
    const double dm = s / (double)rows;
    double var = q / (double)rows - dm * dm;
    if (var < 0.0) var = 0.0;
    const double mu = dm + (double)Elem<T>::ld(x + ch);
    mean[ch] = (float)mu;
    invstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
        running_mean[ch] = (float)((1.0 - momentum) * (double)running_mean[ch] + momentum * mu);
        running_var[ch] = (float)((1.0 - momentum) * (double)running_var[ch] + momentum * unbiased);
    }
}
// y = act((x - mean) * invstd * gamma + beta); var_eps >= 0: the second vector holds VARIANCES (evaluation from the running statistics)
template <typename T>
__global__ __launch_bounds__(256) void bn2_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, int act, float var_eps,
                                                      T* __restrict__ y, int64_t rows, int c) {
    const Slab sl = slab_of(c);
    if (!sl.on) return;
    float sc[kVec], sf[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
        const int ch = sl.ch0 + k;
        const float is = var_eps >= 0.f ? 1.f / sqrtf(invstd[ch] + var_eps) : invstd[ch];
        sc[k] = is * gamma[ch];
        sf[k] = beta[ch] - mean[ch] * sc[k];
    }
    for (int64_t r = (int64_t)blockIdx.x * sl.rpb + sl.rl; r < rows; r += (int64_t)gridDim.x * sl.rpb) {
        float v[kVec];
        ld8(x + r * c + sl.ch0, v);
#pragma unroll
        for (int k = 0; k < kVec; ++k) v[k] = act_fwd(v[k] * sc[k] + sf[k], act);
        st8(y + r * c + sl.ch0, v);
    }
}
// backward: g = dy * act'(yhat), partial sums of g and g * xhat
template <typename T>
__global__ __launch_bounds__(256) void bn2_bwd_partial(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int act, int64_t rows, int c, float* __restrict__ partial) {
    __shared__ float red[2][256][kVec + 1];
    const Slab sl = slab_of(c);
    float s[kVec], q[kVec], mu[kVec], is[kVec], ga[kVec], be[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
        const int ch = sl.ch0 + k;
        s[k] = q[k] = 0.f;
        mu[k] = mean[ch]; is[k] = invstd[ch]; ga[k] = gamma[ch]; be[k] = beta[ch];
    }
    if (sl.on) {
        for (int64_t r = (int64_t)blockIdx.x * sl.rpb + sl.rl; r < rows; r += (int64_t)gridDim.x * sl.rpb) {
            float g[kVec], v[kVec];
            ld8(dy + r * c + sl.ch0, g);
            ld8(x + r * c + sl.ch0, v);
#pragma unroll
            for (int k = 0; k < kVec; ++k) {
                const float xh = (v[k] - mu[k]) * is[k];
                const float gg = g[k] * act_grad(xh * ga[k] + be[k], act);
                s[k] += gg; q[k] += gg * xh;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kVec; ++k) { red[0][threadIdx.x][k] = s[k]; red[1][threadIdx.x][k] = q[k]; }
    __syncthreads();
    for (int ch = threadIdx.x; ch < sl.cv * kVec; ch += 256) {
        const int g = ch / kVec, k = ch % kVec;
        float a = 0.f, b = 0.f;
        for (int j = 0; j < sl.rpb; ++j) { a += red[0][j * sl.cv + g][k]; b += red[1][j * sl.cv + g][k]; }
        const int cc = blockIdx.y * 256 * kVec + ch;
        partial[((size_t)blockIdx.x * 2 + 0) * c + cc] = a;
        partial[((size_t)blockIdx.x * 2 + 1) * c + cc] = b;
    }
}
__global__ __launch_bounds__(64) void bn2_bwd_finalize(const float* __restrict__ partial, int nblocks, int c, float* dgamma, float* dbeta) {
    const int ch = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) {
        s += (double)partial[((size_t)b * 2 + 0) * c + ch];
        q += (double)partial[((size_t)b * 2 + 1) * c + ch];
    }
    s = wave_sum_f64(s);
    q = wave_sum_f64(q);
    if (threadIdx.x == 0) { dbeta[ch] = (float)s; dgamma[ch] = (float)q; }
}
template <typename T>
__global__ __launch_bounds__(256) void bn2_bwd_apply(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                                   const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const float* __restrict__ dgamma, const float* __restrict__ dbeta, int act, T* __restrict__ dx,
                                                   int64_t rows, int c) {
    const Slab sl = slab_of(c);
    if (!sl.on) return;
    const float inv_m = 1.f / (float)rows;
    float mu[kVec], is[kVec], ga[kVec], be[kVec], db[kVec], dg[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
        const int ch = sl.ch0 + k;
        mu[k] = mean[ch]; is[k] = invstd[ch]; ga[k] = gamma[ch]; be[k] = beta[ch];
        db[k] = dbeta[ch] * inv_m; dg[k] = dgamma[ch] * inv_m;
    }
    for (int64_t r = (int64_t)blockIdx.x * sl.rpb + sl.rl; r < rows; r += (int64_t)gridDim.x * sl.rpb) {
        float g[kVec], v[kVec], o[kVec];
        ld8(dy + r * c + sl.ch0, g);
        ld8(x + r * c + sl.ch0, v);
#pragma unroll
        for (int k = 0; k < kVec; ++k) {
            const float xh = (v[k] - mu[k]) * is[k];
            const float gg = g[k] * act_grad(xh * ga[k] + be[k], act);
            o[k] = ga[k] * is[k] * (gg - db[k] - xh * dg[k]);
        }
        st8(dx + r * c + sl.ch0, o);
    }
}

// ---- depthwise k x k convolution, stride s, pad_lo in front ---------------------------------------------------------------------
// A workgroup covers a slab of up to 64 channel vectors (blockIdx.y; its taps [tap][512] in LDS) and 256 / (vectors in the slab) output
// pixels per sweep.
// XT / BCAST: the input is a single-channel fp32 map broadcast over the channels (the stem).
template <typename T, typename XT, bool BCAST>
__global__ __launch_bounds__(256) void dwconv2d_fwd_kernel(const XT* __restrict__ x, const float* __restrict__ wgt, T* __restrict__ y, int n, int h, int w,
                                                         int c, int k, int stride, int pad, int dil, int ho, int wo) {
    extern __shared__ float wl[];     // [k * k][64 * 8]
    const int cv_all = c / kVec, v0 = blockIdx.y * 64, cv = min(64, cv_all - v0), kk = k * k;
    for (int o = threadIdx.x; o < cv * kVec * kk; o += 256) wl[(o % kk) * 512 + o / kk] = wgt[(size_t)v0 * kVec * kk + o];
    __syncthreads();
    const int ppb = 256 / cv, cvi = threadIdx.x % cv, pl = threadIdx.x / cv;     // 256 / cv pixels per sweep (narrow layers: many)
    if (pl >= ppb) return;
    const int ch0 = (v0 + cvi) * kVec;
    const int64_t pixels = (int64_t)n * ho * wo;
    for (int64_t p = (int64_t)blockIdx.x * ppb + pl; p < pixels; p += (int64_t)gridDim.x * ppb) {
        const int ox = (int)(p % wo), oy = (int)(p / wo % ho);
        const int64_t b = p / wo / ho;
        float acc[kVec];
#pragma unroll
        for (int q = 0; q < kVec; ++q) acc[q] = 0.f;
        for (int kh = 0; kh < k; ++kh) {
            const int iy = oy * stride + kh * dil - pad;
            if (iy < 0 || iy >= h) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int ix = ox * stride + kw * dil - pad;
                if (ix < 0 || ix >= w) continue;
                const float* wt = wl + (kh * k + kw) * 512 + cvi * kVec;
                if constexpr (BCAST) {
                    const float v = (float)x[(b * h + iy) * w + ix];
#pragma unroll
                    for (int q = 0; q < kVec; ++q) acc[q] += v * wt[q];
                } else {
                    float v[kVec];
                    ld8(x + ((b * h + iy) * w + ix) * c + ch0, v);
#pragma unroll
                    for (int q = 0; q < kVec; ++q) acc[q] += v[q] * wt[q];
                }
            }
        }
        st8(y + p * c + ch0, acc);
    }
}
// The same convolution, four consecutive outputs of a row per lane: each input row of the window is loaded once ((WT - 1) S + K vectors)
// and reused across the K taps and the WT outputs - 10 instead of 25 vector loads per output at K = 5 (the plain kernel is bound by
// these L2 re-reads).  flip = 1: taps reversed = the data gradient of a stride-1 layer with symmetric padding; accumulate adds to y.
template <typename T, int K, int S, int D>
__global__ __launch_bounds__(256) void dwconv2d_strip_kernel(const T* __restrict__ x, const float* __restrict__ wgt, T* __restrict__ y, int n, int h, int w,
                                                           int c, int pad, int ho, int wo, int flip, int accumulate, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int act) {
    constexpr int WT = 4, NW = (WT - 1) * S + (K - 1) * D + 1, KK = K * K;      // D: dilation (stride-1 stages under DeepLabV3+ / PAN / DeepLabV3)
    extern __shared__ float wl[];     // [K * K][64 * 8]
    const int cv_all = c / kVec, v0 = blockIdx.y * 64, cv = min(64, cv_all - v0);
    for (int o = threadIdx.x; o < cv * kVec * KK; o += 256) {
        const int t = o % KK;
        wl[(flip ? KK - 1 - t : t) * 512 + o / KK] = wgt[(size_t)v0 * kVec * KK + o];
    }
    __syncthreads();
    const int ppb = 256 / cv, cvi = threadIdx.x % cv, pl = threadIdx.x / cv;
    if (pl >= ppb) return;
    const int ch0 = (v0 + cvi) * kVec;
    float sc[kVec], sf[kVec];           // evaluation: the BatchNorm behind the convolution (folded) and its activation in the same sweep
#pragma unroll
    for (int q = 0; q < kVec; ++q) { sc[q] = scale ? scale[ch0 + q] : 1.f; sf[q] = scale ? shift[ch0 + q] : 0.f; }
    const int spr = (wo + WT - 1) / WT;                 // strips per output row
    const int64_t strips = (int64_t)n * ho * spr;
    for (int64_t sidx = (int64_t)blockIdx.x * ppb + pl; sidx < strips; sidx += (int64_t)gridDim.x * ppb) {
        const int sx = (int)(sidx % spr), oy = (int)(sidx / spr % ho);
        const int64_t b = sidx / spr / ho;
        const int ox0 = sx * WT, ix0 = ox0 * S - pad;
        float acc[WT][kVec];
#pragma unroll
        for (int j = 0; j < WT; ++j)
#pragma unroll
            for (int q = 0; q < kVec; ++q) acc[j][q] = 0.f;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int iy = oy * S + kh * D - pad;
            if (iy < 0 || iy >= h) continue;
            const T* row = x + ((b * h + iy) * w) * c + ch0;
            float v[NW][kVec];
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int ix = ix0 + i;
                if (ix >= 0 && ix < w) ld8(row + (size_t)ix * c, v[i]);
                else {
#pragma unroll
                    for (int q = 0; q < kVec; ++q) v[i][q] = 0.f;
                }
            }
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const float* wt = wl + (kh * K + kw) * 512 + cvi * kVec;
                float wq[kVec];
#pragma unroll
                for (int q = 0; q < kVec; ++q) wq[q] = wt[q];
#pragma unroll
                for (int j = 0; j < WT; ++j)
#pragma unroll
                    for (int q = 0; q < kVec; ++q) acc[j][q] += v[j * S + kw * D][q] * wq[q];
            }
        }
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            const int ox = ox0 + j;
            if (ox >= wo) break;
            T* yo = y + (((b * ho + oy) * wo) + ox) * c + ch0;
            if (scale) {
#pragma unroll
                for (int q = 0; q < kVec; ++q) acc[j][q] = act_fwd(acc[j][q] * sc[q] + sf[q], act);
            }
            if (accumulate) {
                float old[kVec];
                ld8(yo, old);
#pragma unroll
                for (int q = 0; q < kVec; ++q) acc[j][q] += old[q];
            }
            st8(yo, acc[j]);
        }
    }
}
template <typename T>
static void launch_dw_strip(const void* x, const float* w, void* y, int n, int h, int wd, int c, int k, int stride, int pad, int dil, int ho, int wo,
                            int flip, int accumulate, hipStream_t s, const float* scale = nullptr, const float* shift = nullptr, int act = 0) {
    const int ppb = 256 / std::min(c / kVec, 64);
    const int64_t strips = (int64_t)n * ho * ((wo + 3) / 4);
    const dim3 grid((unsigned)std::min<int64_t>((strips + ppb - 1) / ppb, 4096), (c / kVec + 63) / 64);
    const size_t lds = (size_t)k * k * 512 * sizeof(float);
#define VS_STRIP(K_, S_, D_) hipLaunchKernelGGL((dwconv2d_strip_kernel<T, K_, S_, D_>), grid, dim3(256), lds, s, (const T*)x, w, (T*)y, n, h, wd, c, pad, ho, wo, flip, accumulate, scale, shift, act)
    if (dil == 2) { if (k == 3) VS_STRIP(3, 1, 2); else VS_STRIP(5, 1, 2); }
    else if (dil == 4) VS_STRIP(3, 1, 4);
    else if (k == 3 && stride == 1) VS_STRIP(3, 1, 1);
    else if (k == 3) VS_STRIP(3, 2, 1);
    else if (stride == 1) VS_STRIP(5, 1, 1);
    else VS_STRIP(5, 2, 1);
#undef VS_STRIP
}

// dx[iy][ix] (+)= sum over the taps (kh, kw) with (iy + pad - kh) and (ix + pad - kw) divisible by the stride of dy[..] * w[kh][kw]
template <typename T>
__global__ __launch_bounds__(256) void dwconv2d_bwd_data_kernel(const T* __restrict__ dy, const float* __restrict__ wgt, T* __restrict__ dx, int n, int h,
                                                              int w, int c, int k, int stride, int pad, int dil, int ho, int wo, int accumulate) {
    extern __shared__ float wl[];
    const int cv_all = c / kVec, v0 = blockIdx.y * 64, cv = min(64, cv_all - v0), kk = k * k;
    for (int o = threadIdx.x; o < cv * kVec * kk; o += 256) wl[(o % kk) * 512 + o / kk] = wgt[(size_t)v0 * kVec * kk + o];
    __syncthreads();
    const int ppb = 256 / cv, cvi = threadIdx.x % cv, pl = threadIdx.x / cv;
    if (pl >= ppb) return;
    const int ch0 = (v0 + cvi) * kVec;
    const int64_t pixels = (int64_t)n * h * w;
    for (int64_t p = (int64_t)blockIdx.x * ppb + pl; p < pixels; p += (int64_t)gridDim.x * ppb) {
        const int ix = (int)(p % w), iy = (int)(p / w % h);
        const int64_t b = p / w / h;
        float acc[kVec];
#pragma unroll
        for (int q = 0; q < kVec; ++q) acc[q] = 0.f;
        for (int kh = 0; kh < k; ++kh) {
            const int ty = iy + pad - kh * dil;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= ho) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int tx = ix + pad - kw * dil;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= wo) continue;
                float g[kVec];
                ld8(dy + ((b * ho + oy) * wo + ox) * c + ch0, g);
                const float* wt = wl + (kh * k + kw) * 512 + cvi * kVec;
#pragma unroll
                for (int q = 0; q < kVec; ++q) acc[q] += g[q] * wt[q];
            }
        }
        if (accumulate) {
            float old[kVec];
            ld8(dx + p * c + ch0, old);
#pragma unroll
            for (int q = 0; q < kVec; ++q) acc[q] += old[q];
        }
        st8(dx + p * c + ch0, acc);
    }
}
// weight gradient, stage 1: blockIdx.z = kernel row kh; partial[blk][c][kk] = sum over the workgroup's output pixels of dy * shifted x,
// the row's k taps accumulated together (dy is read once per kernel row, not once per tap)
template <typename T, typename XT, bool BCAST>
__global__ __launch_bounds__(256) void dwconv2d_wgrad_partial(const XT* __restrict__ x, const T* __restrict__ dy, int n, int h, int w, int c, int k,
                                                            int stride, int pad, int dil, int ho, int wo, float* __restrict__ partial) {
    __shared__ float red[256][kVec + 1];
    const int cv_all = c / kVec, v0 = blockIdx.y * 64, cv = min(64, cv_all - v0), kk = k * k;
    const int kh = blockIdx.z;
    const int ppb = 256 / cv, cvi = threadIdx.x % cv, pl = threadIdx.x / cv;
    const int ch0 = (v0 + cvi) * kVec;
    float s[5][kVec];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int q = 0; q < kVec; ++q) s[t][q] = 0.f;
    const int64_t pixels = (int64_t)n * ho * wo;
    if (pl < ppb) {
        for (int64_t p = (int64_t)blockIdx.x * ppb + pl; p < pixels; p += (int64_t)gridDim.x * ppb) {
            const int ox = (int)(p % wo), oy = (int)(p / wo % ho);
            const int64_t b = p / wo / ho;
            const int iy = oy * stride + kh * dil - pad;
            if (iy < 0 || iy >= h) continue;
            float g[kVec];
            ld8(dy + p * c + ch0, g);
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) {
                const int ix = ox * stride + kw * dil - pad;
                if (kw >= k || ix < 0 || ix >= w) continue;
                if constexpr (BCAST) {
                    const float v = (float)x[(b * h + iy) * w + ix];
#pragma unroll
                    for (int q = 0; q < kVec; ++q) s[kw][q] += g[q] * v;
                } else {
                    float v[kVec];
                    ld8(x + ((b * h + iy) * w + ix) * c + ch0, v);
#pragma unroll
                    for (int q = 0; q < kVec; ++q) s[kw][q] += g[q] * v[q];
                }
            }
        }
    }
#pragma unroll
    for (int kw = 0; kw < 5; ++kw) {
        if (kw >= k) break;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kVec; ++q) red[threadIdx.x][q] = s[kw][q];
        __syncthreads();
        for (int o = threadIdx.x; o < cv * kVec; o += 256) {
            const int g = o / kVec, q = o % kVec;
            float t = 0.f;
            for (int j = 0; j < ppb; ++j) t += red[j * cv + g][q];
            partial[((size_t)blockIdx.x * c + v0 * kVec + o) * kk + kh * k + kw] = t;
        }
    }
}
// the same partial sums with four consecutive outputs of a row per lane and trip (the input row window and the four gradients are
// loaded once and shared by the row's K taps)
template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dwconv2d_wgrad_strip(const T* __restrict__ x, const T* __restrict__ dy, int n, int h, int w, int c, int pad, int ho,
                                                          int wo, float* __restrict__ partial) {
    constexpr int WT = 4, NW = (WT - 1) * S + K, KK = K * K;
    __shared__ float red[256][kVec + 1];
    const int cv_all = c / kVec, v0 = blockIdx.y * 64, cv = min(64, cv_all - v0);
    const int kh = blockIdx.z;
    const int ppb = 256 / cv, cvi = threadIdx.x % cv, pl = threadIdx.x / cv;
    const int ch0 = (v0 + cvi) * kVec;
    float s[K][kVec];
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
        for (int q = 0; q < kVec; ++q) s[t][q] = 0.f;
    const int spr = (wo + WT - 1) / WT;
    const int64_t strips = (int64_t)n * ho * spr;
    if (pl < ppb) {
        for (int64_t sidx = (int64_t)blockIdx.x * ppb + pl; sidx < strips; sidx += (int64_t)gridDim.x * ppb) {
            const int sx = (int)(sidx % spr), oy = (int)(sidx / spr % ho);
            const int64_t b = sidx / spr / ho;
            const int iy = oy * S + kh - pad;
            if (iy < 0 || iy >= h) continue;
            const int ox0 = sx * WT, ix0 = ox0 * S - pad;
            const T* row = x + ((b * h + iy) * w) * c + ch0;
            const T* grow = dy + (((b * ho + oy) * wo) + ox0) * c + ch0;
            float v[NW][kVec], g[WT][kVec];
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int ix = ix0 + i;
                if (ix >= 0 && ix < w) ld8(row + (size_t)ix * c, v[i]);
                else {
#pragma unroll
                    for (int q = 0; q < kVec; ++q) v[i][q] = 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < WT; ++j) {
                if (ox0 + j < wo) ld8(grow + (size_t)j * c, g[j]);
                else {
#pragma unroll
                    for (int q = 0; q < kVec; ++q) g[j][q] = 0.f;
                }
            }
#pragma unroll
            for (int kw = 0; kw < K; ++kw)
#pragma unroll
                for (int j = 0; j < WT; ++j)
#pragma unroll
                    for (int q = 0; q < kVec; ++q) s[kw][q] += g[j][q] * v[j * S + kw][q];
        }
    }
#pragma unroll
    for (int kw = 0; kw < K; ++kw) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kVec; ++q) red[threadIdx.x][q] = s[kw][q];
        __syncthreads();
        for (int o = threadIdx.x; o < cv * kVec; o += 256) {
            const int gq = o / kVec, q = o % kVec;
            float t = 0.f;
            for (int j = 0; j < ppb; ++j) t += red[j * cv + gq][q];
            partial[((size_t)blockIdx.x * c + v0 * kVec + o) * KK + kh * K + kw] = t;
        }
    }
}
__global__ __launch_bounds__(256) void dwconv2d_wgrad_final(const float* __restrict__ partial, float* __restrict__ dw, int nblk, int total) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // one wave per output (channel * kk + tap): the partial
    if (o >= total) return;                                                          // rows spread over its lanes, fp64 butterfly
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += (double)partial[(size_t)b * total + o];
    s = wave_sum_f64(s);
    if (lane == 0) dw[o] = (float)s;
}

// out = x * mask[n] + skip (mask NULL: 1; skip NULL: 0) - drop_connect + the residual sum, and (on the gradient, without skip) its backward
template <typename T>
__global__ void sample_scale_add_kernel(const T* __restrict__ x, const float* __restrict__ mask, const T* __restrict__ skip, T* __restrict__ y, int n,
                                        int64_t per_sample) {
    const int64_t pv = per_sample / kVec, total = (int64_t)n * pv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = mask ? mask[i / pv] : 1.f;
        float v[kVec];
        ld8(x + i * kVec, v);
        if (skip) {
            float s[kVec];
            ld8(skip + i * kVec, s);
#pragma unroll
            for (int k = 0; k < kVec; ++k) v[k] = v[k] * m + s[k];
        } else {
#pragma unroll
            for (int k = 0; k < kVec; ++k) v[k] *= m;
        }
        st8(y + i * kVec, v);
    }
}

// out[n][c] = scale * sum over the sample's hw rows of a (* b): the average pool of the squeeze-excitation branch and, with b, the
// gate's gradient - any channel count that is a multiple of 8 (blockIdx.x = sample, blockIdx.y = slab of 256 channel vectors)
// (1024 lanes: one workgroup per sample and slab is all the parallelism there is, so the rows are spread over as many lanes as fit)
template <typename T>
__global__ __launch_bounds__(1024) void sample_rowsum_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t hw, int c, float scale) {
    __shared__ float red[1024][kVec + 1];
    Slab sl;
    {
        const int cv_all = c / kVec, v0 = blockIdx.y * 256;
        sl.cv = min(256, cv_all - v0);
        sl.rpb = 1024 / sl.cv;
        sl.cvi = threadIdx.x % sl.cv;
        sl.rl = threadIdx.x / sl.cv;
        sl.ch0 = (v0 + sl.cvi) * kVec;
        sl.on = sl.rl < sl.rpb;
    }
    const T* as = a + (size_t)blockIdx.x * hw * c + sl.ch0;
    const T* bs = b ? b + (size_t)blockIdx.x * hw * c + sl.ch0 : nullptr;
    float s[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) s[k] = 0.f;
    if (sl.on) {
        for (int64_t r = sl.rl; r < hw; r += 4 * sl.rpb) {
            float v[4][kVec], g[4][kVec];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t rr = r + (int64_t)u * sl.rpb;
                if (rr < hw) {
                    ld8(as + rr * c, v[u]);
                    if (bs) ld8(bs + rr * c, g[u]);
                } else {
#pragma unroll
                    for (int k = 0; k < kVec; ++k) v[u][k] = 0.f;
                    if (bs) {
#pragma unroll
                        for (int k = 0; k < kVec; ++k) g[u][k] = 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < kVec; ++k) s[k] += bs ? v[u][k] * g[u][k] : v[u][k];
        }
    }
#pragma unroll
    for (int k = 0; k < kVec; ++k) red[threadIdx.x][k] = s[k];
    __syncthreads();
    for (int ch = threadIdx.x; ch < sl.cv * kVec; ch += 1024) {
        const int g = ch / kVec, k = ch % kVec;
        float t = 0.f;
        for (int j = 0; j < sl.rpb; ++j) t += red[j * sl.cv + g][k];
        Elem<T>::st(out + (size_t)blockIdx.x * c + blockIdx.y * 256 * kVec + ch, t * scale);
    }
}

// the same sums with the sample's rows split over blockIdx.z (few samples, large maps: prediction batches): fp32 partials
// ws[(sample * splits + z)][c], then one lane per (sample, channel) adds the splits in order
template <typename T>
__global__ __launch_bounds__(256) void sample_rowsum_split_kernel(const T* __restrict__ a, const T* __restrict__ b, float* __restrict__ ws, int64_t hw, int c,
                                                                int cb) {      // cb: channels of b (<= c, dividing it: b repeats across the blocks of c)
    __shared__ float red[256][kVec + 1];
    const Slab sl = slab_of(c);
    const int splits = gridDim.z;
    const int64_t r0 = hw * blockIdx.z / splits, r1 = hw * (blockIdx.z + 1) / splits;
    const T* as = a + (size_t)blockIdx.x * hw * c + sl.ch0;
    const T* bs = b ? b + (size_t)blockIdx.x * hw * cb + sl.ch0 % cb : nullptr;
    float s[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) s[k] = 0.f;
    if (sl.on) {
        for (int64_t r = r0 + sl.rl; r < r1; r += 4 * sl.rpb) {
            float v[4][kVec], g[4][kVec];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t rr = r + (int64_t)u * sl.rpb;
                const bool ok = rr < r1;
                ld8(as + (ok ? rr : r) * c, v[u]);
                if (bs) ld8(bs + (ok ? rr : r) * cb, g[u]);
                if (!ok) {
#pragma unroll
                    for (int k = 0; k < kVec; ++k) v[u][k] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < kVec; ++k) s[k] += bs ? v[u][k] * g[u][k] : v[u][k];
        }
    }
#pragma unroll
    for (int k = 0; k < kVec; ++k) red[threadIdx.x][k] = s[k];
    __syncthreads();
    for (int ch = threadIdx.x; ch < sl.cv * kVec; ch += 256) {
        const int g = ch / kVec, k = ch % kVec;
        float t = 0.f;
        for (int j = 0; j < sl.rpb; ++j) t += red[j * sl.cv + g][k];
        ws[((size_t)blockIdx.x * splits + blockIdx.z) * c + blockIdx.y * 256 * kVec + ch] = t;
    }
}
template <typename T>
__global__ void sample_rowsum_finish_kernel(const float* __restrict__ ws, T* __restrict__ out, int n, int c, int splits, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * c) return;
    const int b = i / c, ch = i % c;
    float t = 0.f;
    for (int z = 0; z < splits; ++z) t += ws[((size_t)b * splits + z) * c + ch];
    Elem<T>::st(out + i, t * scale);
}

inline int bn2_blocks(int64_t rows, int c) {
    const int cv = std::min(c / kVec, 256), rpb = 256 / cv;
    const int64_t nb = (rows + (int64_t)rpb * 4 - 1) / ((int64_t)rpb * 4);
    return (int)std::max<int64_t>(1, std::min<int64_t>(kBnBlocks, nb));
}

}  // namespace

#define VS_LAUNCH_T(kernel, grid, lds, s, ...)                                                                          \
    do {                                                                                                                \
        VS_FOR_T(dtype, { hipLaunchKernelGGL((kernel<T>), grid, dim3(256), lds, s, __VA_ARGS__); });                   \
        VS_LAUNCH_CHECK();                                                                                              \
    } while (0)

// nn.BatchNorm2d(c, eps, momentum) on x [rows][c], c any multiple of 8; act 0 none / 1 ReLU / 2 swish.  workspace: vs_bn2_workspace(c).
extern "C" size_t vs_bn2_workspace(int c) { return (size_t)kBnBlocks * 2 * c * sizeof(float); }
extern "C" int vs_bn2_stats(int dtype, const void* x, int64_t rows, int c, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                            float* running_var, float* workspace, size_t workspace_bytes, void* stream) {
    VS_REQUIRE(x && mean && invstd && rows > 0 && c > 0 && c % kVec == 0, "bn2_stats: channels must be a multiple of 8 (got %d)", c);
    VS_REQUIRE(workspace && workspace_bytes >= vs_bn2_workspace(c), "bn2_stats: workspace too small");
    const int nb = bn2_blocks(rows, c), slabs = (c / kVec + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    VS_LAUNCH_T(bn2_stats_partial, dim3(nb, slabs), 0, s, (const T*)x, rows, c, workspace);
    VS_FOR_T(dtype, hipLaunchKernelGGL(bn2_stats_finalize<T>, dim3(c), dim3(64), 0, s, workspace, (const T*)x, nb, c, rows, eps, momentum, mean, invstd, running_mean, running_var));
    VS_LAUNCH_CHECK();
    return VS_OK;
}
// y = act((x - mean) * invstd * gamma + beta); var_eps >= 0: `invstd` holds VARIANCES and invstd = 1 / sqrt(var + var_eps) (evaluation)
extern "C" int vs_bn2_apply(int dtype, const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                            float var_eps, void* y, int64_t rows, int c, void* stream) {
    VS_REQUIRE(x && mean && invstd && gamma && beta && y && rows > 0 && c > 0 && c % kVec == 0 && act >= 0 && act <= 2, "bn2_apply: bad arguments");
    VS_LAUNCH_T(bn2_apply_kernel, dim3(bn2_blocks(rows, c) * 2, (c / kVec + 255) / 256), 0, (hipStream_t)stream, (const T*)x, mean, invstd, gamma, beta, act,
                var_eps, (T*)y, rows, c);
    return VS_OK;
}
extern "C" int vs_bn2_bwd(int dtype, const void* dy, const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                          int act, void* dx, float* dgamma, float* dbeta, int64_t rows, int c, float* workspace, size_t workspace_bytes, void* stream) {
    VS_REQUIRE(dy && x && mean && invstd && gamma && beta && dx && dgamma && dbeta && rows > 0 && c > 0 && c % kVec == 0 && act >= 0 && act <= 2,
               "bn2_bwd: bad arguments");
    VS_REQUIRE(workspace && workspace_bytes >= vs_bn2_workspace(c), "bn2_bwd: workspace too small");
    const int nb = bn2_blocks(rows, c), slabs = (c / kVec + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    VS_LAUNCH_T(bn2_bwd_partial, dim3(nb, slabs), 0, s, (const T*)dy, (const T*)x, mean, invstd, gamma, beta, act, rows, c, workspace);
    hipLaunchKernelGGL(bn2_bwd_finalize, dim3(c), dim3(64), 0, s, workspace, nb, c, dgamma, dbeta);
    VS_LAUNCH_CHECK();
    VS_LAUNCH_T(bn2_bwd_apply, dim3(nb * 2, slabs), 0, s, (const T*)dy, (const T*)x, mean, invstd, gamma, beta, dgamma, dbeta, act, (T*)dx, rows, c);
    return VS_OK;
}

// nn.Conv2d(c, c, k, stride, groups=c, bias=False) behind efficientnet-pytorch's Conv2dStaticSamePadding: pad_lo zero rows / columns in
// front (what is needed behind follows from ho / wo); x [n][h][w][c], w fp32 [c][k * k], y [n][ho][wo][c].  x_single_channel = 1: x is an
// fp32 [n][h][w] map broadcast over the c channels - nn.Conv2d(1, c, k, stride, bias=False), the stem on greyscale slices.
extern "C" int vs_dwconv2d(int dtype, const void* x, const float* w, void* y, int n, int h, int wd, int c, int k, int stride, int pad_lo, int dilation, int ho, int wo,
                           int x_single_channel, void* stream) {
    VS_REQUIRE(x && w && y && n > 0 && c > 0 && c % kVec == 0 && (k == 2 || k == 3 || k == 5) && (stride == 1 || stride == 2) && pad_lo >= 0 && dilation >= 1 && pad_lo <= (k - 1) * dilation,
               "dwconv2d: kernel 3 / 5, stride 1 / 2, channels a multiple of 8");
    VS_REQUIRE(ho > 0 && wo > 0 && (ho - 1) * stride - pad_lo < h && (wo - 1) * stride - pad_lo < wd, "dwconv2d: output %dx%d does not fit input %dx%d", ho, wo, h, wd);
    const int ppb = 256 / std::min(c / kVec, 64);
    const dim3 grid((unsigned)std::min<int64_t>(((int64_t)n * ho * wo + ppb - 1) / ppb, 4096), (c / kVec + 63) / 64);
    const size_t lds = (size_t)k * k * 512 * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const bool strip_ok = k != 2 && (dilation == 1 || (stride == 1 && (dilation == 2 || (dilation == 4 && k == 3))));
    if (!x_single_channel && strip_ok) {      // four outputs per lane, every input row of the window loaded once
        VS_FOR_T(dtype, launch_dw_strip<T>(x, w, y, n, h, wd, c, k, stride, pad_lo, dilation, ho, wo, 0, 0, s));
        VS_LAUNCH_CHECK();
        return VS_OK;
    }
    if (x_single_channel) {
        VS_FOR_T(dtype, hipLaunchKernelGGL((dwconv2d_fwd_kernel<T, float, true>), grid, dim3(256), lds, s, (const float*)x, w, (T*)y, n, h, wd, c, k, stride, pad_lo, dilation, ho, wo));
    } else {
        VS_FOR_T(dtype, hipLaunchKernelGGL((dwconv2d_fwd_kernel<T, T, false>), grid, dim3(256), lds, s, (const T*)x, w, (T*)y, n, h, wd, c, k, stride, pad_lo, dilation, ho, wo));
    }
    VS_LAUNCH_CHECK();
    return VS_OK;
}
extern "C" int vs_dwconv2d_bwd_data(int dtype, const void* dy, const float* w, void* dx, int n, int h, int wd, int c, int k, int stride, int pad_lo, int dilation,
                                    int ho, int wo, int accumulate, void* stream) {
    VS_REQUIRE(dy && w && dx && n > 0 && c > 0 && c % kVec == 0 && (k == 2 || k == 3 || k == 5) && (stride == 1 || stride == 2) && pad_lo >= 0 && dilation >= 1 && pad_lo <= (k - 1) * dilation,
               "dwconv2d_bwd_data: kernel 3 / 5, stride 1 / 2, channels a multiple of 8");
    const bool strip_ok = k != 2 && (dilation == 1 || dilation == 2 || (dilation == 4 && k == 3));
    if (stride == 1 && strip_ok && 2 * pad_lo == (k - 1) * dilation && ho == h && wo == wd) {   // a stride-1 "same" layer: the forward sweep with the taps reversed
        VS_FOR_T(dtype, launch_dw_strip<T>(dy, w, dx, n, h, wd, c, k, 1, pad_lo, dilation, h, wd, 1, accumulate, (hipStream_t)stream));
        VS_LAUNCH_CHECK();
        return VS_OK;
    }
    const int ppb = 256 / std::min(c / kVec, 64);
    const dim3 grid((unsigned)std::min<int64_t>(((int64_t)n * h * wd + ppb - 1) / ppb, 4096), (c / kVec + 63) / 64);
    VS_LAUNCH_T(dwconv2d_bwd_data_kernel, grid, (size_t)k * k * 512 * sizeof(float), (hipStream_t)stream, (const T*)dy, w, (T*)dx, n, h, wd, c, k, stride, pad_lo,
                dilation, ho, wo, accumulate);
    return VS_OK;
}
extern "C" size_t vs_dwconv2d_wgrad_workspace(int c, int k) { return (size_t)kDwBlocks * c * k * k * sizeof(float); }
extern "C" int vs_dwconv2d_wgrad(int dtype, const void* x, const void* dy, float* dw, int n, int h, int wd, int c, int k, int stride, int pad_lo, int dilation, int ho, int wo,
                                 int x_single_channel, float* workspace, size_t workspace_bytes, void* stream) {
    VS_REQUIRE(x && dy && dw && n > 0 && c > 0 && c % kVec == 0 && (k == 2 || k == 3 || k == 5) && (stride == 1 || stride == 2) && pad_lo >= 0 && dilation >= 1 && pad_lo <= (k - 1) * dilation,
               "dwconv2d_wgrad: kernel 3 / 5, stride 1 / 2, channels a multiple of 8");
    VS_REQUIRE(workspace && workspace_bytes >= vs_dwconv2d_wgrad_workspace(c, k), "dwconv2d_wgrad: workspace too small");
    const int ppb = 256 / std::min(c / kVec, 64);
    // >= 16 pixels per lane where the layer has them: few partial rows for the deep 8 x 8 maps (their reduction reads nblk * c * k * k floats)
    const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(kDwBlocks, ((int64_t)n * ho * wo + (int64_t)ppb * 16 - 1) / ((int64_t)ppb * 16)));
    const dim3 grid(nblk, (c / kVec + 63) / 64, k);
    hipStream_t s = (hipStream_t)stream;
    if (!x_single_channel && dilation == 1) {
#define VS_WSTRIP(T_, K_, S_) hipLaunchKernelGGL((dwconv2d_wgrad_strip<T_, K_, S_>), grid, dim3(256), 0, s, (const T_*)x, (const T_*)dy, n, h, wd, c, pad_lo, ho, wo, workspace)
        VS_FOR_T(dtype, {
            if (k == 3 && stride == 1) VS_WSTRIP(T, 3, 1); else if (k == 3) VS_WSTRIP(T, 3, 2); else if (stride == 1) VS_WSTRIP(T, 5, 1); else VS_WSTRIP(T, 5, 2);
        });
#undef VS_WSTRIP
    } else if (x_single_channel) {
        VS_FOR_T(dtype, hipLaunchKernelGGL((dwconv2d_wgrad_partial<T, float, true>), grid, dim3(256), 0, s, (const float*)x, (const T*)dy, n, h, wd, c, k, stride, pad_lo, dilation, ho, wo, workspace));
    } else {
        VS_FOR_T(dtype, hipLaunchKernelGGL((dwconv2d_wgrad_partial<T, T, false>), grid, dim3(256), 0, s, (const T*)x, (const T*)dy, n, h, wd, c, k, stride, pad_lo, dilation, ho, wo, workspace));
    }
    VS_LAUNCH_CHECK();
    const int total = c * k * k;
    hipLaunchKernelGGL(dwconv2d_wgrad_final, dim3((total + 3) / 4), dim3(256), 0, s, workspace, dw, nblk, total);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

// y [n][per_sample] = x * mask[n] + skip (mask may be NULL = 1, skip may be NULL = 0): efficientnet-pytorch's drop_connect (mask from
// vs_dropout2d_mask with c = 1: 0 or 1 / keep per sample) and the block's `x + inputs` in one sweep
extern "C" int vs_sample_scale_add(int dtype, const void* x, const float* mask, const void* skip, void* y, int n, int64_t per_sample, void* stream) {
    VS_REQUIRE(x && y && n > 0 && per_sample > 0 && per_sample % kVec == 0, "sample_scale_add: per-sample size must be a multiple of 8");
    VS_LAUNCH_T(sample_scale_add_kernel, dim3(grid_for((int64_t)n * per_sample / kVec)), 0, (hipStream_t)stream, (const T*)x, mask, (const T*)skip, (T*)y, n, per_sample);
    return VS_OK;
}
// out [n][c] = scale * sum over hw of a[n][hw][c] (* b[n][hw][c] when b is given), any c that is a multiple of 8
extern "C" int vs_sample_rowsum(int dtype, const void* a, const void* b, void* out, int n, int64_t hw, int c, float scale, void* stream) {
    VS_REQUIRE(a && out && n > 0 && hw > 0 && c > 0 && c % kVec == 0, "sample_rowsum: channels must be a multiple of 8");
    VS_FOR_T(dtype, hipLaunchKernelGGL(sample_rowsum_kernel<T>, dim3(n, (c / kVec + 255) / 256), dim3(1024), 0, (hipStream_t)stream, (const T*)a, (const T*)b,
                           (T*)out, hw, c, scale));
    VS_LAUNCH_CHECK();
    return VS_OK;
}
// the same with the rows of a sample spread over up to 64 workgroups (workspace: vs_sample_rowsum_workspace(n, c) bytes; NULL or too
// small: the one-workgroup-per-sample form above)
extern "C" size_t vs_sample_rowsum_workspace(int n, int c) { return (size_t)n * 64 * c * sizeof(float); }
extern "C" int vs_sample_rowsum_ws(int dtype, const void* a, const void* b, void* out, int n, int64_t hw, int c, float scale, float* workspace,
                                   size_t workspace_bytes, void* stream) {
    VS_REQUIRE(a && out && n > 0 && hw > 0 && c > 0 && c % kVec == 0, "sample_rowsum: channels must be a multiple of 8");
    const int rpb = 256 / std::min(c / kVec, 256);
    int splits = (int)std::min<int64_t>(64, hw / ((int64_t)rpb * 16));       // >= 16 rows per lane and split
    if (splits < 2 || !workspace || workspace_bytes < vs_sample_rowsum_workspace(n, c)) return vs_sample_rowsum(dtype, a, b, out, n, hw, c, scale, stream);
    VS_LAUNCH_T(sample_rowsum_split_kernel, dim3(n, (c / kVec + 255) / 256, splits), 0, (hipStream_t)stream, (const T*)a, (const T*)b, workspace, hw, c, c);
    VS_LAUNCH_T(sample_rowsum_finish_kernel, dim3((n * c + 255) / 256), 0, (hipStream_t)stream, workspace, (T*)out, n, c, splits, scale);
    return VS_OK;
}
// out [n][c] = sum over hw of a[n][hw][c] * b[n][hw][cb], b repeated across the c / cb channel blocks of a (ResNeSt: the attention's gradient
// = sums of the split tensor times the gated sum's gradient).  Always the split form (workspace as for vs_sample_rowsum_ws).
extern "C" int vs_sample_rowsum_b(int dtype, const void* a, const void* b, int cb, void* out, int n, int64_t hw, int c, float* workspace,
                                  size_t workspace_bytes, void* stream) {
    VS_REQUIRE(a && b && out && n > 0 && hw > 0 && c > 0 && c % kVec == 0 && cb > 0 && cb % kVec == 0 && c % cb == 0, "sample_rowsum_b: bad channel counts %d / %d", c, cb);
    VS_REQUIRE(workspace && workspace_bytes >= vs_sample_rowsum_workspace(n, c), "sample_rowsum_b: workspace too small");
    const int rpb = 256 / std::min(c / kVec, 256);
    const int splits = (int)std::max<int64_t>(1, std::min<int64_t>(64, hw / ((int64_t)rpb * 16)));
    VS_LAUNCH_T(sample_rowsum_split_kernel, dim3(n, (c / kVec + 255) / 256, splits), 0, (hipStream_t)stream, (const T*)a, (const T*)b, workspace, hw, c, cb);
    VS_LAUNCH_T(sample_rowsum_finish_kernel, dim3((n * c + 255) / 256), 0, (hipStream_t)stream, workspace, (T*)out, n, c, splits, 1.f);
    return VS_OK;
}
// evaluation form of depthwise convolution + BatchNorm + activation in ONE sweep: y = act(conv(x) * scale[c] + shift[c]) (scale / shift
// = the folded running statistics, vs_bn_fold).  Takes what the strip kernels take: dilation 1, or stride 1 with dilation 2 (4 for k = 3).
extern "C" int vs_dwconv2d_affine(int dtype, const void* x, const float* w, const float* scale, const float* shift, int act, void* y, int n, int h, int wd,
                                  int c, int k, int stride, int pad_lo, int dilation, int ho, int wo, void* stream) {
    VS_REQUIRE(x && w && scale && shift && y && n > 0 && c > 0 && c % kVec == 0 && (k == 3 || k == 5) && (stride == 1 || stride == 2) && act >= 0 && act <= 2,
               "dwconv2d_affine: bad arguments");
    VS_REQUIRE(dilation == 1 || (stride == 1 && (dilation == 2 || (dilation == 4 && k == 3))), "dwconv2d_affine: dilation %d at stride %d, kernel %d is not taken", dilation, stride, k);
    VS_REQUIRE(ho > 0 && wo > 0 && (ho - 1) * stride - pad_lo < h && (wo - 1) * stride - pad_lo < wd, "dwconv2d_affine: output %dx%d does not fit input %dx%d", ho, wo, h, wd);
    VS_FOR_T(dtype, launch_dw_strip<T>(x, w, y, n, h, wd, c, k, stride, pad_lo, dilation, ho, wo, 0, 0, (hipStream_t)stream, scale, shift, act));
    VS_LAUNCH_CHECK();
    return VS_OK;
}
