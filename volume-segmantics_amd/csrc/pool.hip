// Spatial data-movement kernels on NHWC tensors (gfx950, HBM-bound, 16-byte accesses per lane):
//  * MaxPool2d(3, stride 2, pad 1) forward (+ uint8 argmax-in-window) and gather-form backward
//    (torchvision ResNet stem, inside smp's encoder);
//  * backward of the decoder's nearest x2 upsampling (2x2 sum);
//  * zero stuffing used to express the dgrad of the stride-2 convolutions as a stride-1 convolution.
#include "common.h"
#include "prof.h"

namespace {

constexpr int kVec = 8;

template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int n,
                                   int h, int w, int c, int xcd) {
    const int ho_n = h / 2, wo_n = w / 2, cv = c / kVec;
    const int64_t total = (int64_t)n * ho_n * wo_n * cv;
    // (overlapping 3x3 windows: XCD-local block order keeps a row's second reader on the same L2 - common.h: xcd_block)
    for (int64_t i = (int64_t)xcd_block(xcd) * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int cg = t % cv; t /= cv;
        const int wo = t % wo_n; t /= wo_n;
        const int ho = t % ho_n;
        const int b = t / ho_n;
        float best[kVec];
        int bi[kVec];
#pragma unroll
        for (int k = 0; k < kVec; ++k) { best[k] = -INFINITY; bi[k] = 0; }
        // first max in (kh, kw) scan order wins (torch CPU max_pool2d: `val > maxval`)
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = 2 * ho - 1 + kh;
            if (hi < 0 || hi >= h) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = 2 * wo - 1 + kw;
                if (wi < 0 || wi >= w) continue;
                float v[kVec];
                ld8(x + (((size_t)b * h + hi) * w + wi) * c + cg * kVec, v);
#pragma unroll
                for (int k = 0; k < kVec; ++k)
                    if (v[k] > best[k]) { best[k] = v[k]; bi[k] = kh * 3 + kw; }
            }
        }
        const size_t o = (((size_t)b * ho_n + ho) * wo_n + wo) * c + cg * kVec;
        st8(y + o, best);
        if (idx) {
            uint2 pk;
            pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
            pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
            *reinterpret_cast<uint2*>(idx + o) = pk;
        }
    }
}

// Gather form, one thread per 2x2 block of INPUT pixels (rows 2i, 2i+1; columns 2j, 2j+1) and 8 channels: the block lies
// in exactly four pooling windows, (i, j), (i, j+1), (i+1, j), (i+1, j+1) - even rows / columns belong to one window row /
// column (tap 1), odd ones to two (tap 2 of window i, tap 0 of window i+1) - so four (dy, argmax) loads serve four
// outputs (per-pixel threads needed nine).  No atomics: every input element is written by one thread.
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* __restrict__ dx,
                                   int accumulate, int n, int h, int w, int c) {
    const int ho_n = h / 2, wo_n = w / 2, cv = c / kVec;
    const int64_t total = (int64_t)n * ho_n * wo_n * cv;
    for (int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t0 < total; t0 += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = t0;
        const int cg = t % cv; t /= cv;
        const int j = t % wo_n; t /= wo_n;
        const int i = t % ho_n;
        const int b = t / ho_n;
        float g[2][2][kVec];    // [row parity][column parity]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const size_t xo = (((size_t)b * h + 2 * i + a) * w + 2 * j + e) * c + cg * kVec;
                if (accumulate) ld8(dx + xo, g[a][e]);
                else {
#pragma unroll
                    for (int k = 0; k < kVec; ++k) g[a][e][k] = 0.f;
                }
            }
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
                const int ho = i + di, wo = j + dj;
                if (ho >= ho_n || wo >= wo_n) continue;
                const size_t o = (((size_t)b * ho_n + ho) * wo_n + wo) * c + cg * kVec;
                const uint2 pk = *reinterpret_cast<const uint2*>(idx + o);
                float d[kVec];
                ld8(dy + o, d);
                // window (ho, wo) covers input rows 2ho-1 .. 2ho+1: of this block, row 2i (tap kh = 1) and 2i+1 (kh = 2) when
                // di = 0, only row 2i+1 (kh = 0) when di = 1; columns alike
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int kh = 2 * i + a - (2 * ho - 1);
                    if (kh < 0 || kh > 2) continue;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int kw = 2 * j + e - (2 * wo - 1);
                        if (kw < 0 || kw > 2) continue;
                        const int code = kh * 3 + kw;
#pragma unroll
                        for (int k = 0; k < kVec; ++k) {
                            const int id = ((k < 4 ? pk.x : pk.y) >> (8 * (k & 3))) & 0xff;
                            if (id == code) g[a][e][k] += d[k];
                        }
                    }
                }
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                st8(dx + (((size_t)b * h + 2 * i + a) * w + 2 * j + e) * c + cg * kVec, g[a][e]);
    }
}

template <typename T>
__global__ void upsample2x_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int n, int h, int w, int c, int accumulate) {
    const int cv = c / kVec;
    const int64_t total = (int64_t)n * h * w * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int cg = t % cv; t /= cv;
        const int wi = t % w; t /= w;
        const int hi = t % h;
        const int b = t / h;
        float s[kVec];
#pragma unroll
        for (int k = 0; k < kVec; ++k) s[k] = 0.f;
        for (int a = 0; a < 2; ++a)
            for (int bb = 0; bb < 2; ++bb) {
                float v[kVec];
                ld8(dy + (((size_t)b * 2 * h + 2 * hi + a) * 2 * w + 2 * wi + bb) * c + cg * kVec, v);
#pragma unroll
                for (int k = 0; k < kVec; ++k) s[k] += v[k];
            }
        if (accumulate) {   // the tensor already holds gradient contributions of other consumers (U-Net++ dense skips)
            float o[kVec];
            ld8(dx + (((size_t)b * h + hi) * w + wi) * c + cg * kVec, o);
#pragma unroll
            for (int k = 0; k < kVec; ++k) s[k] += o[k];
        }
        st8(dx + (((size_t)b * h + hi) * w + wi) * c + cg * kVec, s);
    }
}

// Channel-slice copy between two NHWC tensors viewed as [rows][c_src] / [rows][c_dst]: dst[r][dst_off + k] (=|+=) src[r][src_off + k]
// for k < c.  U-Net++'s dense skips: the forward pass gathers the members of a concatenation into one tensor, the backward pass
// adds the slices of its gradient back onto the members (a member is read by several nodes, so its gradient accumulates).
template <typename T>
__global__ void channel_slice_kernel(const T* __restrict__ src, int c_src, int src_off, T* __restrict__ dst, int c_dst, int dst_off,
                                     int c, int64_t rows, int accumulate) {
    const int cv = c / kVec;
    const int64_t total = rows * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cv;
        const int cg = (int)(i - r * cv);
        float v[kVec];
        ld8(src + (size_t)r * c_src + src_off + cg * kVec, v);
        T* d = dst + (size_t)r * c_dst + dst_off + cg * kVec;
        if (accumulate) {
            float o[kVec];
            ld8(d, o);
#pragma unroll
            for (int k = 0; k < kVec; ++k) v[k] += o[k];
        }
        st8(d, v);
    }
}

template <typename T>
__global__ void zero_stuff2x_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c) {
    const int cv = c / kVec;
    const int64_t total = (int64_t)n * 2 * h * 2 * w * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int cg = t % cv; t /= cv;
        const int wo = t % (2 * w); t /= (2 * w);
        const int ho = t % (2 * h);
        const int b = t / (2 * h);
        float v[kVec];
#pragma unroll
        for (int k = 0; k < kVec; ++k) v[k] = 0.f;
        if (!(ho & 1) && !(wo & 1)) ld8(x + (((size_t)b * h + ho / 2) * w + wo / 2) * c + cg * kVec, v);
        st8(y + (((size_t)b * 2 * h + ho) * 2 * w + wo) * c + cg * kVec, v);
    }
}

// Pixel shuffle with block 2 (NHWC): y[n][2i+a][2j+b][k] = f(x[n][i][j][(2a+b)*c + k]), f = (+ bias[k]) (* scale[k] + shift[k]) (ReLU),
// each part optional.  A ConvTranspose2d(kernel 4, stride 2, padding 1) is a 3x3 convolution onto 4*c channels (one 2x2-tap
// sub-kernel per output parity, embedded in the 3x3 taps) followed by this shuffle (smp Linknet's TransposeX2).
template <typename T>
__global__ void depth_to_space2_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c, const float* __restrict__ bias,
                                       const float* __restrict__ scale, const float* __restrict__ shift, int relu) {
    const int cv = c / kVec;
    const int64_t total = (int64_t)n * 2 * h * 2 * w * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int cg = t % cv; t /= cv;
        const int wo = t % (2 * w); t /= (2 * w);
        const int ho = t % (2 * h);
        const int b = t / (2 * h);
        const int sub = (ho & 1) * 2 + (wo & 1);
        float v[kVec];
        ld8(x + (((size_t)b * h + (ho >> 1)) * w + (wo >> 1)) * 4 * c + sub * c + cg * kVec, v);
#pragma unroll
        for (int k = 0; k < kVec; ++k) {
            const int ch = cg * kVec + k;
            if (bias) v[k] += bias[ch];
            if (scale) v[k] = v[k] * scale[ch] + shift[ch];
            if (relu) v[k] = fmaxf(v[k], 0.f);
        }
        st8(y + i * kVec, v);
    }
}

// the inverse gather: y[n][i][j][(2a+b)*c + k] = x[n][2i+a][2j+b][k] (the shuffle's gradient)
template <typename T>
__global__ void space_to_depth2_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c) {
    const int cv = c / kVec;
    const int64_t total = (int64_t)n * h * w * 4 * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int cg = t % cv; t /= cv;
        const int sub = t % 4; t /= 4;
        const int wi = t % w; t /= w;
        const int hi = t % h;
        const int b = t / h;
        float v[kVec];
        ld8(x + (((size_t)b * 2 * h + 2 * hi + (sub >> 1)) * 2 * w + 2 * wi + (sub & 1)) * c + cg * kVec, v);
        st8(y + i * kVec, v);
    }
}

// column sums of x[rows][c] in two fixed-order stages (a bias gradient): partial[block][c], then out[c].  blockIdx.y = slab of
// up to 256 channels (cs of them, starting at 256 * blockIdx.y)
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int64_t rows, int c, int cs, float* __restrict__ partial) {
    // thread -> (row lane, 8-channel group): 256 / (slab width / 8) rows per sweep; blockIdx.y = slab of up to 256 channels (the last
    // one may be narrower: any channel count that is a multiple of 8)
    const int c0 = blockIdx.y * 256;
    cs = min(256, c - c0);
    const int cv = cs / kVec, rl = 256 / cv;
    const int cg = threadIdx.x % cv, r0 = threadIdx.x / cv;
    __shared__ float red[256 * kVec];
    float s[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) s[k] = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * rl + r0; r0 < rl && r < rows; r += (int64_t)gridDim.x * rl) {
        float v[kVec];
        ld8(x + (size_t)r * c + c0 + cg * kVec, v);
#pragma unroll
        for (int k = 0; k < kVec; ++k) s[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < kVec; ++k) red[threadIdx.x * kVec + k] = s[k];
    __syncthreads();
    if (threadIdx.x < cs) {
        const int g = threadIdx.x / kVec, k = threadIdx.x % kVec;
        float t = 0.f;
        for (int r = 0; r < rl; ++r) t += red[(r * cv + g) * kVec + k];
        partial[(size_t)blockIdx.x * c + c0 + threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(64) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblk, int c) {
    const int ch = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 64) s += partial[(size_t)b * c + ch];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[ch] = s;
}

inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

#define VS_DISPATCH_T(kern, total, ...)                                                                      \
    do {                                                                                                     \
        VS_FOR_T(dtype, \
            hipLaunchKernelGGL(kern<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__));  \
        VS_LAUNCH_CHECK();                                                                                   \
    } while (0)

extern "C" int vs_maxpool_fwd(int dtype, const void* x, void* y, uint8_t* idx, int n, int h, int w, int c, void* stream) {
    VS_REQUIRE(c % kVec == 0 && h % 2 == 0 && w % 2 == 0, "maxpool_fwd: bad shape");
    const int64_t total = (int64_t)n * (h / 2) * (w / 2) * (c / kVec);
    VS_FOR_T(dtype, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)x, (T*)y, idx, n, h, w, c, 0));
    VS_LAUNCH_CHECK();
    return VS_OK;
}

extern "C" int vs_maxpool_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int accumulate, int n, int h,
                              int w, int c, void* stream) {
    VS_REQUIRE(c % kVec == 0 && h % 2 == 0 && w % 2 == 0 && idx, "maxpool_bwd: bad arguments");
    const int64_t total = (int64_t)n * (h / 2) * (w / 2) * (c / kVec);
    VS_FOR_T(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)dy, idx, (T*)dx, accumulate, n, h, w, c));
    VS_LAUNCH_CHECK();
    return VS_OK;
}

int launch_upsample2x_bwd(int dtype, const void* dy, void* dx, int n, int h, int w, int c, int accumulate, hipStream_t stream) {
    VS_REQUIRE(c % kVec == 0, "upsample2x_bwd: channels must be a multiple of 8");
    const int64_t total = (int64_t)n * h * w * (c / kVec);
    VS_FOR_T(dtype, hipLaunchKernelGGL(upsample2x_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, stream,
                           (const T*)dy, (T*)dx, n, h, w, c, accumulate));
    VS_LAUNCH_CHECK();
    return VS_OK;
}

extern "C" int vs_upsample2x_bwd(int dtype, const void* dy, void* dx, int n, int h, int w, int c, void* stream) {
    return launch_upsample2x_bwd(dtype, dy, dx, n, h, w, c, 0, (hipStream_t)stream);
}

extern "C" int vs_channel_slice(int dtype, const void* src, int c_src, int src_off, void* dst, int c_dst, int dst_off, int c,
                                int64_t rows, int accumulate, void* stream) {
    VS_REQUIRE(src && dst && c > 0 && c % kVec == 0 && src_off % kVec == 0 && dst_off % kVec == 0 && c_src % kVec == 0 && c_dst % kVec == 0 &&
               src_off + c <= c_src && dst_off + c <= c_dst && rows >= 0,
               "channel_slice: channel counts / offsets must be multiples of 8 and inside their tensors");
    const int64_t total = rows * (c / kVec);
    VS_FOR_T(dtype, hipLaunchKernelGGL(channel_slice_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const T*)src, c_src,
                           src_off, (T*)dst, c_dst, dst_off, c, rows, accumulate));
    VS_LAUNCH_CHECK();
    return VS_OK;
}

extern "C" int vs_zero_stuff2x(int dtype, const void* x, void* y, int n, int h, int w, int c, void* stream) {
    VS_REQUIRE(c % kVec == 0, "zero_stuff2x: channels must be a multiple of 8");
    const int64_t total = (int64_t)n * 4 * h * w * (c / kVec);
    VS_FOR_T(dtype, hipLaunchKernelGGL(zero_stuff2x_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)x, (T*)y, n, h, w, c));
    VS_LAUNCH_CHECK();
    return VS_OK;
}

extern "C" int vs_depth_to_space2(int dtype, const void* x, void* y, int n, int h, int w, int c, const float* bias, const float* scale,
                                  const float* shift, int relu, void* stream) {
    VS_REQUIRE(x && y && c > 0 && c % kVec == 0 && (!scale == !shift), "depth_to_space2: channels must be a multiple of 8, scale and shift come together");
    const int64_t total = (int64_t)n * 4 * h * w * (c / kVec);
    VS_FOR_T(dtype, hipLaunchKernelGGL(depth_to_space2_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                           (T*)y, n, h, w, c, bias, scale, shift, relu));
    VS_LAUNCH_CHECK();
    return VS_OK;
}

extern "C" int vs_space_to_depth2(int dtype, const void* x, void* y, int n, int h, int w, int c, void* stream) {
    VS_REQUIRE(x && y && c > 0 && c % kVec == 0, "space_to_depth2: channels must be a multiple of 8");
    const int64_t total = (int64_t)n * 4 * h * w * (c / kVec);
    VS_FOR_T(dtype, hipLaunchKernelGGL(space_to_depth2_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                           (T*)y, n, h, w, c));
    VS_LAUNCH_CHECK();
    return VS_OK;
}

// out[c] = sum over rows of x[rows][c] (fp32, fixed summation order); workspace: vs_colsum_workspace(c) bytes
constexpr int kColsumBlocks = 512;
extern "C" size_t vs_colsum_workspace(int c) { return (size_t)kColsumBlocks * c * sizeof(float); }
extern "C" int vs_colsum(int dtype, const void* x, int64_t rows, int c, float* out, float* workspace, size_t workspace_bytes, void* stream) {
    const int cs = c < 256 ? c : 256;
    VS_REQUIRE(x && out && workspace && c >= kVec && c % kVec == 0, "colsum: channels must be a multiple of 8 (got %d)", c);
    VS_REQUIRE(workspace_bytes >= vs_colsum_workspace(c), "colsum: workspace too small");
    const int rl = 256 / (cs / kVec);
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(kColsumBlocks, (rows + rl - 1) / rl));
    const dim3 grid(blocks, (c + 255) / 256);
    VS_FOR_T(dtype, hipLaunchKernelGGL(colsum_partial_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, rows, c, cs, workspace));
    VS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3(c), dim3(64), 0, (hipStream_t)stream, workspace, out, blocks, c);
    VS_LAUNCH_CHECK();
    return VS_OK;
}
