// ResNet stem: 7x7 stride-2 pad-3 convolution with ONE input channel (smp patches encoder.conv1 to
// in_channels=1) - forward and weight gradient.  K = 49 is too ragged for the
// 32-channel-chunk implicit GEMM, so both directions run on the exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32: one scalar per lane, so the im2col gather is just a per-lane LDS address):
//   forward : D[cout][pixel] += W[cout][tap] * X[tap][pixel],   k = taps (49 padded to 52)
//   wgrad   : D[cout][tap]   += dY[pixel][cout] * X[pixel][tap], k = output pixels
// The input image stays fp32 (it is the caller's (B,1,H,W) tensor); the output is written in the
// network's activation dtype.  No dgrad: the input image needs no gradient.
#include "common.h"

namespace {

constexpr int TPH = 8, TPW = 32;              // output pixel tile
constexpr int PH = 2 * TPH + 5, PW = 2 * TPW + 5;  // input patch (21 x 69)
constexpr int WS = 53;                        // LDS row stride (floats) of the [64][52] weight image

template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     int relu, T* __restrict__ y, int n, int h, int wd) {
    __shared__ float patch[PH * PW];
    __shared__ float wl[64 * WS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
    const int ho_n = h / 2, wo_n = wd / 2;
    const int tiles_w = (wo_n + TPW - 1) / TPW, tiles_h = (ho_n + TPH - 1) / TPH;
    int b = blockIdx.x;
    const int tx = b % tiles_w; b /= tiles_w;
    const int ty = b % tiles_h;
    const int img = b / tiles_h;
    const int h0 = ty * TPH, w0 = tx * TPW;
    for (int i = tid; i < PH * PW; i += 256) {
        const int ph = i / PW, pw = i % PW;
        const int hi = 2 * h0 - 3 + ph, wi = 2 * w0 - 3 + pw;
        patch[i] = (hi >= 0 && hi < h && wi >= 0 && wi < wd) ? x[((size_t)img * h + hi) * wd + wi] : 0.f;
    }
    for (int i = tid; i < 64 * 52; i += 256) {
        const int co = i / 52, t = i % 52;
        wl[co * WS + t] = t < 49 ? w[co * 49 + t] : 0.f;
    }
    __syncthreads();
    // wave handles pixel rows 2*wave, 2*wave+1 of the tile -> 4 pixel tiles of 16
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int pbase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int py = 2 * wave + (i >> 1), px = (i & 1) * 16 + lr;
        pbase[i] = (2 * py) * PW + 2 * px;
    }
#pragma unroll
    for (int s = 0; s < 13; ++s) {
        const int t = 4 * s + lq;              // this lane's tap (k index)
        const int tc = t < 49 ? t : 48;        // clamp: weight is zero for the padding taps
        const int toff = (tc / 7) * PW + (tc % 7);
        float a[4], bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = wl[(j * 16 + lr) * WS + t];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = patch[pbase[i] + toff];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bv[i], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ho = h0 + 2 * wave + (i >> 1), wo = w0 + (i & 1) * 16 + lr;
        if (ho >= ho_n || wo >= wo_n) continue;
        T* o = y + (((size_t)img * ho_n + ho) * wo_n + wo) * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = j * 16 + lq * 4;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (scale) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] * scale[c + r] + shift[c + r];
            }
            if (relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            st4(o + c, make_float4(v[0], v[1], v[2], v[3]));
        }
    }
}

// wgrad: each workgroup walks a range of 8x32 output tiles; wave w owns cout tile w (16 couts) x 4 tap tiles.
template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                       float* __restrict__ partial, int n, int h, int wd,
                                                       int total_tiles, int tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* patch = reinterpret_cast<float*>(smem);          // PH*PW
    float* dyl = patch + ((PH * PW + 3) & ~3);              // [256 pixels][65] fp32
    constexpr int DS = 65;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
    const int ho_n = h / 2, wo_n = wd / 2;
    const int tiles_w = (wo_n + TPW - 1) / TPW, tiles_h = (ho_n + TPH - 1) / TPH;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int toff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = j * 16 + lr;
        const int tc = t < 49 ? t : 48;
        toff[j] = (tc / 7) * PW + (tc % 7);
    }
    const int t0 = blockIdx.x * tiles_per_block, t1 = min(total_tiles, t0 + tiles_per_block);
    for (int tile = t0; tile < t1; ++tile) {
        int b = tile;
        const int tx = b % tiles_w; b /= tiles_w;
        const int ty = b % tiles_h;
        const int img = b / tiles_h;
        const int h0 = ty * TPH, w0 = tx * TPW;
        __syncthreads();
        for (int i = tid; i < PH * PW; i += 256) {
            const int ph = i / PW, pw = i % PW;
            const int hi = 2 * h0 - 3 + ph, wi = 2 * w0 - 3 + pw;
            patch[i] = (hi >= 0 && hi < h && wi >= 0 && wi < wd) ? x[((size_t)img * h + hi) * wd + wi] : 0.f;
        }
        for (int i = tid; i < 256 * 8; i += 256) {  // 256 pixels x 8 segments of 8 channels
            const int pl = i >> 3, seg = i & 7;
            const int ho = h0 + (pl >> 5), wo = w0 + (pl & 31);
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
            if (ho < ho_n && wo < wo_n) ld8(dy + (((size_t)img * ho_n + ho) * wo_n + wo) * 64 + seg * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) dyl[pl * DS + seg * 8 + k] = v[k];
        }
        __syncthreads();
        for (int s = 0; s < 64; ++s) {
            const int pk = 4 * s + lq;                       // this lane's pixel (k index)
            const float a = dyl[pk * DS + wave * 16 + lr];   // A[row = cout][k = pixel]
            const int pb = (2 * (pk >> 5)) * PW + 2 * (pk & 31);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, patch[pb + toff[j]], acc[j], 0, 0, 0);
        }
    }
    float* out = partial + (size_t)blockIdx.x * 64 * 49;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = j * 16 + lr;
        if (t >= 49) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(wave * 16 + lq * 4 + r) * 49 + t] = acc[j][r];
    }
}

__global__ void stem_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int nparts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 49) return;
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) s += partial[(size_t)k * 64 * 49 + i];
    dw[i] = s;
}

constexpr int kStemBlocks = 1024;

}  // namespace

extern "C" int vs_stem_fwd(int dtype, const float* x, const float* w, const float* scale, const float* shift, int relu,
                           void* y, int n, int h, int w_, void* stream) {
    VS_REQUIRE(h % 2 == 0 && w_ % 2 == 0 && x && w && y, "stem_fwd: bad arguments");
    const int tiles = n * cdiv(h / 2, TPH) * cdiv(w_ / 2, TPW);
    if (dtype == VS_BF16)
        hipLaunchKernelGGL(stem_fwd_kernel<bf16_t>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, relu,
                           (bf16_t*)y, n, h, w_);
    else
        hipLaunchKernelGGL(stem_fwd_kernel<float>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, relu,
                           (float*)y, n, h, w_);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

extern "C" size_t vs_stem_wgrad_workspace(int n, int h, int w_) {
    (void)n; (void)h; (void)w_;
    return (size_t)kStemBlocks * 64 * 49 * sizeof(float);
}

extern "C" int vs_stem_wgrad(int dtype, const float* x, const void* dy, float* dw, float* workspace,
                             size_t workspace_bytes, int n, int h, int w_, void* stream) {
    VS_REQUIRE(h % 2 == 0 && w_ % 2 == 0 && x && dy && dw, "stem_wgrad: bad arguments");
    VS_REQUIRE(workspace && workspace_bytes >= vs_stem_wgrad_workspace(n, h, w_), "stem_wgrad: workspace too small");
    const int total = n * cdiv(h / 2, TPH) * cdiv(w_ / 2, TPW);
    const int per = cdiv(total, kStemBlocks);
    const int blocks = cdiv(total, per);
    const size_t lds = (((PH * PW + 3) & ~3) + 256 * 65) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void*)stem_wgrad_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VS_CHECK_HIP(hipFuncSetAttribute((const void*)stem_wgrad_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    if (dtype == VS_BF16)
        hipLaunchKernelGGL(stem_wgrad_kernel<bf16_t>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, x, (const bf16_t*)dy,
                           workspace, n, h, w_, total, per);
    else
        hipLaunchKernelGGL(stem_wgrad_kernel<float>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, x, (const float*)dy,
                           workspace, n, h, w_, total, per);
    VS_LAUNCH_CHECK();
    return launch_slab_reduce(workspace, dw, 64 * 49, blocks, (hipStream_t)stream);
}
