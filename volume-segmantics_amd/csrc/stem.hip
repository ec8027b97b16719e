// ResNet stem: 7x7 stride-2 pad-3 convolution with ONE input channel (smp patches encoder.conv1 to
// in_channels=1) - forward and weight gradient.  K = 49 is too ragged for the
// 32-channel-chunk implicit GEMM, so both directions run on the exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32: one scalar per lane, so the im2col gather is just a per-lane LDS address):
//   forward : D[cout][pixel] += W[cout][tap] * X[tap][pixel],   k = taps (49 padded to 52)
//   wgrad   : D[cout][tap]   += dY[pixel][cout] * X[pixel][tap], k = output pixels
// The input image stays fp32 (it is the caller's (B,1,H,W) tensor); the output is written in the
// network's activation dtype.  No dgrad: the input image needs no gradient.
#include <algorithm>

#include "common.h"
#include "prof.h"

namespace {

constexpr int TPH = 8, TPW = 32;              // output pixel tile
constexpr int PH = 2 * TPH + 5, PW = 2 * TPW + 5;  // input patch (21 x 69)
constexpr int WS = 53;                        // LDS row stride (floats) of the [64][52] weight image

__device__ __forceinline__ uint2 ds_read_tr16_stem(const char* p) {
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(p));
    return __builtin_bit_cast(uint2, v);
}

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

// ---- bf16 path -------------------------------------------------------------------------------------------------------
// With bf16 activations the stem runs on v_mfma_f32_16x16x32_bf16 as well: K is laid out as kh * 8 + kw (7 x 7 taps
// padded to 8 x 8 = 64 = two MFMA k-steps), so the eight consecutive k of a lane are the eight consecutive input columns
// 2*px .. 2*px+7 of ONE patch row - four aligned dword LDS reads, no im2col.  The weights (64 x 64 bf16, zero in the
// padding taps) sit in registers for the whole workgroup; the input patch is rounded to bf16 while it is staged.
constexpr int PWB = 72, PHB = PH + 1;          // bf16 patch: 22 rows (one spare for the zero-weight kh = 7) x 72 columns

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) { return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16); }

__device__ __forceinline__ void stage_patch_bf16(const float* __restrict__ x, bf16_t* patch, int img, int h, int wd, int h0, int w0, int tid) {
    for (int i = tid; i < PHB * PWB / 2; i += 256) {       // pairs of columns
        const int ph = i / (PWB / 2), pw = (i - ph * (PWB / 2)) * 2;
        const int hi = 2 * h0 - 3 + ph, wi = 2 * w0 - 3 + pw;
        float a = 0.f, b = 0.f;
        if (hi >= 0 && hi < h) {
            const float* row = x + ((size_t)img * h + hi) * wd;
            if (wi >= 0 && wi < wd) a = row[wi];
            if (wi + 1 >= 0 && wi + 1 < wd) b = row[wi + 1];
        }
        reinterpret_cast<uint32_t*>(patch)[i] = pack_bf16(a, b);
    }
}

// the same patch in two halves - loads into registers (issued early: the previous tile's epilogue runs under them), then the rounded
// pairs into LDS
constexpr int kStemPF = (PHB * PWB / 2 + 255) / 256;      // column pairs per thread
__device__ __forceinline__ void patch_load_regs(const float* __restrict__ x, int img, int h, int wd, int h0, int w0, int tid, float2 (&r)[kStemPF]) {
#pragma unroll
    for (int k = 0; k < kStemPF; ++k) {
        const int i = tid + k * 256;
        const int ph = i / (PWB / 2), pw = (i - ph * (PWB / 2)) * 2;
        const int hi = 2 * h0 - 3 + ph, wi = 2 * w0 - 3 + pw;
        float a = 0.f, b = 0.f;
        if (i < PHB * PWB / 2 && hi >= 0 && hi < h) {
            const float* row = x + ((size_t)img * h + hi) * wd;
            if (wi >= 0 && wi < wd) a = row[wi];
            if (wi + 1 >= 0 && wi + 1 < wd) b = row[wi + 1];
        }
        r[k] = make_float2(a, b);
    }
}
__device__ __forceinline__ void patch_store_regs(bf16_t* patch, int tid, const float2 (&r)[kStemPF]) {
#pragma unroll
    for (int k = 0; k < kStemPF; ++k) {
        const int i = tid + k * 256;
        if (i < PHB * PWB / 2) reinterpret_cast<uint32_t*>(patch)[i] = pack_bf16(r[k].x, r[k].y);
    }
}

// this lane's weight fragments: A[row = cout 16j + lr][k = 8 * (4s + lq) + kw]
__device__ __forceinline__ void load_stem_weights_bf16(const float* __restrict__ w, int lr, int lq, uint4 (&wa)[4][2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int kh = 4 * s + lq;
            float v[8];
#pragma unroll
            for (int kw = 0; kw < 8; ++kw) v[kw] = (kh < 7 && kw < 7) ? w[(j * 16 + lr) * 49 + kh * 7 + kw] : 0.f;
            wa[j][s] = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
        }
}

// bins (training, optional): [nb rows][2][64] 64-bit fixed-point sums of the fp32 accumulators (sum x * 2^24, sum x^2 * 2^16), as the
// convolution kernels leave them for their BatchNorm (ConvParams::stats_bins) - the workgroup's tiles are summed in registers and
// added once, so the separate statistics sweep over the stored tensor (67 MB at batch 32 of 256 x 256) does not run.
__global__ __launch_bounds__(256) void stem_fwd_bf16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            int relu, bf16_t* __restrict__ y, int n, int h, int wd,
                                                            unsigned long long* __restrict__ bins, int nb) {
    __shared__ __attribute__((aligned(16))) bf16_t patch[PHB * PWB];
    __shared__ __attribute__((aligned(16))) char otile[256 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
    const int ho_n = h / 2, wo_n = wd / 2;
    const int tiles_w = (wo_n + TPW - 1) / TPW, tiles_h = (ho_n + TPH - 1) / TPH;
    uint4 wa[4][2];
    load_stem_weights_bf16(w, lr, lq, wa);      // once per workgroup: it walks over tiles blockIdx.x, + gridDim.x, ..
    float s1[4][4], s2[4][4];                   // [cout fragment j][r]: sums over this lane's pixels of every tile
    float sc[4][4], sh[4][4];                   // the folded-BatchNorm constants of this lane's 16 couts: once per workgroup, not per tile
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s1[j][r] = s2[j][r] = 0.f;
            sc[j][r] = scale ? scale[j * 16 + lq * 4 + r] : 1.f;
            sh[j][r] = scale ? shift[j * 16 + lq * 4 + r] : 0.f;
        }
    const int ntiles = n * tiles_h * tiles_w;
    auto tile_origin = [&](int tile, int& img, int& h0, int& w0) {
        int b = tile;
        const int tx = b % tiles_w; b /= tiles_w;
        const int ty = b % tiles_h;
        img = b / tiles_h; h0 = ty * TPH; w0 = tx * TPW;
    };
    float2 pf[kStemPF];                          // the NEXT tile's patch: in flight while this tile's MFMAs, transpose and stores run
    {
        int img, h0, w0;
        tile_origin(min((int)blockIdx.x, ntiles - 1), img, h0, w0);
        patch_load_regs(x, img, h, wd, h0, w0, tid, pf);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int img, h0, w0;
    tile_origin(tile, img, h0, w0);
    __syncthreads();                            // the previous tile's fragment reads are done
    patch_store_regs(patch, tid, pf);
    if (tile + (int)gridDim.x < ntiles) {
        int img2, h2, w2;
        tile_origin(tile + gridDim.x, img2, h2, w2);
        patch_load_regs(x, img2, h, wd, h2, w2, tid, pf);
    }
    __syncthreads();
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {                              // pixel tile i: tile row 2*wave + (i >> 1), columns (i & 1) * 16 + lr
        const int py = 2 * wave + (i >> 1), px = (i & 1) * 16 + lr;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(patch + (2 * py + 4 * s + lq) * PWB + 2 * px);
            const uint4 xb = make_uint4(src[0], src[1], src[2], src[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[j][s]), __builtin_bit_cast(bf16x8, xb),
                                                                    acc[i][j], 0, 0, 0);
        }
    }
    if (bins) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = h0 + 2 * wave + (i >> 1) < ho_n && w0 + (i & 1) * 16 + lr < wo_n;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = ok ? acc[i][j][r] : 0.f;
                    s1[j][r] += v; s2[j][r] += v * v;
                }
        }
    }
    // Output through LDS: the accumulators hold 8-byte pieces (4 couts of one pixel); transposed through a [256 pixels][128 B]
    // tile every lane stores 16 contiguous bytes and a wave writes whole 128-byte pixel rows.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pl = (2 * wave + (i >> 1)) * TPW + (i & 1) * 16 + lr;     // pixel index inside the 8 x 32 tile
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = j * 16 + lq * 4;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (scale) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[j][r] + sh[j][r];
            }
            if (relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            // 16-byte slot (c >> 3) of the pixel's row, XOR-ed with the pixel's low bits so the 8-byte writes spread over banks
            *reinterpret_cast<uint2*>(otile + pl * 128 + ((((c >> 3) ^ (pl & 7)) << 4) | ((c & 4) << 1))) =
                make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {                              // 256 pixels x 8 slots of 16 bytes
        const int item = tid + k * 256;
        const int pl = item >> 3, slot = item & 7;
        const int ho = h0 + (pl >> 5), wo = w0 + (pl & 31);
        if (ho < ho_n && wo < wo_n)
            *reinterpret_cast<uint4*>(y + (((size_t)img * ho_n + ho) * wo_n + wo) * 64 + slot * 8) =
                *reinterpret_cast<const uint4*>(otile + pl * 128 + ((slot ^ (pl & 7)) << 4));
    }
    }
    if (bins) {     // the workgroup's sums: lanes of a 16-lane row, then the four waves (LDS), then one fixed-point add per channel and sum
        float* red = reinterpret_cast<float*>(otile);            // [4 waves][2][64]
        __syncthreads();                                         // the last tile's transposed reads are done
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s1[j][r] += __shfl_xor(s1[j][r], o, 64); s2[j][r] += __shfl_xor(s2[j][r], o, 64); }
                if (lr == 0) {
                    red[(wave * 2 + 0) * 64 + j * 16 + lq * 4 + r] = s1[j][r];
                    red[(wave * 2 + 1) * 64 + j * 16 + lq * 4 + r] = s2[j][r];
                }
            }
        __syncthreads();
        if (tid < 128) {
            const int k = tid >> 6, cc = tid & 63;
            const float a = (red[(0 * 2 + k) * 64 + cc] + red[(1 * 2 + k) * 64 + cc]) + (red[(2 * 2 + k) * 64 + cc] + red[(3 * 2 + k) * 64 + cc]);
            atomicAdd(bins + ((size_t)(blockIdx.x & (nb - 1)) * 2 + k) * 64 + cc,
                      (unsigned long long)__double2ll_rn((double)a * (k ? kStatScale2 : kStatScale1)));
        }
    }
}

// Weight gradient, bf16: D[cout][kh*8+kw] += dY^T[cout][pixel] * X[pixel][kh*8+kw], k = 32 output pixels (one tile row).
// A: transposed reads of the [pixel][64 cout] dy tile (32-byte slices XOR-swizzled as in conv_wgrad); B: the eight pixels of
// a lane are eight stride-2 columns of one patch row (ds_read_u16 gathers).  Wave w takes tile rows w and w + 4, all
// 4 x 4 (cout x tap) fragments; the four waves meet in LDS once per workgroup.
__global__ __launch_bounds__(256) void stem_wgrad_bf16_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dy,
                                                              float* __restrict__ partial, int n, int h, int wd,
                                                              int total_tiles, int tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* patch = reinterpret_cast<bf16_t*>(smem);            // PHB * PWB bf16
    char* dyl = smem + ((PHB * PWB * 2 + 255) & ~255);          // [256 pixels][128 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
    const int ho_n = h / 2, wo_n = wd / 2;
    const int tiles_w = (wo_n + TPW - 1) / TPW, tiles_h = (ho_n + TPH - 1) / TPH;
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(dy, n * ho_n * wo_n * 64 * 2);
    f32x4 acc[4][4];                                             // [cout fragment][tap fragment]
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B gather: tap t = 16 j + lr -> (kh, kw) = (t >> 3, t & 7); this lane's pixels are columns 8 lq .. 8 lq + 7 of the tile row
    int goff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = 16 * j + lr;
        goff[j] = ((t >> 3) * PWB + (t & 7) + 16 * lq) * 2;      // bytes; + 2 * tile row * PWB * 2, + 4 * pixel
    }
    // A (dy^T) fragment addresses: pixels 8 lq + (lr >> 2) (+4) of the tile row, slice m ^ g(pixel)
    int a_addr[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int px = 8 * lq + (lr >> 2) + 4 * hh;
        a_addr[hh] = px * 128 + ((((px >> 1) & 1) | (((px >> 3) & 1) << 1)) << 5) + (lr & 3) * 8;
    }
    const int t0 = blockIdx.x * tiles_per_block, t1 = min(total_tiles, t0 + tiles_per_block);
    for (int tile = t0; tile < t1; ++tile) {
        int b = tile;
        const int tx = b % tiles_w; b /= tiles_w;
        const int ty = b % tiles_h;
        const int img = b / tiles_h;
        const int h0 = ty * TPH, w0 = tx * TPW;
        __syncthreads();
        stage_patch_bf16(x, patch, img, h, wd, h0, w0, tid);
#pragma unroll
        for (int i = 0; i < 8; ++i) {                            // 256 pixels x 8 segments of 16 bytes
            const int item = tid + i * 256;
            const int pl = item >> 3, seg = item & 7;
            const int ho = h0 + (pl >> 5), wo = w0 + (pl & 31);
            const bool ok = ho < ho_n && wo < wo_n;
            const uint4 v = bload(rd, ok ? ((((img * ho_n + ho) * wo_n + wo) * 64 + seg * 8) * 2) : -1, 0);
            const int gk = ((pl >> 1) & 1) | (((pl >> 3) & 1) << 1);
            *reinterpret_cast<uint4*>(dyl + pl * 128 + (((seg >> 1) ^ gk) << 5) + (seg & 1) * 16) = v;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int trow = wave + 4 * rr;                      // k-step = tile row trow (32 pixels)
            uint4 af[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint2 lo = ds_read_tr16_stem(dyl + trow * 32 * 128 + (a_addr[0] ^ (m << 5)));
                const uint2 hi = ds_read_tr16_stem(dyl + trow * 32 * 128 + (a_addr[1] ^ (m << 5)));
                af[m] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* base = reinterpret_cast<const char*>(patch) + goff[j] + 2 * trow * PWB * 2;
                uint32_t d[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t lo = *reinterpret_cast<const bf16_t*>(base + (2 * k) * 4);
                    const uint32_t hi = *reinterpret_cast<const bf16_t*>(base + (2 * k + 1) * 4);
                    d[k] = lo | (hi << 16);
                }
                const uint4 bf = make_uint4(d[0], d[1], d[2], d[3]);
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[m]), __builtin_bit_cast(bf16x8, bf),
                                                                        acc[m][j], 0, 0, 0);
            }
        }
    }
    // the four waves (different pixels, same outputs) meet in LDS; fixed order
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    for (int round = 1; round < 4; ++round) {
        __syncthreads();
        if (wave == round) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) red[(m * 4 + j) * 64 + lane] = acc[m][j];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[m][j] += red[(m * 4 + j) * 64 + lane];
        }
    }
    if (wave != 0) return;
    float* out = partial + (size_t)blockIdx.x * 64 * 49;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = 16 * j + lr, kh = t >> 3, kw = t & 7;
        if (kh >= 7 || kw >= 7) continue;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(m * 16 + lq * 4 + r) * 49 + kh * 7 + kw] = acc[m][j][r];
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
    if (dtype == VS_BF16 && vs_option("stem_bf16"))
        hipLaunchKernelGGL(stem_fwd_bf16_kernel, dim3(std::min(tiles, 1024)), dim3(256), 0, (hipStream_t)stream, x, w, scale, shift,
                           relu, (bf16_t*)y, n, h, w_, (unsigned long long*)nullptr, 0);
    else if (dtype == VS_BF16)
        hipLaunchKernelGGL(stem_fwd_kernel<bf16_t>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, relu,
                           (bf16_t*)y, n, h, w_);
    else if (dtype == VS_F16)       // fp32 arithmetic on the fp32 slices (one input channel: 0.5 % of the network's FLOPs), fp16 store
        hipLaunchKernelGGL(stem_fwd_kernel<f16_t>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, relu,
                           (f16_t*)y, n, h, w_);
    else
        hipLaunchKernelGGL(stem_fwd_kernel<float>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, relu,
                           (float*)y, n, h, w_);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

// training forward of the bf16 stem with the batch statistics' sums left in fixed-point bins ([nb][2][64], zeroed by the caller; nb a
// power of two); false: this dtype / option set has no such kernel (the caller runs its statistics sweep)
bool stem_fwd_bins_ok(int dtype) { return dtype == VS_BF16 && vs_option("stem_bf16"); }
int launch_stem_fwd_bins(const float* x, const float* w, void* z, int n, int h, int w_, unsigned long long* bins, int nb, hipStream_t s) {
    VS_REQUIRE(h % 2 == 0 && w_ % 2 == 0 && x && w && z && bins && nb >= 1 && !(nb & (nb - 1)), "stem_fwd_bins: bad arguments");
    const int tiles = n * cdiv(h / 2, TPH) * cdiv(w_ / 2, TPW);
    hipLaunchKernelGGL(stem_fwd_bf16_kernel, dim3(std::min(tiles, 1024)), dim3(256), 0, s, x, w, (const float*)nullptr, (const float*)nullptr, 0,
                       (bf16_t*)z, n, h, w_, bins, nb);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

extern "C" size_t vs_stem_wgrad_workspace(int n, int h, int w_) {
    (void)n; (void)h; (void)w_;
    return (size_t)kStemBlocks * 64 * 49 * sizeof(float);
}

extern "C" int vs_stem_wgrad(int dtype, const float* x, const void* dy, float* dw, float* workspace,
                             size_t workspace_bytes, int n, int h, int w_, void* stream) {
    VS_NO_F16(dtype, "stem_wgrad");
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
    if (dtype == VS_BF16 && vs_option("stem_bf16")) {
        VS_REQUIRE((double)n * (h / 2) * (w_ / 2) * 128.0 < 2.0e9, "stem_wgrad: dy exceeds the 32-bit staging offsets");
        const int per2 = cdiv(total, 512), blocks2 = cdiv(total, per2);   // fewer, longer blocks: the weights stay in registers
        const size_t lds2 = ((PHB * PWB * 2 + 255) & ~255) + 256 * 128;
        hipLaunchKernelGGL(stem_wgrad_bf16_kernel, dim3(blocks2), dim3(256), lds2, (hipStream_t)stream, x, (const bf16_t*)dy,
                           workspace, n, h, w_, total, per2);
        VS_LAUNCH_CHECK();
        return launch_slab_reduce(workspace, dw, 64 * 49, blocks2, (hipStream_t)stream);
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
