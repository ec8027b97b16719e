// Convolution weight gradient on MFMA (gfx950): dw[co][tap][ci] = sum_pixels dy[p][co] * x[p (+) tap][ci].
//
// GEMM view per tap: D[cout][cin] with K = output pixels.  NHWC keeps channels contiguous, so both
// operands need "8 consecutive pixels of one channel" per lane - the transposed fragment.  bf16 gets it
// for free from ds_read_b64_tr_b16 on the naturally laid out [pixel][channel] LDS tiles; f32 uses
// v_mfma_f32_16x16x4_f32 whose fragments are one scalar per lane (plain ds_read_b32).
// The staged input patch is the same virtual tensor cat(upsample(src0), src1) the forward conv read.
// Split-K over pixel tiles: every (workgroup, K-wave) writes an fp32 partial slab, a second kernel
// sums the slabs in a fixed order (bitwise reproducible, no float atomics).
//
// Replaces the conv weight gradients of loss.backward() (vol_seg_2d_trainer.py:429).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "prof.h"

namespace {

constexpr int kXS = 80;  // LDS bytes per staged input pixel (64 data + 16 pad)

template <typename T> struct WT;
template <> struct WT<bf16_t> { static constexpr int CK = 32, EPS = 8, NCI = 2, KSTEP = 32; };
template <> struct WT<float> { static constexpr int CK = 16, EPS = 4, NCI = 1, KSTEP = 4; };

struct WGeom {
    int tw_shift, TH, tiles_h, tiles_w, PH, PW, PT;
    int ctiles, cchunks, nsplit, total_tiles, dys;  // dys: LDS bytes per dy pixel row
};

__device__ __forceinline__ uint2 ds_read_tr16(const char* p) {
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(p));
    return __builtin_bit_cast(uint2, v);
}

constexpr int wg_patch_items(int pt, int stride) { return stride == 2 ? 5 : (pt == 2 ? 3 : 2); }

template <typename T, int WO, int NTAPS, int STRIDE, int PT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p, WGeom g) {
    constexpr int CK = WT<T>::CK, EPS = WT<T>::EPS, NCI = WT<T>::NCI, KSTEP = WT<T>::KSTEP;
    constexpr int WK = 4 / WO, BNO = 16 * WO, KW = NTAPS == 9 ? 3 : 1;
    constexpr int BM = 64 * PT, SEGS = BNO / EPS;      // SEGS: 16-byte segments per dy pixel row
    constexpr int PITEMS = wg_patch_items(PT, STRIDE), DTOTAL = BM * SEGS, DITEMS = (DTOTAL + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int wo = wave % WO, wk = wave / WO;
    const int TW = 1 << g.tw_shift;
    const int Cin = p.C0 + p.C1;
    const int P = g.PH * g.PW;
    char* patch = smem;
    char* dyl = smem + P * kXS;

    const int ct = blockIdx.x / g.cchunks, cc = blockIdx.x % g.cchunks;
    const int co0 = ct * BNO, c0 = cc * CK;
    const int split = blockIdx.y;
    const int per = (g.total_tiles + g.nsplit - 1) / g.nsplit;
    const int t0 = split * per, t1 = min(g.total_tiles, t0 + per);
    const int H0 = p.Hin >> p.up0, W0 = p.Win >> p.up0;
    const bool from0 = c0 < p.C0;
    const T* src = from0 ? (const T*)p.src0 : (const T*)p.src1;
    const int cs = from0 ? p.C0 : p.C1;
    const int cb = from0 ? c0 : c0 - p.C0;
    const int sh = from0 ? p.up0 : 0;
    const int Hs = from0 ? H0 : p.Hin, Ws = from0 ? W0 : p.Win;

    // tile-invariant staging coordinates
    int pph[PITEMS], ppw[PITEMS], pdst[PITEMS];
    bool pcv[PITEMS];
#pragma unroll
    for (int i = 0; i < PITEMS; ++i) {
        const int item = tid + i * 256;
        const int pp = item >> 2, seg = item & 3;
        pph[i] = pp / g.PW; ppw[i] = pp - pph[i] * g.PW;
        pdst[i] = item < P * 4 ? pp * kXS + seg * 16 : -1;
        pcv[i] = item < P * 4 && cb + seg * EPS < cs;
    }
    int dpl[DITEMS], dseg[DITEMS];
#pragma unroll
    for (int i = 0; i < DITEMS; ++i) {
        const int item = tid + i * 256;
        dpl[i] = item < DTOTAL ? item / SEGS : -1;
        dseg[i] = item % SEGS;
    }
    uint4 preg[PITEMS], dreg[DITEMS];
    auto load_tile = [&](int tile) {
        int b = tile;
        const int tx = b % g.tiles_w; b /= g.tiles_w;
        const int ty = b % g.tiles_h;
        const int n = b / g.tiles_h;
        const int h0 = ty * g.TH, w0 = tx * TW;
        const int hbase = h0 * STRIDE - p.pad, wbase = w0 * STRIDE - p.pad;
#pragma unroll
        for (int i = 0; i < PITEMS; ++i) {
            const int hi = hbase + pph[i], wi = wbase + ppw[i];
            const int seg = (tid + i * 256) & 3;
            preg[i] = make_uint4(0, 0, 0, 0);
            if (pcv[i] && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win)
                preg[i] = *reinterpret_cast<const uint4*>(src + (((size_t)n * Hs + (hi >> sh)) * Ws + (wi >> sh)) * cs + cb + seg * EPS);
        }
#pragma unroll
        for (int i = 0; i < DITEMS; ++i) {
            dreg[i] = make_uint4(0, 0, 0, 0);
            if (dpl[i] < 0) continue;
            const int ho = h0 + (dpl[i] >> g.tw_shift), wo_ = w0 + (dpl[i] & (TW - 1));
            const int co = co0 + dseg[i] * EPS;
            if (ho < p.Hout && wo_ < p.Wout && co < p.Cout) {
                const T* sp = (const T*)p.dy + (((size_t)n * p.Hout + ho) * p.Wout + wo_) * p.Cout + co;
                if (co + EPS <= p.Cout) dreg[i] = *reinterpret_cast<const uint4*>(sp);
                else {  // ragged channel tail
                    alignas(16) T tmp[EPS];
                    for (int e = 0; e < EPS; ++e) tmp[e] = (co + e < p.Cout) ? sp[e] : (T)0;
                    dreg[i] = *reinterpret_cast<const uint4*>(tmp);
                }
            }
        }
    };

    f32x4 acc[NTAPS][NCI];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int c = 0; c < NCI; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (t0 < t1) load_tile(t0);
    for (int tile = t0; tile < t1; ++tile) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PITEMS; ++i)
            if (pdst[i] >= 0) *reinterpret_cast<uint4*>(patch + pdst[i]) = preg[i];
#pragma unroll
        for (int i = 0; i < DITEMS; ++i)
            if (dpl[i] >= 0) *reinterpret_cast<uint4*>(dyl + dpl[i] * g.dys + dseg[i] * 16) = dreg[i];
        __syncthreads();
        if (tile + 1 < t1) load_tile(tile + 1);  // in flight while the MFMAs below run

        for (int ks = wk; ks < BM / KSTEP; ks += WK) {
            if constexpr (sizeof(T) == 2) {
                // pixel rows this lane addresses for the two transposed reads: 8*lq + (lr>>2) (+4)
                const int pa = ks * 32 + 8 * lq + (lr >> 2), pb = pa + 4;
                const int coff = (lr & 3) * 8;  // 4 channels * 2 B
                uint4 af;
                {
                    const uint2 lo = ds_read_tr16(dyl + pa * g.dys + wo * 32 + coff);
                    const uint2 hi = ds_read_tr16(dyl + pb * g.dys + wo * 32 + coff);
                    af = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
                const int xa = (((pa >> g.tw_shift) * STRIDE) * g.PW + (pa & (TW - 1)) * STRIDE) * kXS + coff;
                const int xb = (((pb >> g.tw_shift) * STRIDE) * g.PW + (pb & (TW - 1)) * STRIDE) * kXS + coff;
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    const int toff = ((t / KW) * g.PW + (t % KW)) * kXS;
#pragma unroll
                    for (int c = 0; c < NCI; ++c) {
                        const uint2 lo = ds_read_tr16(patch + xa + toff + c * 32);
                        const uint2 hi = ds_read_tr16(patch + xb + toff + c * 32);
                        const uint4 bf = make_uint4(lo.x, lo.y, hi.x, hi.y);
                        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc[t][c], 0, 0, 0);
                    }
                }
            } else {
                const int pk = ks * 4 + lq;
                const float a = *reinterpret_cast<const float*>(dyl + pk * g.dys + (wo * 16 + lr) * 4);
                const int xo = (((pk >> g.tw_shift) * STRIDE) * g.PW + (pk & (TW - 1)) * STRIDE) * kXS + lr * 4;
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    const float bv = *reinterpret_cast<const float*>(patch + xo + ((t / KW) * g.PW + (t % KW)) * kXS);
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[t][0], 0, 0, 0);
                }
            }
        }
    }

    // combine the WK K-wave partial accumulators through LDS (wave wk=0 of every cout tile owns the result)
    if constexpr (WK > 1) {
        float* red = reinterpret_cast<float*>(smem);  // NTAPS*NCI*256 floats per cout-wave, reused per round
        for (int round = 1; round < WK; ++round) {
            __syncthreads();
            if (wk == round) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                    for (int c = 0; c < NCI; ++c)
                        *reinterpret_cast<f32x4*>(red + (((wo * NTAPS + t) * NCI + c) * 64 + lane) * 4) = acc[t][c];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                    for (int c = 0; c < NCI; ++c)
                        acc[t][c] += *reinterpret_cast<const f32x4*>(red + (((wo * NTAPS + t) * NCI + c) * 64 + lane) * 4);
            }
        }
        if (wk != 0) return;
    }
    // partial slab of this split: [Cout][NTAPS][Cin] fp32
    float* out = p.partials + (size_t)split * p.Cout * NTAPS * Cin;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int c = 0; c < NCI; ++c) {
            const int ci = c0 + c * 16 + lr;
            if (ci >= Cin) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wo * 16 + lq * 4 + r;
                if (co < p.Cout) out[((size_t)co * NTAPS + t) * Cin + ci] = acc[t][c][r];
            }
        }
}

// dw[i] = sum over slabs, fixed order: thread (j, g) sums slabs g, g+4, ... of output 64*block + j; 4 groups meet in LDS
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partials, float* __restrict__ dw,
                                                         size_t n, int nparts) {
    __shared__ float red[4][64];
    const int j = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + j;
    float s0 = 0.f, s1 = 0.f;
    if (i < n) {
        int k = g;
        for (; k + 4 < nparts; k += 8) {
            s0 += partials[(size_t)k * n + i];
            s1 += partials[(size_t)(k + 4) * n + i];
        }
        if (k < nparts) s0 += partials[(size_t)k * n + i];
    }
    red[g][j] = s0 + s1;
    __syncthreads();
    if (g == 0 && i < n) dw[i] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
}

template <typename T>
int geom(const WgradParams& p, WGeom& g, int& WO) {
    constexpr int CK = WT<T>::CK, EPS = WT<T>::EPS;
    const int Cin = p.C0 + p.C1;
    VS_REQUIRE(Cin % EPS == 0 && p.C0 % EPS == 0, "conv_wgrad: channel counts must be multiples of %d", EPS);
    VS_REQUIRE(p.C1 == 0 || p.C0 % CK == 0, "conv_wgrad: concat boundary must be a multiple of %d", CK);
    VS_REQUIRE((p.KH == 3 && p.KW == 3) || (p.KH == 1 && p.KW == 1), "conv_wgrad: only 3x3 and 1x1 kernels");
    g.tw_shift = p.Wout >= 16 ? 4 : 3;
    const int TW = 1 << g.tw_shift;
    const int PT = (p.stride == 1 && p.Hout * p.Wout >= 128) ? 2 : 1;
    g.TH = 64 * PT / TW;
    g.PT = PT;
    g.tiles_h = cdiv(p.Hout, g.TH);
    g.tiles_w = cdiv(p.Wout, TW);
    g.PH = (g.TH - 1) * p.stride + p.KH;
    g.PW = (TW - 1) * p.stride + p.KW;
    g.cchunks = cdiv(Cin, CK);
    g.total_tiles = p.N * g.tiles_h * g.tiles_w;
    // Decomposition policy.  Workgroups = (cout tiles) x (cin chunks) x nsplit.  Splitting K (pixels) costs an fp32 slab
    // of |dw| bytes per split, written once and read once by the reduce; deep layers (big |dw|, few pixels) have enough
    // output-dimension parallelism, so they take narrower cout tiles (WO = 2 / 1 waves of 16 couts, the other waves split
    // K inside the workgroup and meet in LDS) instead of more slabs.
    const int target = vs_option("wgrad_target");
    static const double budget = (getenv("VS_WGRAD_SLAB_MB") ? atof(getenv("VS_WGRAD_SLAB_MB")) : 1.0e9) * 1048576.0;  // default: never trade tile width for slabs (measured slower)
    const double dw_bytes = (double)p.Cout * p.KH * p.KW * Cin * 4.0;
    const int wo_max = p.Cout >= 64 ? 4 : (p.Cout >= 32 ? 2 : 1);
    const int BMt = g.TH << g.tw_shift;
    int best_wo = wo_max, best_ns = 1;
    double best_slab = 1e30;
    for (int wo = wo_max; wo >= 1; wo >>= 1) {
        if ((4 / wo) > BMt / WT<T>::KSTEP) continue;  // not enough K-steps in a tile to feed the K-waves
        const int base = cdiv(p.Cout, 16 * wo) * g.cchunks;
        int ns = cdiv(target, base);
        if (ns > g.total_tiles) ns = g.total_tiles;
        ns = cdiv(g.total_tiles, cdiv(g.total_tiles, ns));
        const double slab = ns > 1 ? ns * dw_bytes : 0.0;
        if (slab <= budget) { best_wo = wo; best_ns = ns; best_slab = slab; break; }
        if (slab < best_slab) { best_wo = wo; best_ns = ns; best_slab = slab; }
    }
    WO = best_wo;
    g.nsplit = best_ns;
    g.ctiles = cdiv(p.Cout, 16 * WO);
    g.dys = 16 * WO * (int)sizeof(T) + 16;
    return VS_OK;
}

template <typename T, int WO, int NTAPS, int STRIDE, int PT>
int launch_one(const WgradParams& p, const WGeom& g, hipStream_t s) {
    static bool attr_set = false;
    auto kern = conv_wgrad_kernel<T, WO, NTAPS, STRIDE, PT>;
    VS_REQUIRE(g.PH * g.PW * 4 <= wg_patch_items(PT, STRIDE) * 256, "conv_wgrad: patch exceeds the staging budget");
    const int BM = g.TH << g.tw_shift;
    size_t lds = (size_t)g.PH * g.PW * kXS + (size_t)BM * g.dys;
    if (WO < 4) lds = std::max(lds, (size_t)WO * NTAPS * WT<T>::NCI * 256 * sizeof(float));  // K-wave combine buffer
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    WgradParams q = p;
    if (g.nsplit == 1) q.partials = p.dw;  // no K split: the single slab IS the result
    hipLaunchKernelGGL(kern, dim3(g.ctiles * g.cchunks, g.nsplit), dim3(256), lds, s, q, g);
    VS_LAUNCH_CHECK();
    if (g.nsplit == 1) return VS_OK;
    const size_t n = (size_t)p.Cout * NTAPS * (p.C0 + p.C1);
    const int nparts = g.nsplit;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv((int)n, 64)), dim3(256), 0, s, p.partials, p.dw, n, nparts);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

template <typename T>
int dispatch(const WgradParams& p, hipStream_t s) {
    WGeom g; int WO;
    int rc = geom<T>(p, g, WO);
    if (rc) return rc;
    const size_t need = (size_t)g.nsplit * p.Cout * p.KH * p.KW * (p.C0 + p.C1) * sizeof(float);
    VS_REQUIRE(p.partials && p.partial_bytes >= need, "conv_wgrad: workspace %zu < %zu", p.partial_bytes, need);
    const int nt = p.KH * p.KW;
#define VS_WG_CASE(wo, t)                                                                  \
    if (WO == wo && nt == t) {                                                             \
        if (p.stride == 2) return launch_one<T, wo, t, 2, 1>(p, g, s);                     \
        return g.PT == 2 ? launch_one<T, wo, t, 1, 2>(p, g, s) : launch_one<T, wo, t, 1, 1>(p, g, s); \
    }
    VS_WG_CASE(4, 9) VS_WG_CASE(2, 9) VS_WG_CASE(1, 9) VS_WG_CASE(4, 1) VS_WG_CASE(2, 1) VS_WG_CASE(1, 1)
#undef VS_WG_CASE
    return VS_ERR_UNSUPPORTED;
}

}  // namespace

int launch_slab_reduce(const float* partials, float* dw, size_t n, int nparts, hipStream_t s) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv((int)n, 64)), dim3(256), 0, s, partials, dw, n, nparts);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

size_t wgrad_workspace_bytes(int dtype, const WgradParams& p) {
    WGeom g; int WO;
    if (dtype == VS_BF16) { if (geom<bf16_t>(p, g, WO)) return 0; }
    else { if (geom<float>(p, g, WO)) return 0; }
    return (size_t)g.nsplit * p.Cout * p.KH * p.KW * (p.C0 + p.C1) * sizeof(float);
}

int launch_conv_wgrad(int dtype, const WgradParams& p, hipStream_t s) {
    if (dtype == VS_BF16) return dispatch<bf16_t>(p, s);
    if (dtype == VS_F32) return dispatch<float>(p, s);
    vs_set_error("conv_wgrad: bad dtype %d", dtype);
    return VS_ERR_INVALID;
}
