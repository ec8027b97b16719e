// Convolution weight gradient on MFMA (gfx950): dw[co][tap][ci] = sum_pixels dy[p][co] * x[p (+) tap][ci].
//
// GEMM view per tap: D[cout][cin] with K = output pixels.  NHWC keeps channels contiguous, so both
// operands need "8 consecutive pixels of one channel" per lane - the transposed fragment.  bf16 gets it
// for free from ds_read_b64_tr_b16 on the naturally laid out [pixel][channel] LDS tiles; f32 uses
// v_mfma_f32_16x16x4_f32 whose fragments are one scalar per lane (plain ds_read_b32).
// The staged input patch is the same virtual tensor cat(upsample(src0), src1) the forward conv read.
// Split-K over pixel tiles: every workgroup writes an fp32 partial slab, a second kernel sums the slabs in a fixed
// order (bitwise reproducible, no float atomics).  Two kernels: conv_wgrad_bf16_kernel (bf16 fast path: all 16*MO couts of a
// 16-channel cin slice and 5 / 4 of the 9 taps per wave, swizzled LDS images, buffer-load staging) and the generic
// conv_wgrad_kernel (fp32, ragged channel counts).
//
// Replaces the conv weight gradients of loss.backward() (vol_seg_2d_trainer.py:429).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "prof.h"
#include "conv_wgrad_ring.h"
#include "conv_wgrad_rows.h"

namespace {

constexpr int kXS = 80;  // LDS bytes per staged input pixel (64 data + 16 pad)

template <typename T> struct WT;
template <> struct WT<bf16_t> { static constexpr int CK = 32, EPS = 8, NCI = 2, KSTEP = 32; };
template <> struct WT<float> { static constexpr int CK = 16, EPS = 4, NCI = 1, KSTEP = 4; };

struct WGeom {
    int tw_shift, TH, tiles_h, tiles_w, PH, PW, PT;
    int ctiles, cchunks, nsplit, total_tiles, dys;  // dys: LDS bytes per dy pixel row
    unsigned pw_magic, tw_magic, th_magic;          // x / PW, x / tiles_w, x / tiles_h by umulhi (bf16 fast path)
    int fast;                                        // bf16 fast path (conv_wgrad_bf16_kernel) applies
    int ring;                                        // 1 / 2: conv_wgrad_ring_kernel on 16 x 8 tiles / on pairs of 8 x 8 images
    int xmode, ppx;                                  // workgroup -> (pair, split) assignment (ring::wg_assign), bf16 kernels
    int rows, rows_per;                              // 1: conv_wgrad_rows16_kernel (full-resolution 16 -> 16 layers), output rows per workgroup
    unsigned long long* probe;                       // phase timestamps (tools/convlab); null in normal operation
};

__device__ __forceinline__ uint2 ds_read_tr16(const char* p) {
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(p));
    return __builtin_bit_cast(uint2, v);
}

constexpr int wg_patch_items(int pt, int stride) { return stride == 2 ? 5 : (pt == 2 ? 3 : 2); }

template <typename T, int WO, int NTAPS, int STRIDE, int PT, int DIL = 1>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p, WGeom g) {
    constexpr int CK = WT<T>::CK, EPS = WT<T>::EPS, NCI = WT<T>::NCI, KSTEP = WT<T>::KSTEP;
    constexpr int WK = 4 / WO, BNO = 16 * WO, KW = NTAPS == 9 ? 3 : 1;
    constexpr int BM = 64 * PT, SEGS = BNO / EPS;      // SEGS: 16-byte segments per dy pixel row
    constexpr int PITEMS = wg_patch_items(PT, DIL > 1 ? 2 : STRIDE), DTOTAL = BM * SEGS, DITEMS = (DTOTAL + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int wo = wave % WO, wk = wave / WO;
    const int TW = 1 << g.tw_shift;
    const int Cin = p.C0 + p.C1;
    const int P = g.PH * g.PW;
    char* patch = smem;
    char* dyl = smem + P * kXS;

    // grouped (p.cg: 32-channel super-groups, BNO = 32): the cout tile's own 32 input channels only, slab rows 32 long
    const int ct = blockIdx.x / g.cchunks, cc = blockIdx.x % g.cchunks;
    const int co0 = ct * BNO, c0 = p.cg ? co0 + cc * CK : cc * CK;
    const int Cw = p.cg ? 32 : (p.C0 + p.C1), cwb = p.cg ? co0 : 0;
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
                    const int toff = ((t / KW) * g.PW + (t % KW)) * DIL * kXS;
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
                    const float bv = *reinterpret_cast<const float*>(patch + xo + ((t / KW) * g.PW + (t % KW)) * DIL * kXS);
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
    float* out = p.partials + (size_t)split * p.Cout * NTAPS * Cw;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int c = 0; c < NCI; ++c) {
            const int ci = c0 + c * 16 + lr;
            if (ci >= Cin) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wo * 16 + lq * 4 + r;
                if (co < p.Cout) out[((size_t)co * NTAPS + t) * Cw + ci - cwb] = acc[t][c][r];
            }
        }
}

// ---- bf16 fast path ---------------------------------------------------------------------------------------------
// Workgroup = 16*MO couts x one 32-channel cin chunk x all taps, over a range of pixel tiles (split-K).  Wave (wc, wt):
// cin slice wc (16 channels) of the chunk and, for 3x3 kernels, tap group wt (taps 0-4 / 5-8); 1x1 kernels split the
// k-steps of a tile between the two waves of a slice instead and meet in LDS.  Every wave accumulates ALL 16*MO couts of
// its taps, so one transposed x fragment feeds MO MFMAs and one dy fragment up to 5: 8 + 10 LDS reads per 20 MFMAs at
// MO = 4 (the generic kernel below: 38 per 18, LDS-bound).
// LDS images: x patch rows of 64 B, dy rows of 128 / 64 B; 32-byte slices XOR-swizzled (patch: by bit 3 of the patch
// column, dy: by bits 1 and 3 of the pixel index) so every ds_read_b64_tr_b16 is conflict-free, and tap rows / k-steps
// are immediates on top of 8 address registers.  Staging: raw buffer loads (offset -1 -> zeros), register prefetch of
// the next tile.
constexpr int kXP = 64;

template <int MO, int NTAPS, int STRIDE, int PT, int DIL = 1>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(WgradParams p, WGeom g) {
    constexpr int KW = NTAPS == 9 ? 3 : 1, KH = KW;
    constexpr int BM = 64 * PT, KS = BM / 32;             // pixels / k-steps per tile
    constexpr int BNO = 16 * MO;
    constexpr int DYP = BNO * 2 < 64 ? 64 : BNO * 2;      // dy row pitch
    constexpr int DSEG = BNO / 8;                          // 16-byte segments of real data per dy row
    constexpr int NSL = DYP / 32;
    constexpr int RPP = 256 / DSEG;                        // dy rows one staging pass of the workgroup covers
    constexpr int DITEMS = (BM + RPP - 1) / RPP;
    constexpr bool kStatic = STRIDE == 1 && DIL == 1;
    constexpr int kTWS = PT == 1 ? 3 : 4;
    constexpr int PITEMS = wg_patch_items(PT, DIL > 1 ? 2 : STRIDE);
    constexpr int TG = NTAPS == 9 ? 2 : 1, KG = 2 / TG;    // tap groups / K groups among the two waves of a cin slice
    constexpr int TPW = NTAPS == 9 ? 5 : 1;                // taps per wave (second group: 4)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // one batch of scalar loads for the kernel arguments instead of a chain of dependent ones (see conv_igemm_kernel)
    asm volatile("" ::"s"(p.src0), "s"(p.src1), "s"(p.dy), "s"(p.partials), "s"(p.C0), "s"(p.C1), "s"(p.up0), "s"(p.N), "s"(p.Hin),
                 "s"(p.Win), "s"(p.Hout), "s"(p.Wout), "s"(p.pad), "s"(p.Cout), "s"(g.cchunks), "s"(g.nsplit), "s"(g.total_tiles),
                 "s"(g.tiles_w), "s"(g.tiles_h), "s"(g.tw_magic), "s"(g.th_magic), "s"(g.pw_magic), "s"(g.tw_shift), "s"(p.cg));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    unsigned long long tprobe[5];
    if (g.probe) tprobe[0] = wall_clock64();
    const int wc = wave & 1, wg2 = wave >> 1;
    const int wt = TG == 2 ? wg2 : 0, wk = KG == 2 ? wg2 : 0;
    const int tw_shift = kStatic ? kTWS : g.tw_shift;
    const int TW = 1 << tw_shift, TH = BM >> tw_shift;
    const int PW = (TW - 1) * STRIDE + (KW - 1) * DIL + 1, PH = (TH - 1) * STRIDE + (KH - 1) * DIL + 1;
    const int P = PH * PW;
    const int Cin = p.C0 + p.C1;
    char* patch = smem;
    char* dyl = smem + P * kXP;
    const int dummy = P * kXP + BM * DYP;                  // 16 spare bytes: target of the stores of idle staging items

    int pair, split;
    if (!ring::wg_assign(g.xmode, g.ctiles * g.cchunks, g.nsplit, g.ppx, pair, split)) return;
    const int ct = pair / g.cchunks, cc = pair - ct * g.cchunks;
    const int co0 = ct * BNO, c0 = p.cg ? co0 : cc * 32;    // grouped (MO = 2): the cout tile's own super-group
    const int Cw = p.cg ? 32 : (p.C0 + p.C1), cwb = p.cg ? co0 : 0;
    const int per = (g.total_tiles + g.nsplit - 1) / g.nsplit;
    const int t0 = split * per, t1 = min(g.total_tiles, t0 + per);
    const int H0 = p.Hin >> p.up0, W0 = p.Win >> p.up0;
    const bool from0 = c0 < p.C0;
    const int cs = from0 ? p.C0 : p.C1;
    const int cb = from0 ? c0 : c0 - p.C0;
    const int sh = from0 ? p.up0 : 0;
    const int Hs = from0 ? H0 : p.Hin, Ws = from0 ? W0 : p.Win;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(from0 ? p.src0 : p.src1, p.N * Hs * Ws * cs * 2);
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(p.dy, p.N * p.Hout * p.Wout * p.Cout * 2);

    // tile-invariant staging coordinates.  Patch item i of a thread: pixel (tid + 256 i) >> 2, 16-byte segment tid & 3.
    int ppk[PITEMS], pdst[PITEMS];                         // ppk = patch row << 16 | patch column
#pragma unroll
    for (int i = 0; i < PITEMS; ++i) {
        const int pp = (tid + i * 256) >> 2, seg = tid & 3;
        const int ph = kStatic ? pp / PW : (int)__umulhi((unsigned)pp, g.pw_magic), pw = pp - ph * PW;
        ppk[i] = (ph << 16) | pw;
        pdst[i] = pp < P ? pp * kXP + ((((seg >> 1) ^ (pw >> 3)) & 1) << 5) + (seg & 1) * 16 : dummy;
    }
    const int pco = cb + (tid & 3) * 8 < cs ? (cb + (tid & 3) * 8) * 2 : -1;          // channel byte offset; -1: beyond the source
    // dy item i: row r0 + i * RPP (RPP is a multiple of the tile width and of 16: same column, same swizzle), segment tid % DSEG
    const int r0 = tid / DSEG, dseg = tid % DSEG;
    const int dgk = NSL == 4 ? (((r0 >> 1) & 1) | (((r0 >> 3) & 1) << 1)) : ((r0 >> 3) & 1);
    const int ddst0 = P * kXP + r0 * DYP + (((dseg >> 1) ^ dgk) << 5) + (dseg & 1) * 16;
    const int dco = co0 + dseg * 8 < p.Cout ? (co0 + dseg * 8) * 2 : -1;
    const int dth = r0 >> tw_shift, dtw = r0 & (TW - 1);

    uint4 preg[PITEMS], dreg[DITEMS];
    auto load_tile = [&](int tile) {
        const int q = g.tiles_w == 1 ? tile : (int)__umulhi((unsigned)tile, g.tw_magic);
        const int tx = tile - q * g.tiles_w;
        const int n = g.tiles_h == 1 ? q : (int)__umulhi((unsigned)q, g.th_magic);
        const int ty = q - n * g.tiles_h;
        const int h0 = ty * TH, w0 = tx * TW;
        const int hbase = h0 * STRIDE - p.pad, wbase = w0 * STRIDE - p.pad;
#pragma unroll
        for (int i = 0; i < PITEMS; ++i) {
            const int hi = hbase + (ppk[i] >> 16), wi = wbase + (ppk[i] & 0xffff);
            const bool ok = pco >= 0 && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win;
            const int off = ((n * Hs + (hi >> sh)) * Ws + (wi >> sh)) * cs * 2 + pco;
            preg[i] = bload(rx, ok ? off : -1, 0);
        }
        const int wo_ = w0 + dtw;
#pragma unroll
        for (int i = 0; i < DITEMS; ++i) {
            const int ho = h0 + dth + i * (RPP >> tw_shift);
            const bool ok = dco >= 0 && r0 + i * RPP < BM && ho < p.Hout && wo_ < p.Wout;
            const int off = ((n * p.Hout + ho) * p.Wout + wo_) * p.Cout * 2 + dco;
            dreg[i] = bload(rd, ok ? off : -1, 0);
        }
    };

    // fragment addresses of k-step 0 (pixels pa = 8 lq + (lr >> 2), pb = pa + 4); k-step ks adds 32 pixels = whole tile
    // rows: a constant on top of these, and it leaves the swizzle keys (patch column; pixel bits 1 and 3) unchanged
    int a_addr[2], x_addr[2][KW];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int px = 8 * lq + (lr >> 2) + 4 * h;
        const int gk = NSL == 4 ? (((px >> 1) & 1) | (((px >> 3) & 1) << 1)) : ((px >> 3) & 1);
        a_addr[h] = px * DYP + (gk << 5) + (lr & 3) * 8;                  // fragment m: ^ (m << 5)
        const int row0 = ((px >> tw_shift) * STRIDE) * PW, col0 = (px & (TW - 1)) * STRIDE;
#pragma unroll
        for (int kw = 0; kw < KW; ++kw)
            x_addr[h][kw] = (row0 + col0 + kw * DIL) * kXP + ((wc ^ ((col0 + kw * DIL) >> 3)) & 1) * 32 + (lr & 3) * 8;
    }
    const int ks_rows = (32 >> tw_shift) * STRIDE * PW * kXP;            // patch bytes per k-step

    f32x4 acc[TPW][MO];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int m = 0; m < MO; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (t0 < t1) load_tile(t0);
    if (g.probe) tprobe[1] = wall_clock64();
    for (int tile = t0; tile < t1; ++tile) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PITEMS; ++i) *reinterpret_cast<uint4*>(smem + pdst[i]) = preg[i];
#pragma unroll
        for (int i = 0; i < DITEMS; ++i)
            *reinterpret_cast<uint4*>(smem + ((DITEMS * RPP <= BM || r0 + i * RPP < BM) ? ddst0 + i * RPP * DYP : dummy)) = dreg[i];
        __syncthreads();
        if (g.probe && tile == t0) tprobe[2] = wall_clock64();
        if (tile + 1 < t1) load_tile(tile + 1);  // in flight while the MFMAs below run
#pragma unroll
        for (int ks = wk; ks < KS; ks += KG) {
            uint4 af[MO];
#pragma unroll
            for (int m = 0; m < MO; ++m) {
                const uint2 lo = ds_read_tr16(dyl + ((a_addr[0] ^ (m << 5)) + ks * 32 * DYP));
                const uint2 hi = ds_read_tr16(dyl + ((a_addr[1] ^ (m << 5)) + ks * 32 * DYP));
                af[m] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) {
                // this wave's tt-th tap: group 0 = taps 0..4, group 1 = taps 5..8 (its 5th slot is idle)
                if (TG == 2 && tt == TPW - 1 && wt == 1) continue;
#pragma unroll
                for (int g2 = 0; g2 < TG; ++g2) {
                    if (TG == 2 && g2 != wt) continue;
                    const int t = g2 * TPW + tt;
                    if (t >= NTAPS) continue;
                    const int kh = t / KW, kw = t % KW;
                    const uint2 lo = ds_read_tr16(patch + x_addr[0][kw] + kh * DIL * PW * kXP + ks * ks_rows);
                    const uint2 hi = ds_read_tr16(patch + x_addr[1][kw] + kh * DIL * PW * kXP + ks * ks_rows);
                    const uint4 bf = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
                    for (int m = 0; m < MO; ++m)
                        acc[tt][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[m]),
                                                                             __builtin_bit_cast(bf16x8, bf), acc[tt][m], 0, 0, 0);
                }
            }
        }
    }

    if (g.probe) tprobe[3] = wall_clock64();
    if constexpr (KG == 2) {   // 1x1 kernels: the two K-groups of a cin slice meet in LDS
        f32x4* red = reinterpret_cast<f32x4*>(smem);
        __syncthreads();
        if (wk == 1) {
#pragma unroll
            for (int m = 0; m < MO; ++m) red[(wc * MO + m) * 64 + lane] = acc[0][m];
        }
        __syncthreads();
        if (wk == 1) return;
#pragma unroll
        for (int m = 0; m < MO; ++m) acc[0][m] += red[(wc * MO + m) * 64 + lane];
    }
    // partial slab of this split: [Cout][NTAPS][Cin] fp32
    float* out = p.partials + (size_t)split * p.Cout * NTAPS * Cw;
    const int ci = c0 + wc * 16 + lr;
    if (ci < Cin) {
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const int t = wt * TPW + tt;
            if (t >= NTAPS) continue;
#pragma unroll
            for (int m = 0; m < MO; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + m * 16 + lq * 4 + r;
                    if (co < p.Cout) out[((size_t)co * NTAPS + t) * Cw + ci - cwb] = acc[tt][m][r];
                }
        }
    }
    if (g.probe) {
        __builtin_amdgcn_s_waitcnt(0);  // stores issued and acknowledged
        tprobe[4] = wall_clock64();
        if (tid == 0) {
            unsigned long long* o = g.probe + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
            for (int i = 0; i < 5; ++i) o[i] = tprobe[i];
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            o[5] = hw; o[6] = xcc; o[7] = 0;
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
        // slabs g, g+4, g+8, ..: even ones into s0, odd ones into s1; eight loads are issued together (latency-bound kernel)
        for (int k = g; k < nparts; k += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (k + 4 * u < nparts) ? partials[(size_t)(k + 4 * u) * n + i] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
        }
    }
    red[g][j] = s0 + s1;
    __syncthreads();
    if (g == 0 && i < n) dw[i] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
}

// grouped convolutions: dense[cout][taps][32] (super-group slabs) -> dw[cout][taps][cg], the group's own block
__global__ __launch_bounds__(256) void wgrad_group_extract_kernel(const float* __restrict__ dense, float* __restrict__ dw, int n, int taps,
                                                                int cg) {
    const int i = blockIdx.x * 256 + threadIdx.x;      // index into dw
    if (i >= n) return;
    const int row = i / cg, j = i - row * cg;           // row = cout * taps + tap
    const int co = row / taps;
    dw[i] = dense[(size_t)row * 32 + ((co & 31) / cg) * cg + j];
}

template <typename T>
int geom(const WgradParams& p, WGeom& g, int& WO) {
    constexpr int CK = WT<T>::CK, EPS = WT<T>::EPS;
    const int Cin = p.C0 + p.C1;
    VS_REQUIRE(Cin % EPS == 0 && p.C0 % EPS == 0, "conv_wgrad: channel counts must be multiples of %d", EPS);
    VS_REQUIRE(p.C1 == 0 || p.C0 % CK == 0, "conv_wgrad: concat boundary must be a multiple of %d", CK);
    VS_REQUIRE((p.KH == 3 && p.KW == 3) || (p.KH == 1 && p.KW == 1), "conv_wgrad: only 3x3 and 1x1 kernels");
    // tile: 16x8 pixels (PT = 2) for stride-1 layers at least 16 wide, else 64 pixels (8x8; stride 2: 16x4 when wide enough)
    const int PT = (p.stride == 1 && p.Wout >= 16 && p.Hout * p.Wout >= 128 && p.dil < 4) ? 2 : 1;
    g.tw_shift = p.stride == 1 ? (PT == 2 ? 4 : 3) : (p.Wout >= 16 ? 4 : 3);
    const int TW = 1 << g.tw_shift;
    g.TH = 64 * PT / TW;
    g.PT = PT;
    g.tiles_h = cdiv(p.Hout, g.TH);
    g.tiles_w = cdiv(p.Wout, TW);
    const int dil = p.dil > 1 ? p.dil : 1;
    VS_REQUIRE(dil == 1 || ((dil == 2 || dil == 4) && p.KH == 3 && p.stride == 1 && !p.up0 && p.C1 == 0), "conv_wgrad: dilation 2 / 4 is built for plain stride-1 3x3 layers");
    g.PH = (g.TH - 1) * p.stride + (p.KH - 1) * dil + 1;
    g.PW = (TW - 1) * p.stride + (p.KW - 1) * dil + 1;
    g.cchunks = p.cg ? 32 / CK : cdiv(Cin, CK);      // grouped: cin chunks per 32-channel cout tile
    VS_REQUIRE(p.cg == 0 || (p.C1 == 0 && p.C0 == p.Cout && p.Cout % 32 == 0 && p.cg >= 4 && p.cg <= 32 && 32 % p.cg == 0 && !p.up0),
               "conv_wgrad: grouped convolutions need c0 == cout in 32-channel super-groups, 4 / 8 / 16 / 32 channels per group");
    g.total_tiles = p.N * g.tiles_h * g.tiles_w;
    g.pw_magic = 0xffffffffu / (unsigned)g.PW + 1u;
    g.tw_magic = 0xffffffffu / (unsigned)g.tiles_w + 1u;   // unused when the divisor is 1
    g.th_magic = 0xffffffffu / (unsigned)g.tiles_h + 1u;
    const double dw_bytes = (double)p.Cout * p.KH * p.KW * (p.cg ? 32 : Cin) * 4.0;
    // bf16 fast path: 16*MO couts per workgroup (all of them in every wave), K split across workgroups only
    g.probe = nullptr;
    g.xmode = 0; g.ppx = 0;
    g.fast = sizeof(T) == 2 && p.Cout % 8 == 0 && g.total_tiles < 65536 &&
             (double)p.N * p.Hin * p.Win * std::max(p.C0, p.C1) * 2.0 < 2.0e9 && (double)p.N * p.Hout * p.Wout * p.Cout * 2.0 < 2.0e9;
    g.ring = 0;
    g.rows = 0; g.rows_per = 0;
    // full-resolution 16-cout layers: the row-streaming kernels, one slab per workgroup (conv_wgrad_rows.h): 16 plain input channels
    // (rows = 1) or 32 at half resolution behind the loader's nearest upsampling (rows = 2).  256 workgroups x 3 rows in flight:
    // measured on the batch-32 step's launches against 128 / 512 / 768 / 1024 workgroups and 1 - 4 rows (profiles/r4_wgrad_rows_sweep.txt)
    if (g.fast && vs_option("wgrad_ring") && p.KH == 3 && p.stride == 1 && dil == 1 && !p.cg && p.C1 == 0 && p.Cout == 16 && p.pad == 1 &&
        ((p.C0 == 16 && !p.up0) || (p.C0 == 32 && p.up0 == 1 && p.Hout % 2 == 0)) &&
        (p.Wout == 128 || p.Wout == 256 || p.Wout == 512) && p.Hout >= 2 && (double)p.N * p.Hout * p.Wout * 32.0 < 2.0e9) {
        g.rows = p.up0 ? 2 : 1;
        const int R = p.N * p.Hout / g.rows;               // steps: output rows / source rows
        g.rows_per = std::max(2, cdiv(R, 256));
        g.nsplit = cdiv(R, g.rows_per);
        WO = 1; g.ctiles = 1; g.dys = 0;
        return VS_OK;
    }
    if (g.fast && vs_option("wgrad_ring") && p.KH == 3 && p.stride == 1 && dil == 1 && !p.cg && (p.C0 % 32) == 0 && (p.C1 % 32) == 0) {
        if (PT == 2) g.ring = 1;
        else if (p.Hout == 8 && p.Wout == 8 && p.N % 2 == 0 && p.Cout >= 64) {   // 8 x 8 maps: two images per 128-pixel tile
            g.ring = 2;
            g.total_tiles = p.N / 2;
        }
    }
    // K-split target (workgroups per launch).  The ring kernel is sized to half-fill the chip or less: the weight gradients
    // run on the side stream beside the caller's latency-bound chain, and fewer, longer-running workgroups leave CUs to it
    // (measured on the batch-32 step: 96 -> 4.68 ms, 256 -> 4.72, 512 -> 5.1; the plain kernel needs its 256: 4.91 at 128)
    const int target = vs_option(g.ring ? "wgrad_target" : "wgrad_target_plain");
    if (g.fast) {
        WO = p.Cout >= 64 ? 4 : (p.Cout >= 32 ? 2 : 1);   // = MO
        if (p.cg) WO = 2;
        g.ctiles = cdiv(p.Cout, 16 * WO);
        const int base = g.ctiles * g.cchunks;
        // K splits: `target` workgroups, and up to 4x that for layers whose slabs stay small (the wide, shallow decoder
        // layers are HBM-bound and need more workgroups per CU to cover their per-tile latency)
        const int by_slab = (int)std::min(1.0e6, (double)vs_option("wgrad_slab_mb") * 1048576.0 / dw_bytes);
        int ns = std::max(cdiv(target, base), std::min(cdiv(4 * target, base), by_slab));
        ns = std::min(ns, g.total_tiles);
        g.nsplit = cdiv(g.total_tiles, cdiv(g.total_tiles, ns));
        // XCD-aware assignment (ring::wg_assign): whole K splits per XCD from 8 splits on; 1 / 2 / 4 splits span 8 / 4 / 2 XCDs
        g.xmode = 0; g.ppx = 0;
        if (vs_option("wgrad_xcd")) {
            int want = g.nsplit;
            if (want == 3) want = 4;
            else if (want >= 5) want = (want + 7) / 8 * 8;
            want = std::min(want, g.total_tiles);
            const int got = cdiv(g.total_tiles, cdiv(g.total_tiles, want));
            if (got >= 8) { g.nsplit = got; g.xmode = 1; }
            else if (got == 1 || got == 2 || got == 4) { g.nsplit = got; g.xmode = 2; g.ppx = cdiv(base, 8 / got); }
        }
        g.dys = 0;
        return VS_OK;
    }
    // Generic kernel.  Workgroups = (cout tiles) x (cin chunks) x nsplit.  Splitting K (pixels) costs an fp32 slab of |dw|
    // bytes per split, written once and read once by the reduce; narrower cout tiles (WO = 2 / 1 waves of 16 couts, the
    // other waves split K inside the workgroup and meet in LDS) trade slabs for operand re-reads.
    static const double budget = (getenv("VS_WGRAD_SLAB_MB") ? atof(getenv("VS_WGRAD_SLAB_MB")) : 1.0e9) * 1048576.0;  // default: never trade tile width for slabs (measured slower)
    const int wo_max = p.cg ? 2 : (p.Cout >= 64 ? 4 : (p.Cout >= 32 ? 2 : 1));
    const int BMt = g.TH << g.tw_shift;
    int best_wo = wo_max, best_ns = 1;
    double best_slab = 1e30;
    for (int wo = wo_max; wo >= (p.cg ? 2 : 1); wo >>= 1) {
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

template <int MO, int NTAPS, int STRIDE, int PT, int DIL = 1>
int launch_fast(const WgradParams& p, const WGeom& g, hipStream_t s) {
    static bool attr_set = false;
    auto kern = conv_wgrad_bf16_kernel<MO, NTAPS, STRIDE, PT, DIL>;
    VS_REQUIRE(g.PH * g.PW * 4 <= wg_patch_items(PT, DIL > 1 ? 2 : STRIDE) * 256, "conv_wgrad: patch exceeds the staging budget");
    constexpr int BM = 64 * PT, DYP = 32 * MO < 64 ? 64 : 32 * MO;
    size_t lds = (size_t)g.PH * g.PW * kXP + (size_t)BM * DYP + 16;
    if (NTAPS == 1) lds = std::max(lds, (size_t)2 * MO * 1024);   // K-group combine buffer
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    WgradParams q = p;
    if (g.nsplit == 1) q.partials = p.dw;  // no K split: the single slab IS the result
    WGeom gg = g;
    gg.probe = vs_probe_buffer((size_t)g.ctiles * g.cchunks * g.nsplit);
    const dim3 grid = g.xmode ? dim3(ring::wg_grid(g.xmode, g.ctiles * g.cchunks, g.nsplit, g.ppx)) : dim3(g.ctiles * g.cchunks, g.nsplit);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, q, gg);
    VS_LAUNCH_CHECK();
    if (g.nsplit == 1) return VS_OK;
    return launch_slab_reduce(p.partials, p.dw, (size_t)p.Cout * NTAPS * (p.cg ? 32 : p.C0 + p.C1), g.nsplit, s);
}

// the row-streaming kernel of the full-resolution 16 -> 16 layers (conv_wgrad_rows.h)
template <int WQ>
int launch_rows_wgrad(const WgradParams& p, const WGeom& g, hipStream_t s) {
    constexpr int PF = 3;                                   // rows in flight per workgroup
    static bool attr_set[3] = {false, false, false};
    const int up = g.rows == 2;
    const bool planes = !up && p.dy_planes > 0;
    VS_REQUIRE(!p.dy_planes || (planes && p.dy_planes <= 7 && p.cout_live == p.dy_planes), "conv_wgrad: gradient planes are taken by the 16-channel row kernel only (<= 7 of them)");
    auto kern = up ? rows::conv_wgrad_rows_up32_kernel<WQ, PF> : (planes ? rows::conv_wgrad_rows16_kernel<WQ, PF, true> : rows::conv_wgrad_rows16_kernel<WQ, PF, false>);
    const size_t lds = up ? rows::up_lds_bytes(128 * WQ) : rows::lds_bytes(128 * WQ);
    if (!attr_set[planes ? 2 : up]) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[planes ? 2 : up] = true;
    }
    const int live = !up && p.cout_live > 0 && p.cout_live < 16 ? p.cout_live : 16;
    rows::RGeom gr{p.N * p.Hout, g.rows_per, g.nsplit, p.Hout, p.Wout, live};
    hipLaunchKernelGGL(kern, dim3(g.nsplit), dim3(256), lds, s, p, gr);
    VS_LAUNCH_CHECK();
    return launch_slab_reduce(p.partials, p.dw, (size_t)live * 9 * p.C0, g.nsplit, s);
}

// the ring-staged, pipelined kernel (conv_wgrad_ring.h): stride-1 3x3 layers on 128-pixel tiles
template <int MO, int TWS, int IMGS>
int launch_ring_wgrad(const WgradParams& p, const WGeom& g, hipStream_t s) {
    static bool attr_set = false;
    auto kern = ring::conv_wgrad_ring_kernel<MO, TWS, IMGS, 2>;
    constexpr size_t lds = ring::wgrad_ring_lds<MO, TWS, IMGS>();
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    WgradParams q = p;
    if (g.nsplit == 1) q.partials = p.dw;
    ring::WGeomR gr{};
    gr.tiles_h = g.tiles_h; gr.tiles_w = g.tiles_w; gr.total_tiles = g.total_tiles; gr.cchunks = g.cchunks; gr.nsplit = g.nsplit;
    gr.tw_magic = g.tw_magic; gr.th_magic = g.th_magic;
    gr.xmode = g.xmode; gr.npairs = g.ctiles * g.cchunks; gr.ppx = g.ppx;
    gr.probe = vs_probe_buffer((size_t)g.ctiles * g.cchunks * g.nsplit);
    const dim3 grid = g.xmode ? dim3(ring::wg_grid(g.xmode, gr.npairs, g.nsplit, g.ppx)) : dim3(g.ctiles * g.cchunks, g.nsplit);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, q, gr);
    VS_LAUNCH_CHECK();
    if (g.nsplit == 1) return VS_OK;
    return launch_slab_reduce(p.partials, p.dw, (size_t)p.Cout * 9 * (p.C0 + p.C1), g.nsplit, s);
}

template <typename T, int WO, int NTAPS, int STRIDE, int PT, int DIL = 1>
int launch_one(const WgradParams& p, const WGeom& g, hipStream_t s) {
    static bool attr_set = false;
    auto kern = conv_wgrad_kernel<T, WO, NTAPS, STRIDE, PT, DIL>;
    VS_REQUIRE(g.PH * g.PW * 4 <= wg_patch_items(PT, DIL > 1 ? 2 : STRIDE) * 256, "conv_wgrad: patch exceeds the staging budget");
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
    return launch_slab_reduce(p.partials, p.dw, (size_t)p.Cout * NTAPS * (p.cg ? 32 : p.C0 + p.C1), g.nsplit, s);
}

template <typename T>
int dispatch(const WgradParams& p, hipStream_t s) {
    WGeom g; int WO;
    int rc = geom<T>(p, g, WO);
    if (rc) return rc;
    const size_t slab = (size_t)p.Cout * p.KH * p.KW * (p.cg ? 32 : p.C0 + p.C1) * sizeof(float);
    const bool extract = p.cg && p.cg < 32;          // the super-group slabs are denser than dw: one more slab, then the gather
    const size_t need = ((size_t)g.nsplit + (extract ? 1 : 0)) * slab;
    VS_REQUIRE(p.partials && p.partial_bytes >= need, "conv_wgrad: workspace %zu < %zu", p.partial_bytes, need);
    const int nt = p.KH * p.KW;
    if (extract) {
        WgradParams q = p;
        q.cg = 32;                                    // as a grouped layer whose groups ARE the super-groups ...
        q.dw = p.partials + (size_t)g.nsplit * (slab / sizeof(float));
        const int rc2 = dispatch<T>(q, s);
        if (rc2) return rc2;
        const int n = p.Cout * nt * p.cg;             // ... then every group's cg x cg block out of its 32 x 32 slab
        hipLaunchKernelGGL(wgrad_group_extract_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, q.dw, p.dw, n, nt, p.cg);
        VS_LAUNCH_CHECK();
        return VS_OK;
    }
    VS_REQUIRE(!p.dy_planes || (sizeof(T) == 2 && g.rows == 1), "conv_wgrad: this launch cannot read gradient planes (ask conv_wgrad_takes_planes)");
    if constexpr (sizeof(T) == 2) {
        if (g.rows) return p.Wout == 128 ? launch_rows_wgrad<1>(p, g, s) : (p.Wout == 256 ? launch_rows_wgrad<2>(p, g, s) : launch_rows_wgrad<4>(p, g, s));
        if (g.fast) {
            if (p.dil == 2) {   // the dilated 3x3 layers (>= 32 channels): 64-pixel and 128-pixel tiles
                if (WO == 4) return g.PT == 2 ? launch_fast<4, 9, 1, 2, 2>(p, g, s) : launch_fast<4, 9, 1, 1, 2>(p, g, s);
                if (WO == 2) return g.PT == 2 ? launch_fast<2, 9, 1, 2, 2>(p, g, s) : launch_fast<2, 9, 1, 1, 2>(p, g, s);
                return VS_ERR_UNSUPPORTED;
            }
            if (p.dil == 4) {   // (64-pixel tiles only: the 128-pixel tile's 24 x 16 patch exceeds the staging budget)
                if (WO == 4) return launch_fast<4, 9, 1, 1, 4>(p, g, s);
                if (WO == 2) return launch_fast<2, 9, 1, 1, 4>(p, g, s);
                return VS_ERR_UNSUPPORTED;
            }
            if (g.ring == 1) {        // 16 x 8 tiles
                if (WO == 4) return launch_ring_wgrad<4, 4, 1>(p, g, s);
                if (WO == 2) return launch_ring_wgrad<2, 4, 1>(p, g, s);
                if (WO == 1) return launch_ring_wgrad<1, 4, 1>(p, g, s);
            }
            if (g.ring == 2 && WO == 4) return launch_ring_wgrad<4, 3, 2>(p, g, s);   // two 8 x 8 images per tile
#define VS_WGF_CASE(mo, t)                                                                \
    if (WO == mo && nt == t) {                                                            \
        if (p.stride == 2) return launch_fast<mo, t, 2, 1>(p, g, s);                      \
        return g.PT == 2 ? launch_fast<mo, t, 1, 2>(p, g, s) : launch_fast<mo, t, 1, 1>(p, g, s); \
    }
            VS_WGF_CASE(4, 9) VS_WGF_CASE(2, 9) VS_WGF_CASE(1, 9) VS_WGF_CASE(4, 1) VS_WGF_CASE(2, 1) VS_WGF_CASE(1, 1)
#undef VS_WGF_CASE
            return VS_ERR_UNSUPPORTED;
        }
    }
    if (p.dil == 2) {
        if (WO == 4) return g.PT == 2 ? launch_one<T, 4, 9, 1, 2, 2>(p, g, s) : launch_one<T, 4, 9, 1, 1, 2>(p, g, s);
        if (WO == 2) return g.PT == 2 ? launch_one<T, 2, 9, 1, 2, 2>(p, g, s) : launch_one<T, 2, 9, 1, 1, 2>(p, g, s);
        return VS_ERR_UNSUPPORTED;
    }
    if (p.dil == 4) {
        if (WO == 4) return launch_one<T, 4, 9, 1, 1, 4>(p, g, s);
        if (WO == 2) return launch_one<T, 2, 9, 1, 1, 4>(p, g, s);
        return VS_ERR_UNSUPPORTED;
    }
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

// (slab_reduce4_body<G> lives in common.h: the per-layer kernel below and the per-group launch of optim.hip share it)
template <int G>
__global__ __launch_bounds__(256) void slab_reduce4_kernel(const float4* __restrict__ partials, float4* __restrict__ dw, size_t n4, int nparts) {
    slab_reduce4_body<G>(partials, dw, n4, nparts, blockIdx.x);
}

int slab_reduce_groups(const float* partials, const float* dw, size_t n, int nparts) {
    if ((n & 3) || ((uintptr_t)partials & 15) || ((uintptr_t)dw & 15)) return 0;
    const size_t n4 = n / 4;
    if (n4 >= 131072 || nparts < 8) return 1;
    if (n4 >= 32768 || nparts < 32) return 4;
    return 16;
}

int launch_slab_reduce(const float* partials, float* dw, size_t n, int nparts, hipStream_t s) {
    const int G = slab_reduce_groups(partials, dw, n, nparts);
    if (G == 0) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv((int)n, 64)), dim3(256), 0, s, partials, dw, n, nparts);
    } else {
        const size_t n4 = n / 4;
        if (G == 1)
            hipLaunchKernelGGL(slab_reduce4_kernel<1>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const float4*)partials, (float4*)dw, n4, nparts);
        else if (G == 4)
            hipLaunchKernelGGL(slab_reduce4_kernel<4>, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, s, (const float4*)partials, (float4*)dw, n4, nparts);
        else
            hipLaunchKernelGGL(slab_reduce4_kernel<16>, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, s, (const float4*)partials, (float4*)dw, n4, nparts);
    }
    VS_LAUNCH_CHECK();
    return VS_OK;
}

bool conv_wgrad_takes_planes(int dtype, const WgradParams& p) {
    WGeom g; int WO;
    return dtype == VS_BF16 && p.dy_planes >= 1 && p.dy_planes <= 7 && geom<bf16_t>(p, g, WO) == VS_OK && g.rows == 1;
}

bool conv_wgrad_honours_cout_live(int dtype, const WgradParams& p) {
    WGeom g; int WO;
    return dtype == VS_BF16 && geom<bf16_t>(p, g, WO) == VS_OK && g.rows == 1;
}

size_t wgrad_workspace_bytes(int dtype, const WgradParams& p) {
    WGeom g; int WO;
    if (dtype == VS_BF16) { if (geom<bf16_t>(p, g, WO)) return 0; }
    else { if (geom<float>(p, g, WO)) return 0; }
    return ((size_t)g.nsplit + (p.cg && p.cg < 32 ? 1 : 0)) * p.Cout * p.KH * p.KW * (p.cg ? 32 : p.C0 + p.C1) * sizeof(float);
}

int launch_conv_wgrad(int dtype, const WgradParams& p, hipStream_t s) {
    if (dtype == VS_BF16) return dispatch<bf16_t>(p, s);
    if (dtype == VS_F32) return dispatch<float>(p, s);
    vs_set_error("conv_wgrad: bad dtype %d", dtype);
    return VS_ERR_INVALID;
}
