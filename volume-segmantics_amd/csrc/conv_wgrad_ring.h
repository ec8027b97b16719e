// conv_wgrad_ring_kernel: the bf16 weight gradient of a stride-1 3x3 convolution with LDS-DMA staging and a pinned
// software pipeline.  Same decomposition, same LDS images and the same fragment addressing as conv_wgrad_bf16_kernel
// (conv_wgrad.hip): a workgroup owns 16*MO couts x one 32-channel cin chunk x all 9 taps over a range of 128-pixel tiles
// (split-K, fp32 slab per split); wave (wc, wt) = cin slice wc (16 channels) x tap group wt (taps 0-4 / 5-8), every wave
// accumulating ALL 16*MO couts of its taps.  What changes is how a tile gets into LDS and how the k-loop is issued:
//   * the x patch (10 x 18 pixels x 32 channels) and the dy tile (128 pixels x 16*MO couts) of a pixel tile are fetched
//     by `buffer_load_dwordx4 ... lds` pieces into a two-stage ring (no staging registers, no ds_write pass); the 32-byte
//     slice swizzles of the two images sit in the per-lane SOURCE segment of a piece;
//   * a tile is a flat list of steps (k-step, tap): 4 x 5 or 4 x 4 per wave.  The transposed fragment reads
//     (ds_read_b64_tr_b16) run TWO steps ahead of the MFMAs that use them - the x fragment of step s+2 and, where step
//     s+2 opens a k-step, that k-step's MO dy fragments - pinned with sched_group_barrier.  (The plain kernel's loop read,
//     waited lgkmcnt(0) and multiplied tap by tap: 4 600 clocks per tile for 1 280 clocks of MFMA.)
//   * ONE barrier per tile, two steps before the tile's end (every read of the tile is issued by then): the wave waits
//     for its pieces of the next tile and its outstanding reads, meets the others, and the pieces of the tile after next
//     go into the stage just vacated, spread over four steps; the look-ahead reads cross into the next stage.
// Wave roles come from readfirstlane, so the two tap groups are two scalar branches with compile-time tap lists (the
// plain kernel's per-tap `continue`s compiled to exec-masked branches around every tap).
// Replaces the conv weight gradients of loss.backward() (vol_seg_2d_trainer.py:429) for the stride-1 3x3 layers.
#pragma once
#include "conv_ring.h"

namespace ring {

// (cout tile x cin chunk pair, K split) of a workgroup.  xmode 0: 2-D grid (pair = blockIdx.x, split = blockIdx.y).  Otherwise a
// 1-D grid, whose consecutive ids the hardware deals round-robin to the 8 XCDs (id & 7):
//   1 (8 or more splits): a K split lives on ONE XCD - every (cout tile, cin chunk) workgroup of a pixel range shares that L2,
//     so a pixel tile's input patches and dy tiles leave HBM once, not once per XCD that happens to hold one of their readers
//     (the step is HBM-bound: PMC traffic 13.6 GB per step at ~2.9 TB/s, of which the weight gradients were 3.6 GB - 1.8x
//     what they read algorithmically);
//   2 (1 / 2 / 4 splits): a split spans 8 / nsplit XCDs, each takes a contiguous cout-major range of `ppx` pairs (whole cout
//     tiles where they divide: dy once per split, an input chunk once per XCD of the split).
// false: this workgroup has nothing to do (ragged tail of the 1-D grid)
__device__ __forceinline__ bool wg_assign(int xmode, int npairs, int nsplit, int ppx, int& pair, int& split) {
    if (xmode == 0) { pair = blockIdx.x; split = blockIdx.y; return true; }
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    if (xmode == 1) {
        const int sl = j / npairs;
        pair = j - sl * npairs;
        split = sl * 8 + xcd;
        return split < nsplit;
    }
    const int X = 8 / nsplit;
    split = xcd / X;
    pair = (xcd - split * X) * ppx + j;
    return j < ppx && pair < npairs;
}
inline unsigned wg_grid(int xmode, int npairs, int nsplit, int ppx) {
    return xmode == 0 ? 0u : (xmode == 1 ? 8u * ((nsplit + 7) / 8) * npairs : 8u * ppx);
}

struct WGeomR {
    int tiles_h, tiles_w, total_tiles, cchunks, nsplit;
    int xmode, npairs, ppx;
    unsigned tw_magic, th_magic;
    unsigned long long* probe;
};

__device__ __forceinline__ uint2 tr16(const char* p) {
    const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(p));
    return __builtin_bit_cast(uint2, v);
}

// TWS / IMGS: a 128-pixel tile is 16 x 8 pixels of one image (4, 1) or two whole 8 x 8 images (3, 2)
template <int MO, int TWS, int IMGS, int WPS>
__global__ __launch_bounds__(256, WPS) void conv_wgrad_ring_kernel(WgradParams p, WGeomR g) {
    constexpr int NTAPS = 9, BM = 128, KS = BM / 32, TW = 1 << TWS, TH = BM / IMGS / TW, PW = TW + 2, PH = TH + 2, PP = PH * PW, P = IMGS * PP;
    constexpr int kXP = 64;
    constexpr int BNO = 16 * MO, DYP = BNO * 2 < 64 ? 64 : BNO * 2, DSLOTS = DYP / 16, DSEG = BNO / 8, NSL = DYP / 32;
    constexpr int PIT = cdivc(P * 4, 256), DIT = BM * DSLOTS / 256;
    constexpr int PATCH_B = PIT * 256 * 16, DY_B = BM * DYP, STAGE_B = PATCH_B + DY_B;
    constexpr int D = PIT + DIT, NG = 4, DG = cdivc(D, NG);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    const int wc = wave & 1, wt = wave >> 1;
    unsigned long long tprobe[5];
    if (g.probe) tprobe[0] = wall_clock64();

    int pair, split;
    if (!wg_assign(g.xmode, g.npairs, g.nsplit, g.ppx, pair, split)) return;
    const int ct = pair / g.cchunks, cc = pair - ct * g.cchunks;
    const int co0 = ct * BNO, c0 = cc * 32;
    const int Cin = p.C0 + p.C1;
    const int per = (g.total_tiles + g.nsplit - 1) / g.nsplit;
    const int t0 = split * per, t1 = min(g.total_tiles, t0 + per);
    const int H0 = p.Hin >> p.up0, W0 = p.Win >> p.up0;
    const bool from0 = c0 < p.C0;
    const int cs = from0 ? p.C0 : p.C1;
    const int cb = from0 ? c0 : c0 - p.C0;
    const int sh = from0 ? p.up0 : 0;
    const int Hs = from0 ? H0 : p.Hin, Ws = from0 ? W0 : p.Win;
    const u32x4 srdx = make_srd(from0 ? p.src0 : p.src1, (unsigned)(p.N * Hs * Ws * cs * 2));
    const u32x4 srdd = make_srd(p.dy, (unsigned)(p.N * p.Hout * p.Wout * p.Cout * 2));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- tile-invariant piece coordinates.  Piece i of a thread fills 16-byte LDS slot tid + 256 i of its image ----
    int pph[PIT], ppw[PIT], pco[PIT];      // pph: image << 8 | patch row
#pragma unroll
    for (int i = 0; i < PIT; ++i) {
        const int it = tid + i * 256;
        const int pp = it >> 2, sp = it & 3;
        const int img = pp / PP, q0 = pp - img * PP;
        const int ph = q0 / PW, pw = q0 - ph * PW;
        const int seg = ((((sp >> 1) ^ (pw >> 3)) & 1) << 1) | (sp & 1);       // source segment behind this slot (32-byte slices XOR patch column bit 3)
        pph[i] = (img << 8) | ph; ppw[i] = pw;
        pco[i] = (pp < P && cb + seg * 8 < cs) ? (cb + seg * 8) * 2 : -1;
    }
    int drow[DIT], dco[DIT];
#pragma unroll
    for (int i = 0; i < DIT; ++i) {
        const int it = tid + i * 256;
        const int r = it / DSLOTS, sp = it % DSLOTS;
        const int dgk = NSL == 4 ? (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) : ((r >> 3) & 1);
        const int dseg = ((((sp >> 1) ^ dgk) & (NSL - 1)) << 1) | (sp & 1);
        drow[i] = r;
        dco[i] = (dseg < DSEG && co0 + dseg * 8 < p.Cout) ? (co0 + dseg * 8) * 2 : -1;
    }
    // pieces [LO, HI) of pixel tile `tile` into the stage at LDS byte address sb
    auto issue = [&](auto lo_, auto hi_, int tile, unsigned sb) {
        constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value;
        int n, h0, w0;
        if constexpr (IMGS > 1) {
            n = tile * IMGS; h0 = 0; w0 = 0;
        } else {
            const int q = g.tiles_w == 1 ? tile : (int)__umulhi((unsigned)tile, g.tw_magic);
            const int tx = tile - q * g.tiles_w;
            n = g.tiles_h == 1 ? q : (int)__umulhi((unsigned)q, g.th_magic);
            const int ty = q - n * g.tiles_h;
            h0 = ty * TH; w0 = tx * TW;
        }
        const unsigned wbase = sb + (unsigned)(wave * 1024);
        static_for<0, PIT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i >= LO && i < HI) {
                const int ni = n + (pph[i] >> 8);
                const int hi = h0 - 1 + (pph[i] & 255), wi = w0 - 1 + ppw[i];
                const bool ok = pco[i] >= 0 && ni < p.N && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win;
                const int off = ((ni * Hs + (hi >> sh)) * Ws + (wi >> sh)) * cs * 2 + pco[i];
                dma16<i * 4096>(srdx, wbase, ok ? off : -1, 0);
            }
        });
        static_for<0, DIT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (PIT + i >= LO && PIT + i < HI) {
                const int rr = drow[i] >> TWS;                    // row of the tile: image rr / TH, row rr % TH
                const int ni = n + rr / TH;
                const int ho = h0 + rr % TH, wo = w0 + (drow[i] & (TW - 1));
                const bool ok = dco[i] >= 0 && ni < p.N && ho < p.Hout && wo < p.Wout;
                const int off = ((ni * p.Hout + ho) * p.Wout + wo) * p.Cout * 2 + dco[i];
                dma16<PATCH_B + i * 4096>(srdd, wbase, ok ? off : -1, 0);
            }
        });
    };

    // fragment addresses of k-step 0 inside a stage (pixels pa = 8 lq + (lr >> 2), pb = pa + 4); see conv_wgrad_bf16_kernel
    int a_addr[2][MO], x_addr[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int px = 8 * lq + (lr >> 2) + 4 * h;
        const int gk = NSL == 4 ? (((px >> 1) & 1) | (((px >> 3) & 1) << 1)) : ((px >> 3) & 1);
#pragma unroll
        for (int m = 0; m < MO; ++m) a_addr[h][m] = PATCH_B + ((px * DYP + (gk << 5) + (lr & 3) * 8) ^ (m << 5));
        const int row0 = (px >> TWS) * PW, col0 = px & (TW - 1);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            x_addr[h][kw] = (row0 + col0 + kw) * kXP + ((wc ^ ((col0 + kw) >> 3)) & 1) * 32 + (lr & 3) * 8;
    }
    // patch byte offset of k-step ks (32 pixels = 32 / TW tile rows; the rows of one image are contiguous in the patch)
    auto ks_off = [](int ks) { const int r = ks * (32 >> TWS); return ((r / TH) * PP + (r % TH) * PW) * kXP; };

    f32x4 acc[5][MO];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int m = 0; m < MO; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (t0 < t1) {
        issue(IC<0>{}, IC<D>{}, t0, lds0);
        if (t0 + 1 < t1) {
            issue(IC<0>{}, IC<D>{}, t0 + 1, lds0 + STAGE_B);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __builtin_amdgcn_s_barrier();
    if (g.probe) tprobe[1] = tprobe[2] = wall_clock64();

    // One tap group's share of every tile: NTP taps from T0.
    auto run = [&](auto t0_, auto ntp_) {
        constexpr int T0 = decltype(t0_)::value, NTP = decltype(ntp_)::value, NS = KS * NTP;
        uint4 af[2][MO], bf[4];   // x fragments: ring of 4 (NS = 20 or 16 is a multiple of 4, so the ring index runs on across tiles)
        auto read_b = [&](const char* st, int s, uint4& b) {       // x fragment of step s = (k-step s / NTP, tap T0 + s % NTP)
            const int ks = s / NTP, t = T0 + s % NTP, kh = t / 3, kw = t % 3;
            const uint2 lo = tr16(st + x_addr[0][kw] + kh * PW * kXP + ks_off(ks));
            const uint2 hi = tr16(st + x_addr[1][kw] + kh * PW * kXP + ks_off(ks));
            b = make_uint4(lo.x, lo.y, hi.x, hi.y);
        };
        auto read_a = [&](const char* st, int ks, uint4 (&a)[MO]) {   // the MO dy fragments of k-step ks
#pragma unroll
            for (int m = 0; m < MO; ++m) {
                const uint2 lo = tr16(st + a_addr[0][m] + ks * 32 * DYP);
                const uint2 hi = tr16(st + a_addr[1][m] + ks * 32 * DYP);
                a[m] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        };
        // everything step s2 needs, from stage st (reads of the NEXT tile's first steps come from the other stage)
        auto reads_for = [&](const char* st, auto s2_) {
            constexpr int s2 = decltype(s2_)::value;
            if constexpr (s2 % NTP == 0) read_a(st, s2 / NTP, af[(s2 / NTP) & 1]);
            read_b(st, s2, bf[s2 % 4]);
        };
        auto tile_body = [&](int tile, auto more_) {
            constexpr bool more = decltype(more_)::value != 0;
            const int sidx = (tile - t0) & 1;
            const char* cur = smem + sidx * STAGE_B;
            const char* nxt = smem + (sidx ^ 1) * STAGE_B;
            static_for<0, NS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s == NS - 2 && more) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                if constexpr (s >= NS - 2 && more) {
                    if (tile + 2 < t1) issue(IC<minc((s - (NS - 2)) * DG, D)>{}, IC<minc((s - (NS - 2) + 1) * DG, D)>{}, tile + 2, lds0 + sidx * STAGE_B);
                } else if constexpr (s < NG - 2 && more) {
                    if (tile > t0) issue(IC<minc((s + 2) * DG, D)>{}, IC<minc((s + 3) * DG, D)>{}, tile + 1, lds0 + (sidx ^ 1) * STAGE_B);
                }
                constexpr int s2 = s + 2;
                constexpr bool has_reads = s2 < NS || more;
                if constexpr (s2 < NS) reads_for(cur, IC<s2>{});
                else if constexpr (more) reads_for(nxt, IC<s2 - NS>{});
                constexpr int ks = s / NTP, tt = s % NTP;
#pragma unroll
                for (int m = 0; m < MO; ++m)
                    acc[tt][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[ks & 1][m]),
                                                                         __builtin_bit_cast(bf16x8, bf[s % 4]), acc[tt][m], 0, 0, 0);
                constexpr int NR = has_reads ? 2 + (((s2 % NS) % NTP == 0) ? 2 * MO : 0) : 0;
                if constexpr (NR > 0) __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MO, 0);
            });
        };
        if (t0 < t1) {
            static_assert(NS % 4 == 0 && KS % 2 == 0, "fragment ring indices must run on across tiles");
            reads_for(smem, IC<0>{});
            reads_for(smem, IC<1>{});
            for (int tile = t0; tile + 1 < t1; ++tile) tile_body(tile, IC<1>{});
            tile_body(t1 - 1, IC<0>{});
        }
    };
    if (wt == 0) run(IC<0>{}, IC<5>{});
    else run(IC<5>{}, IC<4>{});

    if (g.probe) tprobe[3] = wall_clock64();
    // partial slab of this split: [Cout][9][Cin] fp32
    float* out = p.partials + (size_t)split * p.Cout * NTAPS * Cin;
    const int ci = c0 + wc * 16 + lr;
    if (ci < Cin) {
#pragma unroll
        for (int tt = 0; tt < 5; ++tt) {
            const int t = wt * 5 + tt;
            if (t >= NTAPS) continue;
#pragma unroll
            for (int m = 0; m < MO; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + m * 16 + lq * 4 + r;
                    if (co < p.Cout) out[((size_t)co * NTAPS + t) * Cin + ci] = acc[tt][m][r];
                }
        }
    }
    if (g.probe) {
        __builtin_amdgcn_s_waitcnt(0);
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

template <int MO, int TWS, int IMGS>
constexpr size_t wgrad_ring_lds() {
    constexpr int BNO = 16 * MO, DYP = BNO * 2 < 64 ? 64 : BNO * 2, TW = 1 << TWS, TH = 128 / IMGS / TW;
    return 2 * (size_t)(cdivc(IMGS * (TH + 2) * (TW + 2) * 4, 256) * 256 * 16 + 128 * DYP);
}

}  // namespace ring
