// conv_ring_kernel: the stride-1 3x3 bf16 tile convolution with its staging done by LDS-DMA into a two-stage ring.
//
// Same implicit GEMM, same LDS images and the same accumulator layout as conv_igemm_kernel (conv_igemm.hip) - so the same
// epilogue (conv_common.h) - but nothing is staged through registers:
//   * every 32-channel chunk (input halo patch + BN x 9 weight slab) is fetched by `buffer_load_dwordx4 ... lds` pieces
//     (1 KiB per wave instruction, lane l's 16 bytes land at base + 16 l).  The XOR swizzle of the images therefore sits in
//     the per-lane SOURCE address (which 16-byte channel segment a lane fetches), the fragment reads apply the same XOR;
//     out-of-range offsets (zero padding, ragged tiles, channel tails, idle slots) make the hardware write zeros;
//   * two stages, ONE barrier per chunk: the fragment reads run LOOK = 2 taps ahead of the MFMAs, so at tap 7 of chunk c
//     every read of that chunk has been issued; there the wave waits for its pieces of chunk c+1 (issued a whole chunk
//     earlier) and for its outstanding fragment reads, meets the others, and the pieces of chunk c+2 go into the stage
//     chunk c just vacated - spread over the next four taps so that no tap carries more than a few DMA issues;
//   * the freed staging registers pay for the deeper fragment pipeline and for 64-wide cout tiles at one wave per SIMD
//     (6 LDS fragment reads per 8 MFMAs instead of 4 per 4), or for several images per pixel tile in the 8x8 layers (one
//     weight slab serves IMGS images).
// The DMA pieces are issued from inline assembly: hipcc orders every ds_read behind a pending LDS-DMA builtin with
// s_waitcnt vmcnt(0) (it cannot see that the pieces go to the other stage), which serialises exactly what this kernel
// overlaps.  Completion is tracked by hand: the only vector-memory operations between the prologue and the epilogue are
// these pieces, so vmcnt(0) at the barrier means "chunk c+1 has landed".
//
// Serves the stride-1 3x3 convolutions of smp.Unet(resnet34) forward and, with flipped weights, their data gradients
// (reference call sites vol_seg_2d_trainer.py:424,429, vol_seg_2d_predictor.py:44), including the decoder's nearest-x2
// upsample + concat and the zero-stuffed stride-2 gradients, all folded into the piece addresses.
#pragma once
#include "conv_common.h"

namespace ring {

constexpr int kRow = 64;   // LDS bytes per staged pixel / weight row: one 32-channel bf16 chunk

// 128-bit buffer descriptor held in SGPRs (raw buffer, 32-bit offsets, out-of-range reads return zero)
__device__ __forceinline__ u32x4 make_srd(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    u32x4 d;
    d.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    d.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    d.z = __builtin_amdgcn_readfirstlane(bytes);
    d.w = 0x00020000u;
    return d;
}
// One LDS-DMA piece: lane l's 16 bytes from (srd base + voff + soff) land at LDS byte address lds_base + IMM + 16 l.
// lds_base and soff are wave-uniform; IMM is added inside the statement so that the unrolled pieces of a chunk do not each
// pin a scalar register.  M0 is saved and restored around the piece (the compiler owns it).
template <int IMM>
__device__ __forceinline__ void dma16(const u32x4& srd, unsigned lds_base, int voff, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %1, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_base), "v"(voff), "s"(srd), "s"(soff), "n"(IMM) : "memory", "scc");
}

// the kernel's FIRST argument, read again from the kernel-argument segment at this point of the program
template <typename A>
__device__ __forceinline__ A reload_first_kernarg() {
#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((address_space(4))) A* pk = (const __attribute__((address_space(4))) A*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(pk));
    return *pk;
#else
    return A{};
#endif
}

struct Geom {
    int tiles_h, tiles_w;            // pixel tiles per image (1 x 1 when a tile spans IMGS whole images)
    int groups, ctiles;              // pixel tiles over the whole batch, cout tiles
    int xcs, xgn, ctl;               // the 8 XCDs as (8 >> xcs) pixel-tile partitions x (1 << xcs) cout partitions; ctl = cout tiles per partition
    unsigned ct_magic, ti_magic, tw_magic;   // ct_magic divides by ctl
    int out_nchw;
    unsigned long long* probe;       // phase timestamps (tools/conv_probe.py); null in normal operation
};

constexpr int cdivc(int a, int b) { return (a + b - 1) / b; }
constexpr int minc(int a, int b) { return a < b ? a : b; }
template <int V> struct IC { static constexpr int value = V; };
template <int LO, int HI, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (LO < HI) { f(IC<LO>{}); static_for<LO + 1, HI>(f); }
}

// BN: cout tile; PT: 16-pixel tiles per wave; NW: waves; TWS: log2 of the tile width; IMGS: images per pixel tile (> 1 only
// when the tile height equals the image height: 8x8 maps)
// NLOAD (ConvParams::nl_*, see conv_igemm_kernel): src0 is a PRE-norm tensor.  The prologue sums the producer's statistics bins
// into an LDS table while the first two chunks are in flight; the thread that issued a patch piece normalises it IN LDS behind the
// wait that says its pieces have landed and in front of the chunk barrier (own pieces only: no extra barrier), and the workgroups
// of the first cout tile store the normalised interior of their patch to nl_y from the same registers.
template <int BN, int PT, int NW, int TWS, int IMGS, int WPS, int PIN, bool NLOAD = false>
__global__ __launch_bounds__(NW * 64, WPS) void conv_ring_kernel(ConvParams p, Geom g) {
    typedef bf16_t T;
    constexpr int NT = NW * 64, NJ = BN / 16, BM = NW * PT * 16, TW = 1 << TWS, TH = BM / IMGS / TW;
    constexpr int PH = TH + 2, PW = TW + 2, PP = PH * PW, P = IMGS * PP;
    constexpr int PIT = cdivc(P * 4, NT);                 // patch pieces per wave and chunk
    constexpr int WROWS = 9 * BN, TS = NT / 4 / BN;       // weight rows; taps one piece row of the workgroup covers
    static_assert((NT / 4) % BN == 0, "cout tile must divide the rows of a piece pass");
    constexpr int WIT = cdivc(WROWS * 4, NT);             // weight pieces per wave and chunk
    constexpr int PATCH_B = PIT * NT * 16, WGT_B = WIT * NT * 16, STAGE_B = PATCH_B + WGT_B;
    constexpr int D = PIT + WIT, NG = 4, DG = cdivc(D, NG);   // pieces per chunk, issue groups, pieces per group
    static_assert(BM % (IMGS * TW) == 0 && TH * TW * IMGS == BM, "tile geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    unsigned long long tprobe[5];
    if (g.probe) tprobe[0] = wall_clock64();

    // workgroup -> tile, XCD-aware (see conv_igemm_kernel): the cout tiles of one pixel tile run back to back on one XCD.  An XCD is
    // (pixel-tile partition xg, cout partition xc): it fetches 1 / xgn of the input and 1 / (1 << xcs) of the weights - the split that
    // keeps an XCD's share inside its 4 MB L2 (launch_ring: layer4's 4.7 MB of weights against 2 MB of activations want 2 x 4, the
    // shallow layers 8 x 1)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int xc = xcd & ((1 << g.xcs) - 1), xg = xcd >> g.xcs;
    const int gl = g.ctl == 1 ? slot : (int)__umulhi((unsigned)slot, g.ct_magic);
    const int ytile = xc * g.ctl + (slot - gl * g.ctl);
    const int grp = gl * g.xgn + xg;
    if (grp >= g.groups) return;
    int n, h0, w0, nimg;
    if constexpr (IMGS > 1) {
        n = grp * IMGS; h0 = 0; w0 = 0;
        nimg = min(IMGS, p.N - n);
    } else {
        const int tiles_img = g.tiles_h * g.tiles_w;
        n = tiles_img == 1 ? grp : (int)__umulhi((unsigned)grp, g.ti_magic);
        const int timg = grp - n * tiles_img;
        const int ty = g.tiles_w == 1 ? timg : (int)__umulhi((unsigned)timg, g.tw_magic);
        const int tx = timg - ty * g.tiles_w;
        h0 = ty * TH; w0 = tx * TW;
        nimg = 1;
    }
    const int n0 = ytile * BN;
    const int Cin = p.C0 + p.C1;
    const int ush = p.up0 ? 1 : 0;
    const bool stuffed = p.up0 == 2;
    const int H0 = p.Hin >> ush, W0 = p.Win >> ush;
    const u32x4 srd0 = make_srd((const T*)p.src0 + (size_t)n * H0 * W0 * p.C0, (unsigned)(nimg * H0 * W0 * p.C0 * 2));
    const u32x4 srd1 = make_srd((const T*)p.src1 + (size_t)n * p.Hin * p.Win * p.C1, p.src1 ? (unsigned)(nimg * p.Hin * p.Win * p.C1 * 2) : 0u);
    const u32x4 srdw = make_srd(p.w, (unsigned)(p.Cout * 9 * Cin * 2));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- chunk-invariant piece addresses.  Piece i of this thread fills LDS slot (tid + i NT): pixel / row slot >> 2,
    // 16-byte position slot & 3, which holds channel segment (slot & 3) ^ key of the chunk ----
    int poff0[PIT], poff1[PIT];
    unsigned nl_meta = 0;       // NLOAD, per piece i: bits 4i, 4i+1 = channel segment, bit 4i+2 = this workgroup stores it to nl_y
#pragma unroll
    for (int i = 0; i < PIT; ++i) {
        const int it = tid + i * NT;
        const int pp = it >> 2;
        const int img = pp / PP, q = pp - img * PP;
        const int ph = q / PW, pw = q - ph * PW;
        const int seg = (it & 3) ^ ((pw >> 1) & 3);
        const int hi = h0 - 1 + ph, wi = w0 - 1 + pw;
        const bool ok = pp < P && img < nimg && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win;
        poff0[i] = (ok && !(stuffed && ((hi | wi) & 1))) ? (((img * H0 + (hi >> ush)) * W0 + (wi >> ush)) * p.C0 + seg * 8) * 2 : -1;
        poff1[i] = ok ? (((img * p.Hin + hi) * p.Win + wi) * p.C1 + seg * 8) * 2 : -1;
        if constexpr (NLOAD) {    // the tile's own pixels (not the halo), once per source pixel
            const bool mine = ok && ph >= 1 && ph <= TH && pw >= 1 && pw <= TW && !(ush && ((hi | wi) & 1));
            nl_meta |= ((unsigned)seg | (mine ? 4u : 0u)) << (4 * i);
        }
    }
    // weights: piece i covers rows i (NT / 4) + (tid >> 2); the rows advance by whole taps (TS per piece)
    const int wrow0 = tid >> 2;
    const int wnr = wrow0 % BN, wtap0 = wrow0 / BN;
    const int wseg = (tid & 3) ^ ((wrow0 >> 1) & 3);
    int woff[WIT];
#pragma unroll
    for (int i = 0; i < WIT; ++i)
        woff[i] = (n0 + wnr < p.Cout && wtap0 + i * TS < 9) ? (((n0 + wnr) * 9 + wtap0 + i * TS) * Cin + wseg * 8) * 2 : -1;
    const bool ragged_c = ((p.C0 | p.C1) & 31) != 0;     // some chunk has fewer than 32 valid channels

    // pieces [LO, HI) of the chunk whose first channel is c0, into the stage at LDS byte address `sb`
    auto issue = [&](auto lo_, auto hi_, int c0, unsigned sb) {
        constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value;
        const bool from0 = c0 < p.C0;
        const int cs = from0 ? p.C0 : p.C1;
        const int cb = from0 ? c0 : c0 - p.C0;
        const u32x4 srd = from0 ? srd0 : srd1;
        const int nseg = ragged_c ? (cs - cb) >> 3 : 4, nsegw = ragged_c ? (Cin - c0) >> 3 : 4;
        const unsigned wbase = sb + (unsigned)(wave * 1024);
        static_for<0, PIT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i >= LO && i < HI) {
                int off = from0 ? poff0[i] : poff1[i];
                if (ragged_c) {
                    const int it = tid + i * NT;
                    const int pp = it >> 2, q = pp % PP, pw = q % PW;
                    if (((it & 3) ^ ((pw >> 1) & 3)) >= nseg) off = -1;
                }
                dma16<i * NT * 16>(srd, wbase, off, cb * 2);
            }
        });
        static_for<0, WIT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (PIT + i >= LO && PIT + i < HI) {
                int off = woff[i];
                if (ragged_c && wseg >= nsegw) off = -1;
                dma16<PATCH_B + i * NT * 16>(srdw, wbase, off, c0 * 2);
            }
        });
    };

    // per-lane fragment read offsets inside a stage: pixel tile i, tap column kw (tap row kh adds kh * PW * kRow)
    int xb[PT][3];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int pl = (tid >> 6) * (PT * 16) + i * 16 + lr;
        const int r = pl >> TWS, tw = pl & (TW - 1);
        const int img = r / TH, th = r - img * TH;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            xb[i][kw] = (img * PP + th * PW + tw + kw) * kRow + ((lq ^ (((tw + kw) >> 1) & 3)) << 4);
    }
    const int wb = PATCH_B + lr * kRow + ((lq ^ ((lr >> 1) & 3)) << 4);

    f32x4 acc[PT][NJ];
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nch = (Cin + 31) >> 5;
    // NLOAD: [3][C0] floats behind the two stages - mean, invstd * gamma, beta of src0's channels (C0 a multiple of 32)
    float* cst = reinterpret_cast<float*>(smem + 2 * STAGE_B);
    __amdgpu_buffer_rsrc_t sry = __builtin_amdgcn_make_buffer_rsrc(nullptr, 0, 0, 0x00020000);
    // normalise this thread's own patch pieces of the chunk at c0 in place (they have landed: the caller waited), store the
    // first cout tile's share of nl_y
    auto nl_stage = [&](char* st, int c0) {
        if (c0 >= p.C0) return;                      // (uniform) a chunk of the skip tensor: already an activation
        static_for<0, PIT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            char* a = st + (tid + i * NT) * 16;
            const unsigned m = nl_meta >> (4 * i);
            const float* q = cst + c0 + (int)(m & 3u) * 8;
            float nm[8], na[8], nb[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 m4 = *reinterpret_cast<const float4*>(q + 4 * h), a4 = *reinterpret_cast<const float4*>(q + p.C0 + 4 * h),
                             b4 = *reinterpret_cast<const float4*>(q + 2 * p.C0 + 4 * h);
                nm[4 * h] = m4.x; nm[4 * h + 1] = m4.y; nm[4 * h + 2] = m4.z; nm[4 * h + 3] = m4.w;
                na[4 * h] = a4.x; na[4 * h + 1] = a4.y; na[4 * h + 2] = a4.z; na[4 * h + 3] = a4.w;
                nb[4 * h] = b4.x; nb[4 * h + 1] = b4.y; nb[4 * h + 2] = b4.z; nb[4 * h + 3] = b4.w;
            }
            const uint4 v = nl_apply8(*reinterpret_cast<const uint4*>(a), poff0[i] >= 0, nm, na, nb);
            *reinterpret_cast<uint4*>(a) = v;
            if (ytile == (c0 >> 5) % g.ctiles)     // (uniform) the chunks' stores are dealt round-robin to the cout tiles of this pixel tile
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, sry, (m & 4u) ? poff0[i] : (int)0x80000000, c0 * 2, 0);
        });
    };
    issue(IC<0>{}, IC<D>{}, 0, lds0);
    if (nch > 1) issue(IC<0>{}, IC<D>{}, 32, lds0 + STAGE_B);
    if constexpr (NLOAD) {
        // the statistics of src0's channels from the producer's fixed-point bins (the arithmetic of bn_apply_inline_kernel<T, true>),
        // summed while the first chunks are in flight; workgroup 0 publishes them
        sry = __builtin_amdgcn_make_buffer_rsrc((void*)((T*)p.nl_y + (size_t)n * H0 * W0 * p.C0), 0, nimg * H0 * W0 * p.C0 * 2, 0x00020000);
        const long long* bins = reinterpret_cast<const long long*>(p.nl_bins);
        const double rows = (double)p.nl_rows;
        for (int ch = tid; ch < p.C0; ch += NT) {
            long long sv = 0, qv = 0;
#pragma unroll 8
            for (int r = 0; r < p.nl_nb; ++r) { sv += bins[((size_t)r * 2 + 0) * p.C0 + ch]; qv += bins[((size_t)r * 2 + 1) * p.C0 + ch]; }
            const double mu = ((double)sv * (1.0 / kStatScale1)) / rows;
            double var = ((double)qv * (1.0 / kStatScale2)) / rows - mu * mu;
            if (var < 0.0) var = 0.0;
            const float is = (float)(1.0 / sqrt(var + (double)p.nl_eps));
            cst[ch] = (float)mu;
            cst[p.C0 + ch] = is * p.nl_gamma[ch];
            cst[2 * p.C0 + ch] = p.nl_beta[ch];
            if (blockIdx.x == 0) {
                p.nl_mean[ch] = (float)mu;
                p.nl_invstd[ch] = is;
                if (p.nl_rm) {
                    const double unbiased = p.nl_rows > 1 ? var * rows / (double)(p.nl_rows - 1) : var;
                    p.nl_rm[ch] = (float)((1.0 - p.nl_mom) * (double)p.nl_rm[ch] + p.nl_mom * mu);
                    p.nl_rv[ch] = (float)((1.0 - p.nl_mom) * (double)p.nl_rv[ch] + p.nl_mom * unbiased);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // both chunks have landed, the table is written
        __builtin_amdgcn_s_barrier();                                  // ... by every thread
        nl_stage(smem, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (nch > 1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");     // chunk 0 has landed (chunk 1 may still be in flight)
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (g.probe) tprobe[1] = tprobe[2] = wall_clock64();

    uint4 wf[3][NJ], xf[3][PT];
    auto read_frags = [&](const char* st, int tap, uint4 (&w)[NJ], uint4 (&x)[PT]) {
        const int kh = tap / 3, kw = tap % 3;
#pragma unroll
        for (int j = 0; j < NJ; ++j) w[j] = *reinterpret_cast<const uint4*>(st + wb + (tap * BN + j * 16) * kRow);
#pragma unroll
        for (int i = 0; i < PT; ++i) x[i] = *reinterpret_cast<const uint4*>(st + xb[i][kw] + kh * PW * kRow);
    };
    read_frags(smem, 0, wf[0], xf[0]);
    read_frags(smem, 1, wf[1], xf[1]);
    // one chunk: 9 taps.  MORE: another chunk follows (its stage is read from tap 7 on)
    auto chunk_body = [&](int c, auto more_) {
        constexpr bool more = decltype(more_)::value != 0;
        const int s = c & 1;
        const char* cur = smem + s * STAGE_B;
        const char* nxt = smem + (s ^ 1) * STAGE_B;
        static_for<0, 9>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t == 7 && more) {
                // all reads of chunk c are issued: wait for them and for this wave's pieces of chunk c+1, meet the others
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if constexpr (NLOAD) {      // this thread's pieces of chunk c+1 are in LDS: normalise them before anyone reads them
                    nl_stage(const_cast<char*>(nxt), (c + 1) * 32);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
            }
            // pieces of the chunk after next, in NG groups at taps 7, 8, 0, 1: into the stage the barrier above freed
            if constexpr (t >= 7 && more) {
                if (c + 2 < nch) issue(IC<minc((t - 7) * DG, D)>{}, IC<minc((t - 6) * DG, D)>{}, (c + 2) * 32, lds0 + s * STAGE_B);
            } else if constexpr (t < NG - 2 && more) {
                if (c >= 1) issue(IC<minc((t + 2) * DG, D)>{}, IC<minc((t + 3) * DG, D)>{}, (c + 1) * 32, lds0 + (s ^ 1) * STAGE_B);
            }
            if constexpr (t + 2 < 9) read_frags(cur, t + 2, wf[(t + 2) % 3], xf[(t + 2) % 3]);
            else if constexpr (more) read_frags(nxt, t + 2 - 9, wf[(t + 2) % 3], xf[(t + 2) % 3]);
#pragma unroll
            for (int i = 0; i < PT; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) mma16<T>(acc[i][j], wf[t % 3][j], xf[t % 3][i]);
            if constexpr (PIN == 1) {          // reads of tap t+2 first, then the MFMAs of tap t
                if constexpr (t + 2 < 9 || more) __builtin_amdgcn_sched_group_barrier(0x100, NJ + PT, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NJ * PT, 0);
            } else if constexpr (PIN == 2) {   // one read, then its share of the MFMAs
                constexpr int NR = (t + 2 < 9 || more) ? NJ + PT : 0, NM = NJ * PT;
                static_for<0, NR>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, (NM * (r + 1)) / NR - (NM * r) / NR, 0);
                });
                if constexpr (NR == 0) __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
            }
        });
    };
    for (int c = 0; c + 1 < nch; ++c) chunk_body(c, IC<1>{});
    chunk_body(nch - 1, IC<0>{});

    if (g.probe) tprobe[3] = wall_clock64();
    // The epilogue's parameters are read again from the kernel-argument segment (p is the first argument): held in scalar
    // registers across the main loop they would crowd out the descriptors the DMA statements need there.
    const ConvParams pe = reload_first_kernarg<ConvParams>();
    if constexpr (IMGS > 1) {   // the IMGS images of the tile as one tall image: same memory, same epilogue
        ConvParams q = pe;
        q.Hout = nimg * pe.Hout;
        conv_epilogue<T, BN, PT, NW>(q, TWS, g.out_nchw, grp, 0, 0, n0, grp, acc, smem);
    } else {
        conv_epilogue<T, BN, PT, NW>(pe, TWS, g.out_nchw, n, h0, w0, n0, grp, acc, smem);
    }
    if (g.probe) {
        __builtin_amdgcn_s_waitcnt(0);
        tprobe[4] = wall_clock64();
        if (tid == 0) {
            unsigned long long* o = g.probe + (size_t)blockIdx.x * 8;
            for (int i = 0; i < 5; ++i) o[i] = tprobe[i];
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            o[5] = hw; o[6] = xcc; o[7] = 0;
        }
    }
}

template <int BN, int PT, int NW, int TWS, int IMGS>
constexpr size_t ring_lds_bytes() {
    constexpr int NT = NW * 64, BM = NW * PT * 16, TW = 1 << TWS, TH = BM / IMGS / TW;
    constexpr int P = IMGS * (TH + 2) * (TW + 2);
    return 2 * (size_t)(cdivc(P * 4, NT) + cdivc(9 * BN * 4, NT)) * NT * 16;
}

// whether this geometry can run on conv_ring_kernel<.., TWS, IMGS> tiles
template <int PT, int NW, int TWS, int IMGS>
inline bool ring_geom_ok(const ConvParams& p) {
    constexpr int BM = NW * PT * 16, TW = 1 << TWS, TH = BM / IMGS / TW;
    if (IMGS > 1) return p.Hout == TH && p.Wout == TW && p.N % IMGS == 0 && !(p.pool0);
    return true;
}

template <int BN, int PT, int NW, int TWS, int IMGS, int WPS, int PIN = 0, bool NLOAD = false>
int launch_ring(const ConvParams& p, int out_nchw, unsigned long long* probe, hipStream_t s) {
    static bool attr_set = false;
    auto kern = conv_ring_kernel<BN, PT, NW, TWS, IMGS, WPS, PIN, NLOAD>;
    constexpr size_t lds0 = ring_lds_bytes<BN, PT, NW, TWS, IMGS>();
    static_assert(lds0 <= 160 * 1024, "conv_ring: LDS ring too large");
    const size_t lds = lds0 + (NLOAD ? (size_t)3 * p.C0 * sizeof(float) : 0);
    VS_REQUIRE(lds <= 160 * 1024 && (!NLOAD || (p.nl_bins && !(p.C0 & 31) && p.up0 != 2)), "conv_ring: normalise-on-load needs C0 in whole chunks and room for its table");
    constexpr int BM = NW * PT * 16, TW = 1 << TWS, TH = BM / IMGS / TW;
    VS_REQUIRE(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.dil <= 1 && !p.gc, "conv_ring: stride-1 3x3 only");
    VS_REQUIRE((double)IMGS * p.Hin * p.Win * std::max(p.C0, p.C1) * 2.0 < 4.0e9 && (double)p.Cout * 9 * (p.C0 + p.C1) * 2.0 < 4.0e9,
               "conv_ring: image or weight tensor exceeds the 32-bit piece offsets");
    VS_REQUIRE((ring_geom_ok<PT, NW, TWS, IMGS>(p)), "conv_ring: geometry does not fit the tile");
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    Geom g{};
    g.tiles_h = IMGS > 1 ? 1 : cdiv(p.Hout, TH);
    g.tiles_w = IMGS > 1 ? 1 : cdiv(p.Wout, TW);
    g.groups = IMGS > 1 ? cdiv(p.N, IMGS) : p.N * g.tiles_h * g.tiles_w;
    g.ctiles = cdiv(p.Cout, BN);
    {   // XCD split: minimise the bytes ONE XCD pulls through its L2 = weights / cout partitions + input / pixel-tile partitions
        const double wb = (double)p.Cout * 9 * (p.C0 + p.C1) * 2.0;
        const double ib = (double)p.N * ((double)(p.Hin >> (p.up0 ? 1 : 0)) * (p.Win >> (p.up0 ? 1 : 0)) * p.C0 + (double)p.Hin * p.Win * p.C1) * 2.0;
        int best = 0;
        double best_cost = 0;
        for (int xcs = 0; xcs <= 3; ++xcs) {
            const int xc = 1 << xcs;
            if (g.ctiles % xc) continue;
            const double cost = wb / xc + ib * xc / 8.0;
            if (xcs == 0 || !best_cost || cost < best_cost * 0.9) { best = xcs; best_cost = cost; }      // (a clear win only: the 8 x 1 map also shares the patch among neighbours in time)
        }
        g.xcs = best; g.xgn = 8 >> best; g.ctl = g.ctiles >> best;
    }
    g.ct_magic = 0xffffffffu / (unsigned)g.ctl + 1u;
    g.ti_magic = 0xffffffffu / (unsigned)(g.tiles_h * g.tiles_w) + 1u;
    g.tw_magic = 0xffffffffu / (unsigned)g.tiles_w + 1u;
    g.out_nchw = out_nchw;
    g.probe = probe;
    VS_REQUIRE(g.tiles_h * g.tiles_w < 65536 && p.N < 65536, "conv_ring: tile grid too large");
    const long nwg = (long)cdiv(g.groups, g.xgn) * g.ctl * 8;
    VS_REQUIRE(nwg < (1L << 31), "conv_ring: tile grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NW * 64), lds, s, p, g);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

}  // namespace ring
