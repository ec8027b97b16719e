// conv_stream_kernel: conv_ring_kernel (conv_ring.h) made PERSISTENT - one workgroup per CU walks a sequence of pixel tiles
// and the LDS-DMA rings never drain between them.
//
// Why: on the large grids of prediction (batch 64 of 512 x 512 slices: 4096 tiles of a 64 -> 64 layer) a workgroup of the
// tile kernels lives ~12 us of which the MFMA loop is 2.4 - 4.6 us (tools/convlab probe stamps: setup + first data in LDS
// 3.5 us, epilogue + store acknowledgement 3.8 us) - per-tile latencies that two co-resident workgroups only partly hide.
// Here the chunk stream is flattened across tiles: while the last chunks of tile k are multiplied the pieces of tile k+1's
// first chunks are already landing; the finished tile's outputs leave as fire-and-forget buffer stores that the next wait
// steps over (see "counted wait"), and the next tile's MFMAs start right behind them.
//
//   * workgroup b: XCD b & 7, cout tile (b >> 3) % ctiles, tile sequence (b >> 3) / ctiles; its k-th pixel tile is
//     ((k * lanes + sequence) * 8 + xcd): the cout tiles of a pixel tile run on one XCD (shared halo in that L2);
//   * piece addresses are offsets into WHOLE-tensor descriptors (n folded into the offset), recomputed per tile for the
//     tile after the current one - 3 pieces x ~12 VALU instructions per 1152 MFMAs;
//   * two rings: weight slabs in TWO slots (they come from L2), input patches in THREE (they come from HBM).  Chunk x of the
//     stream: taps 0, 1 issue the patch pieces of chunk x+2; tap 7 waits for chunk x+1 and meets the other waves (every
//     read of chunk x is done: fragment reads run two taps ahead); taps 7, 8 issue the weight pieces of chunk x+2 into
//     the slot just vacated.  With 64 input channels (two chunks) the two weight slabs of the cout tile simply STAY in
//     their slots: no weight traffic after the prologue;
//   * counted wait: vector-memory operations retire in issue order (MI355X_MICROARCH.md, s_waitcnt), and the epilogue
//     issues EXACTLY PT * NJ buffer stores per lane (unconditional; masked lanes use an out-of-range offset) and not one
//     compiler-visible vector load (scale / shift sit in LDS, the residual quads are fetched by hand-issued loads at the
//     start of the tile's last chunk and retired by that chunk's own wait) - so `s_waitcnt vmcnt(PIT [+ PT * NJ])` at tap 7
//     retires chunk x+1 and leaves the patch pieces of chunk x+2 (and, behind a tile boundary, the stores) in flight.
// Measured (tools/convlab, prediction shapes at batch 64, same process as the tile kernel): 64 -> 64 @128^2 1.05x (that
// layer moves 268 MB for 77 GFLOP - the HBM read + write stream bounds it), 128 -> 128 @64^2 1.17x, 256 -> 256 @32^2 1.24x,
// 512 -> 512 @16^2 1.12x, up64+64 -> 32 @256^2 1.25x; the 512^3 12-direction prediction 0.587 -> 0.512 s (tools/ab_predict.py).
// Evaluation-mode epilogue only (folded BatchNorm scale / shift, residual, ReLU / swish, 16-bit NHWC store; T = bf16_t or f16_t): the layers this
// kernel is chosen for are the prediction forward's (conv_igemm.hip: stream_mode); everything else stays on
// conv_igemm_kernel / conv_ring_kernel.  Accumulation order (chunk-major, taps 0..8) and the epilogue's arithmetic are those
// of conv_igemm_kernel: the outputs are bit-identical (tests/test_hip_ops.py), so a slice's prediction does not depend on
// which kernel its batch size selected.
#pragma once
#include "conv_ring.h"

namespace ring {

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// one residual quad by a hand-issued load (its completion is tracked by the kernel's own waits), and the statement that
// orders later uses of such a register behind those waits
template <int IMM>
__device__ __forceinline__ void quad_load(u32x2& dst, int voff, const u32x4& srd) {
    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen offset:%3" : "=v"(dst) : "v"(voff), "s"(srd), "n"(IMM) : "memory");
}
__device__ __forceinline__ void quad_tie(u32x2& r) { asm volatile("" : "+v"(r)); }

struct SGeom {
    int tiles_h, tiles_w;            // pixel tiles per image
    int groups, ctiles, lanes;       // pixel tiles over the batch; cout tiles; tile sequences per XCD and cout tile
    unsigned ct_magic, ti_magic, tw_magic;
    unsigned long long* probe;       // per-workgroup stamps (tools/convlab); null in normal operation
    int stagger;                     // start delay, in steps of 512 clocks x ((workgroup >> 3) & 7): spreads the tile-boundary store bursts
};

// MODE: 0 = one input tensor, no residual; 1 = one input tensor + residual in the epilogue (the second convolution of a ResNet
// block); 2 = two input tensors (decoder: x2-upsampled features + skip), no residual.  Each keeps only its own registers.
template <typename T, int BN, int PT, int NW, int TWS, int WPS, int PIN, int MODE, bool PROBE = false>
__global__ __launch_bounds__(NW * 64, WPS) void conv_stream_kernel(ConvParams p, SGeom g) {
    static_assert(sizeof(T) == 2, "conv_stream_kernel: 16-bit storage (bf16_t or f16_t)");
    constexpr int NT = NW * 64, NJ = BN / 16, BM = NW * PT * 16, TW = 1 << TWS, TH = BM / TW;
    constexpr int PH = TH + 2, PW = TW + 2, PP = PH * PW;
    constexpr int PIT = cdivc(PP * 4, NT);
    constexpr int WROWS = 9 * BN, TS = NT / 4 / BN;
    static_assert((NT / 4) % BN == 0, "cout tile must divide the rows of a piece pass");
    constexpr int WIT = cdivc(WROWS * 4, NT);
    constexpr int PATCH_B = PIT * NT * 16, WGT_B = WIT * NT * 16;
    constexpr int NST = PT * NJ;                         // buffer stores per lane and tile (the counted wait relies on it)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    unsigned long long t_begin = 0, t_loop = 0, t_wait = 0, t_epi = 0, t_a = 0;   // probe: 100 MHz stamps / sums
    int n_tiles = 0;
    if (PROBE && g.probe) t_begin = wall_clock64();

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int seq = g.ctiles == 1 ? slot : (int)__umulhi((unsigned)slot, g.ct_magic);
    const int ytile = slot - seq * g.ctiles;
    const int gstep = g.lanes * 8;
    int grp = seq * 8 + xcd;
    if (grp >= g.groups) return;
    if (g.stagger) {
        for (int k = ((blockIdx.x >> 3) & 7) * g.stagger; k > 0; --k) __builtin_amdgcn_s_sleep(8);      // 512 clocks (~0.2 us) each
    }
    const int n0 = ytile * BN;
    const int Cin = p.C0 + p.C1;
    const int ush = p.up0 ? 1 : 0;
    const int H0 = p.Hin >> ush, W0 = p.Win >> ush;
    const u32x4 srd0 = make_srd(p.src0, (unsigned)((size_t)p.N * H0 * W0 * p.C0 * 2));
    const u32x4 srd1 = make_srd(p.src1, p.src1 ? (unsigned)((size_t)p.N * p.Hin * p.Win * p.C1 * 2) : 0u);
    const u32x4 srdw = make_srd(p.w, (unsigned)(p.Cout * 9 * Cin * 2));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int tiles_img = g.tiles_h * g.tiles_w;

    // pixel tile index -> image, tile origin
    auto decode = [&](int gi, int& n, int& h0, int& w0) {
        n = tiles_img == 1 ? gi : (int)__umulhi((unsigned)gi, g.ti_magic);
        const int timg = gi - n * tiles_img;
        const int ty = g.tiles_w == 1 ? timg : (int)__umulhi((unsigned)timg, g.tw_magic);
        h0 = ty * TH; w0 = (timg - ty * g.tiles_w) * TW;
    };
    // byte offsets of this thread's patch pieces for the tile at (n, h0, w0), into the whole-tensor descriptors
    auto tile_offsets = [&](int n, int h0, int w0, int (&o0)[PIT], int (&o1)[PIT]) {
        int tv = tid;
        asm volatile("" : "+v"(tv));      // per-thread constants are recomputed here, once per tile, not carried across the tile loop
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            const int it = tv + i * NT;
            const int pp = it >> 2;
            const int ph = pp / PW, pw = pp - ph * PW;
            const int seg = (it & 3) ^ ((pw >> 1) & 3);
            const int hi = h0 - 1 + ph, wi = w0 - 1 + pw;
            const bool ok = pp < PP && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win;
            o0[i] = ok ? (int)(((unsigned)((n * H0 + (hi >> ush)) * W0 + (wi >> ush)) * (unsigned)p.C0 + seg * 8) * 2u) : -1;
            if constexpr (MODE == 2) o1[i] = ok ? (int)(((unsigned)((n * p.Hin + hi) * p.Win + wi) * (unsigned)p.C1 + seg * 8) * 2u) : -1;
        }
    };
    int poff0[PIT], poff1[PIT], noff0[PIT], noff1[PIT];
    int n, h0, w0;
    decode(grp, n, h0, w0);
    tile_offsets(n, h0, w0, poff0, poff1);
    bool has_next = grp + gstep < g.groups;
    {
        int nn = 0, nh = 0, nw = 0;
        if (has_next) decode(grp + gstep, nn, nh, nw);
        tile_offsets(nn, nh, nw, noff0, noff1);
    }
    const int wrow0 = tid >> 2;
    const int wnr = wrow0 % BN, wtap0 = wrow0 / BN;
    const int wseg = (tid & 3) ^ ((wrow0 >> 1) & 3);
    int woff[WIT];
#pragma unroll
    for (int i = 0; i < WIT; ++i)
        woff[i] = (n0 + wnr < p.Cout && wtap0 + i * TS < 9) ? (((n0 + wnr) * 9 + wtap0 + i * TS) * Cin + wseg * 8) * 2 : -1;

    // LDS: two weight slots, three patch slots, the affine table.  Patch pieces [LO, HI) of the chunk whose first channel is
    // c0 - of the current tile, or (nx) of the one after it - into patch slot `slot`; weight pieces likewise
    constexpr int W0_B = 0, P0_B = 2 * WGT_B, AFF_B = P0_B + 3 * PATCH_B;      // byte offsets of the two rings and the affine table
    auto issue_p = [&](auto lo_, auto hi_, int c0, int slot, bool nx) {
        constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value;
        const bool from0 = MODE != 2 || c0 < p.C0;
        const int cb = from0 ? c0 : c0 - p.C0;
        const u32x4 srd = from0 ? srd0 : srd1;
        const unsigned wbase = lds0 + (unsigned)(P0_B + slot * PATCH_B + wave * 1024);
        static_for<0, PIT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i >= LO && i < HI) {
                int a = poff0[i], b = noff0[i];
                if constexpr (MODE == 2) { a = from0 ? a : poff1[i]; b = from0 ? b : noff1[i]; }
                dma16<i * NT * 16>(srd, wbase, nx ? b : a, cb * 2);
            }
        });
    };
    auto issue_w = [&](auto lo_, auto hi_, int c0, int slot) {
        constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value;
        const unsigned wbase = lds0 + (unsigned)(W0_B + slot * WGT_B + wave * 1024);
        static_for<0, WIT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i >= LO && i < HI) dma16<i * NT * 16>(srdw, wbase, woff[i], c0 * 2);
        });
    };

    int xb[PT][3];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int pl = (tid >> 6) * (PT * 16) + i * 16 + lr;
        const int th = pl >> TWS, tw = pl & (TW - 1);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            xb[i][kw] = (th * PW + tw + kw) * kRow + ((lq ^ (((tw + kw) >> 1) & 3)) << 4);
    }
    const int wb = lr * kRow + ((lq ^ ((lr >> 1) & 3)) << 4);

    f32x4 acc[PT][NJ];
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- everything the epilogue needs, fetched ONCE: no compiler-visible vector load may sit in the tile loop (hipcc would
    // wait for it with a vmcnt that also drains the LDS-DMA pieces in flight).  Cout is a multiple of BN (launcher).
    const int emode = (p.scale ? 1 : 0) | (p.shift ? 2 : 0);
    const int erelu = p.relu;
    float* eaff = reinterpret_cast<float*>(smem + AFF_B);                     // [2][BN]: this cout tile's scale and shift, behind the rings
    if (tid < 2 * BN) {
        const float* src = tid < BN ? p.scale : p.shift;
        eaff[tid] = src ? src[n0 + (tid & (BN - 1))] : (tid < BN ? 1.f : 0.f);
    }
    const unsigned obytes = (unsigned)((size_t)p.N * p.Hout * p.Wout * p.Cout * 2);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.out, (int)obytes);
    const u32x4 srdr = make_srd(p.residual, p.residual ? obytes : 0u);
    const int eHout = p.Hout, eWout = p.Wout, eCout2 = p.Cout * 2;
    constexpr int kOob = (int)0x80000000;                // beyond every descriptor (tensors < 2 GB): loads return 0, stores vanish
    u32x2 rres[MODE == 1 ? PT : 1][MODE == 1 ? NJ : 1];  // the residual's raw bf16 quads of the current tile
    int opix[PT];
#pragma unroll
    for (int i = 0; i < (MODE == 1 ? PT : 1); ++i)
#pragma unroll
        for (int j = 0; j < (MODE == 1 ? NJ : 1); ++j) rres[i][j] = u32x2{0u, 0u};
    // output / residual byte offsets of this lane's pixels of the tile at (tn, th0, tw0)
    auto pixel_offsets = [&](int tn, int th0, int tw0) {
        int tv = tid;
        asm volatile("" : "+v"(tv));
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int pl = (tv >> 6) * (PT * 16) + i * 16 + (tv & 15);
            const int ho = th0 + (pl >> TWS), wo = tw0 + (pl & (TW - 1));
            opix[i] = (ho < eHout && wo < eWout) ? (int)((unsigned)((tn * eHout + ho) * eWout + wo) * (unsigned)eCout2) + (n0 + ((tv >> 4) & 3) * 4) * 2 : kOob;
        }
    };
    // the residual quads of the current tile, by hand-issued loads (the waits around them are this kernel's own)
    auto residual_loads = [&]() {
        static_for<0, (MODE == 1 ? PT : 0)>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            static_for<0, NJ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                quad_load<j * 32>(rres[i][j], opix[i], srdr);
            });
        });
    };

    const int nch = Cin >> 5;                            // >= 2, every chunk 32 channels wide (checked by the launcher)
    const bool wres = nch == 2;                          // both weight chunks of this cout tile stay in their slots for good
    constexpr int WH = cdivc(WIT, 2), PHF = cdivc(PIT, 2);   // pieces per issue tap (weights at taps 7 / 8, patches at taps 0 / 1)
    issue_w(IC<0>{}, IC<WIT>{}, 0, 0);
    issue_p(IC<0>{}, IC<PIT>{}, 0, 0, false);
    issue_w(IC<0>{}, IC<WIT>{}, 32, 1);
    issue_p(IC<0>{}, IC<PIT>{}, 32, 1, false);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WIT + PIT) : "memory");     // chunk 0 has landed
    __builtin_amdgcn_s_barrier();

    uint4 wf[3][NJ], xf[3][PT];
    auto read_frags = [&](const char* ws, const char* ps, int tap, uint4 (&w)[NJ], uint4 (&x)[PT]) {
        const int kh = tap / 3, kw = tap % 3;
#pragma unroll
        for (int j = 0; j < NJ; ++j) w[j] = *reinterpret_cast<const uint4*>(ws + wb + (tap * BN + j * 16) * kRow);
#pragma unroll
        for (int i = 0; i < PT; ++i) x[i] = *reinterpret_cast<const uint4*>(ps + xb[i][kw] + kh * PW * kRow);
    };
    read_frags(smem + W0_B, smem + P0_B, 0, wf[0], xf[0]);
    read_frags(smem + W0_B, smem + P0_B, 1, wf[1], xf[1]);

    int qw = 0, qp = 0;                                  // weight slot (chunk index & 1) and patch slot (chunk index % 3) of the current chunk
    // (Tried and measured slower end to end, 512^3 x 12 prediction on one box: transposing the finished tile through LDS - the patch
    // slot its last chunk vacated - so that a lane stores 16 bytes and a wave instruction covers whole 128-byte rows halves the
    // stores and cut the epilogue from 1.8 to 1.1 us per tile in tools/convlab, but the next tile's first patch pieces then
    // have to wait for that slot: 0.559 s against 0.538 s.  A start delay staggered by workgroup: neutral.)
    // (Spreading a finished tile's stores over the next tile's first chunk instead of issuing them back to back was tried:
    // the time only moves from the store issue to the chunk barriers - on the 64 -> 64 layers the sum of input and output
    // traffic, 3.3 TB/s of mixed reads and writes, is what bounds the tile rate - and with streamed weights the stores
    // then sit in front of the next weight pieces, whose wait has to retire them.)
    int st_age = 99;                                     // chunks since the previous tile's stores were issued (0: just before this chunk)
    // One chunk (9 taps) of the current tile; chunk x of the stream (over all tiles of this workgroup):
    //   taps 0, 1: the patch pieces of chunk x+2 into patch slot (x+2) % 3 (vacated by chunk x-1 at its barrier);
    //   tap 7:     wait for everything but those pieces (and but the previous tile's stores, right behind a tile boundary):
    //              chunk x+1 has landed; barrier: every wave's reads of chunk x are done;
    //   taps 7, 8: the weight pieces of chunk x+2 into weight slot x & 1 (vacated just now) - unless the weights are resident.
    // MORE: another chunk follows in the stream
    auto chunk_body = [&](int c, auto more_) {
        constexpr bool more = decltype(more_)::value != 0;
        const char* wcur = smem + W0_B + qw * WGT_B;
        const char* wnxt = smem + W0_B + (qw ^ 1) * WGT_B;
        const char* pcur = smem + P0_B + qp * PATCH_B;
        const int qp1 = qp == 2 ? 0 : qp + 1, qp2 = qp == 0 ? 2 : qp - 1;
        const char* pnxt = smem + P0_B + qp1 * PATCH_B;
        const bool n2 = c + 2 >= nch;                   // chunk x+2 belongs to the next tile
        const int c2 = (n2 ? c + 2 - nch : c + 2) * 32;
        const bool ex2 = more && (!n2 || has_next);     // chunk x+2 exists
        if (c + 1 == nch) {      // the tile's last chunk: its residual quads first - this chunk's tap-7 wait (or the final drain) retires them
            pixel_offsets(n, h0, w0);
            if constexpr (MODE == 1) residual_loads();
        }
        static_for<0, 9>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t < 2) {
                if (ex2) issue_p(IC<minc(t * PHF, PIT)>{}, IC<minc((t + 1) * PHF, PIT)>{}, c2, qp2, n2);
            }
            if constexpr (t == 7 && more) {
                if (PROBE && g.probe) t_a = wall_clock64();
                // Chunk x+1 must have landed.  Issued after its youngest piece (vector-memory operations retire in issue order):
                // the NST stores of the previous tile, if this is the first chunk behind them, then the PIT patch pieces of
                // chunk x+2 (taps 0, 1 of this chunk)
                const bool skip_st = st_age == 0;
                if (skip_st) {
                    if (ex2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NST + PIT) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NST) : "memory");
                } else {
                    if (ex2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PIT) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                if (PROBE && g.probe) t_wait += wall_clock64() - t_a;
            }
            if constexpr (t >= 7 && more) {
                if (ex2 && !wres) issue_w(IC<minc((t - 7) * WH, WIT)>{}, IC<minc((t - 6) * WH, WIT)>{}, c2, qw);
            }
            if constexpr (t + 2 < 9) read_frags(wcur, pcur, t + 2, wf[(t + 2) % 3], xf[(t + 2) % 3]);
            else if constexpr (more) read_frags(wnxt, pnxt, t + 2 - 9, wf[(t + 2) % 3], xf[(t + 2) % 3]);
#pragma unroll
            for (int i = 0; i < PT; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) mma16<T>(acc[i][j], wf[t % 3][j], xf[t % 3][i]);
            if constexpr (PIN == 2) {   // one read, then its share of the MFMAs
                constexpr int NR = (t + 2 < 9 || more) ? NJ + PT : 0, NM = NJ * PT;
                static_for<0, NR>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, (NM * (r + 1)) / NR - (NM * r) / NR, 0);
                });
                if constexpr (NR == 0) __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
            }
        });
        qw ^= 1;
        qp = qp1;
        ++st_age;
    };

    // evaluation-mode epilogue: affine, residual, activation, EXACTLY NST unconditional buffer stores per lane and not one
    // vector load.  DRAIN: no tap-7 wait preceded it (the stream's final chunk): retire the residual loads here
    auto epilogue = [&](auto drain_) {
        constexpr bool drain = decltype(drain_)::value != 0;
        if (PROBE && g.probe) { t_a = wall_clock64(); ++n_tiles; }
        if constexpr (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (MODE == 1)
            static_for<0, PT>([&](auto ic) {      // orders every use of the quads behind the wait that retired them
                constexpr int i = decltype(ic)::value;
                static_for<0, NJ>([&](auto jc) { constexpr int j = decltype(jc)::value; quad_tie(rres[i][j]); });
            });
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float4 sc = *reinterpret_cast<const float4*>(eaff + j * 16 + lq * 4), sh = *reinterpret_cast<const float4*>(eaff + BN + j * 16 + lq * 4);
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (emode & 1) { v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w; }
                else if (emode & 2) { v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w; }
                if constexpr (MODE == 1) {
                    typename Raw4<T>::type rq;
                    if constexpr (std::is_same<T, bf16_t>::value) rq = make_uint2(rres[i][j].x, rres[i][j].y);
                    else rq = f16raw4{make_uint2(rres[i][j].x, rres[i][j].y)};
                    const float4 rv = unpack4(rq);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                }
                if (erelu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                else if (erelu == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.f + __expf(-v[r]));
                }
                const u32x2 pk = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
                __builtin_amdgcn_raw_buffer_store_b64(pk, ro, opix[i] + j * 32, 0, 0);
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (PROBE && g.probe) t_epi += wall_clock64() - t_a;
    };
    if (PROBE && g.probe) t_loop = wall_clock64();

    for (;;) {
        for (int c = 0; c + 1 < nch; ++c) chunk_body(c, IC<1>{});
        if (!has_next) {
            chunk_body(nch - 1, IC<0>{});
            epilogue(IC<1>{});
            break;
        }
        chunk_body(nch - 1, IC<1>{});
        asm volatile("" ::: "memory");
        epilogue(IC<0>{});
        asm volatile("" ::: "memory");
        st_age = 0;
        // the next tile becomes the current one; offsets of the tile after it
        grp += gstep;
        decode(grp, n, h0, w0);
#pragma unroll
        for (int i = 0; i < PIT; ++i) { poff0[i] = noff0[i]; if constexpr (MODE == 2) poff1[i] = noff1[i]; }
        has_next = grp + gstep < g.groups;
        int nn = 0, nh = 0, nw = 0;
        if (has_next) decode(grp + gstep, nn, nh, nw);
        tile_offsets(nn, nh, nw, noff0, noff1);
    }
    if (PROBE && g.probe) {
        __builtin_amdgcn_s_waitcnt(0);
        if (tid == 0) {
            unsigned long long* o = g.probe + (size_t)blockIdx.x * 8;
            o[0] = t_begin; o[1] = t_loop; o[2] = t_wait; o[3] = t_epi; o[4] = wall_clock64(); o[5] = (unsigned long long)n_tiles; o[6] = 0; o[7] = 0x53;
        }
    }
}

template <int BN, int PT, int NW, int TWS>
constexpr size_t stream_lds_bytes() {
    constexpr int NT = NW * 64, BM = NW * PT * 16, TW = 1 << TWS, TH = BM / TW;
    return (size_t)(3 * cdivc((TH + 2) * (TW + 2) * 4, NT) + 2 * cdivc(9 * BN * 4, NT)) * NT * 16 + 2 * BN * sizeof(float);
}

// whether conv_stream_kernel's restrictions hold for this layer (the caller adds its own policy: grid size, option)
inline bool stream_ok(const ConvParams& p, int out_nchw) {
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.dil > 1 || p.gc || p.scatter || out_nchw || p.out_f32 || p.out1 ||
        p.pool0 || p.stats_partial || p.stats_bins || p.bz || (p.Cout & 31) || p.up0 > 1 || (p.scale && !p.shift))
        return false;
    if ((p.C0 & 31) || (p.C1 & 31) || p.C0 + p.C1 < 64) return false;
    if (p.Hout < 16 || p.Wout < 16) return false;
    const double lim = 2.0e9;      // whole-tensor descriptors with 32-bit byte offsets
    const int ush = p.up0 ? 1 : 0;
    return (double)p.N * (p.Hin >> ush) * (p.Win >> ush) * p.C0 * 2.0 < lim && (double)p.N * p.Hin * p.Win * p.C1 * 2.0 < lim &&
           (double)p.N * p.Hout * p.Wout * p.Cout * 2.0 < lim && (double)p.Cout * 9 * (p.C0 + p.C1) * 2.0 < lim;
}

template <typename T, int BN, int PT, int NW, int TWS, int WPS, int PIN = 2>
int launch_stream(const ConvParams& p, unsigned long long* probe, hipStream_t s, int workgroups = 256, int stagger = 0) {
    static bool attr_set[6] = {false, false, false, false, false, false};
    const int mode = p.C1 ? 2 : (p.residual ? 1 : 0);
    VS_REQUIRE(!(p.C1 && p.residual), "conv_stream: a two-tensor input and a residual do not occur together");
    auto kern = mode == 2 ? (probe ? conv_stream_kernel<T, BN, PT, NW, TWS, WPS, PIN, 2, true> : conv_stream_kernel<T, BN, PT, NW, TWS, WPS, PIN, 2, false>)
              : mode == 1 ? (probe ? conv_stream_kernel<T, BN, PT, NW, TWS, WPS, PIN, 1, true> : conv_stream_kernel<T, BN, PT, NW, TWS, WPS, PIN, 1, false>)
                          : (probe ? conv_stream_kernel<T, BN, PT, NW, TWS, WPS, PIN, 0, true> : conv_stream_kernel<T, BN, PT, NW, TWS, WPS, PIN, 0, false>);
    constexpr size_t lds = stream_lds_bytes<BN, PT, NW, TWS>();
    static_assert(lds <= 160 * 1024, "conv_stream: LDS ring too large");
    constexpr int BM = NW * PT * 16, TW = 1 << TWS, TH = BM / TW;
    VS_REQUIRE(stream_ok(p, 0) && p.Cout % BN == 0, "conv_stream: layer outside the persistent kernel's restrictions");
    if (!attr_set[mode * 2 + (probe ? 1 : 0)]) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[mode * 2 + (probe ? 1 : 0)] = true;
    }
    SGeom g{};
    g.tiles_h = cdiv(p.Hout, TH);
    g.tiles_w = cdiv(p.Wout, TW);
    g.groups = p.N * g.tiles_h * g.tiles_w;
    g.ctiles = cdiv(p.Cout, BN);
    g.lanes = std::max(1, std::min(cdiv(g.groups, 8), workgroups / 8 / g.ctiles));
    g.ct_magic = 0xffffffffu / (unsigned)g.ctiles + 1u;
    g.ti_magic = 0xffffffffu / (unsigned)(g.tiles_h * g.tiles_w) + 1u;
    g.tw_magic = 0xffffffffu / (unsigned)g.tiles_w + 1u;
    g.probe = probe;
    g.stagger = stagger;
    VS_REQUIRE(g.tiles_h * g.tiles_w < 65536 && p.N < 65536 && g.groups < (1 << 24), "conv_stream: tile grid too large");
    const int nwg = 8 * g.lanes * g.ctiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NW * 64), lds, s, p, g);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

}  // namespace ring
