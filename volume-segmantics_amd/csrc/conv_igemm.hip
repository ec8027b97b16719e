// Implicit-GEMM NHWC convolution on MFMA (gfx950), im2col-free.  Two kernels:
//
// conv_igemm_kernel - the general one.  A workgroup (8 waves on 16x16-pixel tiles; 4 waves on 8x16 / 8x8 tiles for small
// images and stride 2) computes a tile of output pixels of one image times BN (64 / 32 / 16) output channels.  For every
// chunk of CK input channels (64 bytes per pixel: 32 bf16 / 16 f32) the input halo patch and the BN x taps weight slab are
// staged in LDS once and reused by all KH*KW taps: a tap is a constant byte offset into the staged patch, so there is no
// im2col buffer anywhere.  MFMA orientation: A = weights (rows = cout), B = pixels (cols), so each lane ends up with 4
// consecutive output channels of one pixel -> 8/16-byte NHWC stores.
//   * staging: raw buffer loads (one descriptor per image, 32-bit lane offsets, offset -1 -> zeros) issued one or two chunks
//     ahead into registers, written to LDS between two barriers; compile-time tile geometry for stride 1;
//   * LDS: unpadded 64-byte rows, 16-byte segments XOR-swizzled by the patch column (conflict-free ds_read_b128);
//   * 32-wide tiles: fragments of tap t+1 are read while the MFMAs of tap t run, pinned with sched_group_barrier;
//   * workgroup -> tile map is XCD-aware: cout tiles sharing an input patch run back to back on one XCD's L2;
//   * epilogue (conv_common.h): BN-statistics partials, affine, residual, ReLU, split output, 2x2 sum-pool, fp32 / NCHW.
//
// conv_direct_kernel - the HBM-bound shallow layers (<= 16 couts, Cin within one chunk, large images) and the segmentation
// head: no LDS staging and no barriers, every wave walks down a 16-pixel-wide strip with a ring of input rows held as MFMA
// B fragments loaded straight from global memory; in prediction the head's epilogue writes labels / probabilities / packed
// keys into the output volume itself.
//
// bf16: v_mfma_f32_16x16x32_bf16 (one per 32-channel chunk-tap); f32: 4 x v_mfma_f32_16x16x4_f32 on the same 16-byte
// fragments (exact fp32 FMA chain - the parity path).
//
// Serves every 3x3 / 1x1 convolution of smp.Unet(resnet34) forward (reference call sites vol_seg_2d_trainer.py:424,
// vol_seg_2d_predictor.py:44), with the decoder's nearest-x2 upsample + concat folded into the patch loader, and - fed with
// flipped/transposed weights - their dgrad.
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdlib>

#include "conv_common.h"
#include "conv_stream.h"
#include "prof.h"

namespace {

constexpr int kPS = 64;  // LDS bytes per staged pixel / weight row: unpadded; the 16-byte segment s of a patch pixel in patch
                         // column pw lives in slot s ^ ((pw >> 1) & 3), of weight row r in slot s ^ ((r >> 1) & 3).  With
                         // gfx950's ds_read_b128 lane groups this XOR makes the fragment reads conflict-free (tools/lds_bank_sim.py;
                         // 80-byte padded rows are 2-way conflicted, 96-byte ones conflict-free but 50 % bigger).  Keying the
                         // patch swizzle on the column keeps the kh tap offsets plain constants.
__device__ __forceinline__ int swz(int row, int key, int seg) { return row * kPS + ((seg ^ ((key >> 1) & 3)) << 4); }


struct TileGeom {
    int tw_shift;  // TW = 1 << tw_shift (8 or 16)
    int TH;
    int tiles_h, tiles_w;
    int PH, PW;  // staged patch dims
    int out_nchw;
    unsigned pw_magic, tw_magic;  // x / PW == umulhi(x, pw_magic), x / tiles_w == umulhi(x, tw_magic) for x < 65536
    int groups, ctiles;           // pixel tiles over the whole batch, cout tiles (see the workgroup -> tile map in the kernel)
    unsigned ct_magic, ti_magic;  // x / ctiles, x / (tiles_h * tiles_w)
    unsigned long long* probe;  // phase timestamps (tools/conv_probe.py); null in normal operation
};

// 16-byte patch items per thread (NW = waves per workgroup; the tile has NW*PT*16 pixels)
constexpr int patch_items(int pt, int stride, int nw = 4) {
    return nw == 8 ? 3 : (stride == 2 ? (pt == 2 ? 9 : 5) : (pt == 4 ? 6 : (pt == 2 ? 3 : 2)));
}

// DIL: dilation of a 3x3 kernel (1 or 2; torchvision / smp "replace stride with dilation" stages).  Dilated variants take their
// tile geometry from the launch like the stride-2 ones and use the same staging budget (the patch is (tile + 2 DIL) wide).
// NLOAD: src0 is the PRE-norm output z of the conv -> BN -> ReLU unit in front (ConvParams::nl_*).  The prologue sums the unit's
// statistics bins (every workgroup for itself; workgroup 0 publishes), the chunks are normalised in registers on their way to LDS -
// y = max((z - mean) * (invstd * gamma) + beta, 0) rounded to bf16, the value the normalisation sweep would have stored, zero
// padding left zero - and the workgroups of the first cout tile store the normalised interior of their patch (the activation the
// weight gradient reads later): no normalisation sweep, no launch between the two convolutions.
template <typename T, int BN, int PT, int NTAPS, int STRIDE, int NW = 4, int DIL = 1, bool NLOAD = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : 1) void conv_igemm_kernel(ConvParams p, TileGeom g) {
    constexpr int NT = NW * 64;
    constexpr int CK = CT<T>::CK, EPS = CT<T>::EPS;
    constexpr int NJ = BN / 16, KW = NTAPS == 9 ? 3 : 1;
    constexpr int PITEMS = patch_items(PT, DIL > 1 ? 2 : STRIDE, NW);
    constexpr int WROWS = NT / 4;                 // weight rows one pass of the workgroup stages
    constexpr int TS = WROWS / BN;                // taps per pass
    static_assert(WROWS % BN == 0, "cout tile must divide the rows of a staging pass");
    constexpr int WITEMS = (NTAPS + TS - 1) / TS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    // stride-1 kernels have a compile-time tile (16 wide; 8 wide for the 64-pixel tiles of small images), so patch
    // row offsets are instruction immediates; stride-2 kernels take it from the launch
    constexpr bool kStatic = STRIDE == 1 && DIL == 1;
    constexpr int kTWS = PT == 1 ? 3 : 4, kTW = 1 << kTWS, kTH = NW * 16 * PT / kTW, kKH = NTAPS == 9 ? 3 : 1;
    const int tw_shift = kStatic ? kTWS : g.tw_shift;
    const int TW = 1 << tw_shift;
    const int PW = kStatic ? kTW - 1 + KW : g.PW;
    const int PH = kStatic ? kTH - 1 + kKH : g.PH;
    const int TH = kStatic ? kTH : g.TH;
    const int Cin = p.C0 + p.C1;
    const int P = PH * PW;
    char* patch = smem;
    char* wl = smem + P * kPS;
    // pull every kernel argument the prologue needs into SGPRs with ONE batch of scalar loads: left to itself the compiler
    // loads them where first used, a chain of four dependent ~0.3 us round trips at the head of every workgroup
    asm volatile("" ::"s"(p.src0), "s"(p.src1), "s"(p.w), "s"(p.out), "s"(p.C0), "s"(p.C1), "s"(p.up0), "s"(p.Hin), "s"(p.Win),
                 "s"(p.Hout), "s"(p.Wout), "s"(p.pad), "s"(p.Cout), "s"(g.tiles_w), "s"(g.tw_magic), "s"(g.pw_magic), "s"(g.PW),
                 "s"(g.PH), "s"(g.tw_shift), "s"(g.probe), "s"(p.gc));
    unsigned long long tprobe[5];
    if (g.probe) tprobe[0] = wall_clock64();

    // Workgroup -> tile, XCD-aware: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own L2), so the
    // cout tiles that share one input patch are given ids 8 apart - they run back to back on ONE XCD and the patch comes
    // from HBM once (ids differing only in the low 3 bits take different pixel tiles).  With the cout tile as the slow
    // index every XCD would fetch every image: ~2x the traffic in the 256-/512-channel layers (rocprofv3 FETCH_SIZE).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int gl = g.ctiles == 1 ? slot : (int)__umulhi((unsigned)slot, g.ct_magic);
    const int ytile = slot - gl * g.ctiles;
    const int grp = gl * 8 + xcd;                     // pixel tile index over the whole batch
    if (grp >= g.groups) return;                      // padding of the last round (uniform per workgroup)
    const int tiles_img = g.tiles_h * g.tiles_w;
    const int n = tiles_img == 1 ? grp : (int)__umulhi((unsigned)grp, g.ti_magic);
    const int timg = grp - n * tiles_img;
    const int ty = g.tiles_w == 1 ? timg : (int)__umulhi((unsigned)timg, g.tw_magic);
    const int tx = timg - ty * g.tiles_w;
    const int h0 = ty * TH, w0 = tx * TW;
    const int n0 = ytile * BN;
    const int hbase = h0 * STRIDE - p.pad, wbase = w0 * STRIDE - p.pad;
    // up0: 0 = plain source, 1 = nearest x2 upsampling of a half-resolution source, 2 = the same addressing with zeros at
    // the odd rows / columns (zero stuffing: data gradient of a stride-2 convolution, never materialised)
    const int ush = p.up0 ? 1 : 0;
    const bool stuffed = p.up0 == 2;
    const int H0 = p.Hin >> ush, W0 = p.Win >> ush;
    // Staging loads are raw buffer loads: one descriptor per source (base = this image, so per-lane offsets are 32-bit
    // byte offsets), the chunk's channel offset rides in the scalar offset, and offset -1 (out of range) returns zeros -
    // zero padding, ragged tiles and channel tails cost no branches.
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)p.src0 + (size_t)n * H0 * W0 * p.C0), 0, H0 * W0 * p.C0 * (int)sizeof(T), 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)p.src1 + (size_t)n * p.Hin * p.Win * p.C1), 0, p.src1 ? p.Hin * p.Win * p.C1 * (int)sizeof(T) : 0, 0x00020000);
    // Grouped convolution (p.gc = 32, BN = 32): the cout tile IS one 32-channel super-group, whose outputs read only the
    // super-group's own 32 input channels: the chunk loop covers [n0, n0 + 32) and the weight rows are 32 channels long
    // (groups narrower than 32 channels are block-diagonal inside the super-group's 32 x 32 slab, zeros elsewhere).
    const int kbeg = p.gc ? n0 : 0, kend = p.gc ? n0 + p.gc : Cin;
    const int CinW = p.gc ? p.gc : Cin;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.Cout * NTAPS * CinW * (int)sizeof(T), 0x00020000);
    const int dummy = (P + NTAPS * BN) * kPS;    // 64 spare bytes behind the staged tiles: target of the stores of idle items
    // NLOAD: [3][C0 rounded up to 32] floats behind them - mean, invstd * gamma, beta of src0's channels
    float* cst = reinterpret_cast<float*>(smem + dummy + 64);
    const int C0r = (p.C0 + 31) & ~31;
    if constexpr (NLOAD) {
        const long long* bins = reinterpret_cast<const long long*>(p.nl_bins);     // [nl_nb][2][C0]: sum z * 2^24, sum z^2 * 2^16
        const double rows = (double)p.nl_rows;
        for (int ch = tid; ch < p.C0; ch += NT) {
            long long sv = 0, qv = 0;
#pragma unroll 8
            for (int r = 0; r < p.nl_nb; ++r) { sv += bins[((size_t)r * 2 + 0) * p.C0 + ch]; qv += bins[((size_t)r * 2 + 1) * p.C0 + ch]; }
            // (the arithmetic of bn_apply_inline_kernel<T, true>: the same bits as the sweep's statistics)
            const double mu = ((double)sv * (1.0 / kStatScale1)) / rows;
            double var = ((double)qv * (1.0 / kStatScale2)) / rows - mu * mu;
            if (var < 0.0) var = 0.0;
            const float is = (float)(1.0 / sqrt(var + (double)p.nl_eps));
            cst[ch] = (float)mu;
            cst[C0r + ch] = is * p.nl_gamma[ch];
            cst[2 * C0r + ch] = p.nl_beta[ch];
            if (blockIdx.x == 0) {        // one workgroup publishes: statistics for the backward pass, running statistics
                p.nl_mean[ch] = (float)mu;
                p.nl_invstd[ch] = is;
                if (p.nl_rm) {
                    const double unbiased = p.nl_rows > 1 ? var * rows / (double)(p.nl_rows - 1) : var;
                    p.nl_rm[ch] = (float)((1.0 - p.nl_mom) * (double)p.nl_rm[ch] + p.nl_mom * mu);
                    p.nl_rv[ch] = (float)((1.0 - p.nl_mom) * (double)p.nl_rv[ch] + p.nl_mom * unbiased);
                }
            }
        }   // visible to every wave behind the first barrier of the chunk loop
    }

    // ---- chunk-invariant staging addresses (byte offsets; -1 = zero fill) ----
    int poff0[PITEMS], poff1[PITEMS], pdst[PITEMS];
    unsigned ystore = 0;
#pragma unroll
    for (int i = 0; i < PITEMS; ++i) {
        const int item = tid + i * NT;
        const int pp = item >> 2, seg = item & 3;
        const int ph = kStatic ? pp / PW : (int)__umulhi((unsigned)pp, g.pw_magic), pw = pp - ph * PW;
        const int hi = hbase + ph, wi = wbase + pw;
        const bool ok = pp < P && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win;
        poff0[i] = (ok && !(stuffed && ((hi | wi) & 1))) ? (((hi >> ush) * W0 + (wi >> ush)) * p.C0 + seg * EPS) * (int)sizeof(T) : -1;
        poff1[i] = ok ? ((hi * p.Win + wi) * p.C1 + seg * EPS) * (int)sizeof(T) : -1;
        pdst[i] = pp < P ? swz(pp, pw, seg) : dummy;
        if constexpr (NLOAD) {   // items whose normalised value this workgroup stores to nl_y: the tile's own pixels (not the halo), once per source pixel
            const bool mine = ok && hi >= h0 && hi < h0 + TH && wi >= w0 && wi < w0 + TW && !(ush && ((hi | wi) & 1));
            ystore |= mine ? 1u << i : 0u;
        }
    }
    // weight staging: pass i covers rows i*WROWS + (tid >> 2); row = tap*BN + nr, so a pass advances TS taps
    const int wrow0 = tid >> 2, wseg = tid & 3;
    const int wnr = wrow0 % BN, wtap0 = wrow0 / BN;
    const bool wok = n0 + wnr < p.Cout;
    const int woff0 = (((n0 + wnr) * NTAPS + wtap0) * CinW + wseg * EPS) * (int)sizeof(T);
    const int wdst0 = swz(wrow0, wrow0, wseg);   // i*WROWS is a multiple of 16 rows: same swizzle in every pass

    // Prefetch depth: PF chunks are in flight (in registers) while one is being multiplied.  Kernels that run with one or
    // two workgroups per CU and have registers to spare use 2, so a staging load gets two chunk periods to land.
    constexpr int PF = (((NW == 8 && BN <= 32) || (NW == 4 && PT == 1 && STRIDE == 1)) && !(NLOAD && NW == 8)) ? 2 : 1;   // (NLOAD, 8 waves: the second set spills)
    uint4 pregs[PF][PITEMS], wregs[PF][WITEMS];
    auto load_chunk = [&](int c0, uint4 (&preg)[PITEMS], uint4 (&wreg)[WITEMS]) {
        const bool from0 = c0 < p.C0;
        const int cs = from0 ? p.C0 : p.C1;
        const int cb = from0 ? c0 : c0 - p.C0;
        const int nseg = (cs - cb) / EPS;         // valid 16-byte segments of this chunk (< 4 only in a ragged channel tail)
        const int nsegw = (kend - c0) / EPS;
#pragma unroll
        for (int i = 0; i < PITEMS; ++i) {
            int off = from0 ? poff0[i] : poff1[i];
            if (((tid + i * NT) & 3) >= nseg) off = -1;
            preg[i] = from0 ? bload(r0, off, cb * (int)sizeof(T)) : bload(r1, off, cb * (int)sizeof(T));
        }
#pragma unroll
        for (int i = 0; i < WITEMS; ++i) {
            int off = woff0 + i * TS * CinW * (int)sizeof(T);
            if (!wok || wtap0 + i * TS >= NTAPS || wseg >= nsegw) off = -1;
            wreg[i] = bload(rw, off, (c0 - kbeg) * (int)sizeof(T));
        }
    };

    // per-lane LDS read bases: pixel tile i, tap column kw (tap row kh adds the constant kh * PW * kPS)
    int xb[PT][KW];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int pl = wave * (PT * 16) + i * 16 + lr;
        const int th = pl >> tw_shift, tw = pl & (TW - 1);
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) xb[i][kw] = swz((th * STRIDE) * PW + tw * STRIDE + kw * DIL, tw * STRIDE + kw * DIL, lq);
    }
    const int wbase_l = swz(lr, lr, lq);                  // (tap*BN + 16j) is a multiple of 16: it does not change the swizzle
    const int khs = DIL * PW * kPS;

    f32x4 acc[PT][NJ];
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_chunk(kbeg, pregs[0], wregs[0]);
    if (PF == 2 && kbeg + CK < kend) load_chunk(kbeg + CK, pregs[PF - 1], wregs[PF - 1]);
    if (g.probe) tprobe[1] = wall_clock64();
    for (int cbase = kbeg; cbase < kend; cbase += PF * CK) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int c0 = cbase + s * CK;
            if (c0 >= kend) break;
            __syncthreads();  // every wave is done reading the previous chunk
            if constexpr (NLOAD) {
                if (c0 < p.C0) {   // (uniform) a chunk of src0: this thread's 8 channels are the same in all of its items
                    const float* q = cst + c0 + (tid & 3) * EPS;
                    float nm[8], na[8], nb[8];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 m4 = *reinterpret_cast<const float4*>(q + 4 * h), a4 = *reinterpret_cast<const float4*>(q + C0r + 4 * h),
                                     b4 = *reinterpret_cast<const float4*>(q + 2 * C0r + 4 * h);
                        nm[4 * h] = m4.x; nm[4 * h + 1] = m4.y; nm[4 * h + 2] = m4.z; nm[4 * h + 3] = m4.w;
                        na[4 * h] = a4.x; na[4 * h + 1] = a4.y; na[4 * h + 2] = a4.z; na[4 * h + 3] = a4.w;
                        nb[4 * h] = b4.x; nb[4 * h + 1] = b4.y; nb[4 * h + 2] = b4.z; nb[4 * h + 3] = b4.w;
                    }
                    const bool segok = (tid & 3) * EPS < p.C0 - c0;
#pragma unroll
                    for (int i = 0; i < PITEMS; ++i) pregs[s][i] = nl_apply8(pregs[s][i], segok && poff0[i] >= 0, nm, na, nb);
                    if (p.nl_y && ytile == (c0 / CK) % g.ctiles) {     // (uniform) the activation tensor as a by-product, the chunks dealt round-robin to the
                                                                       // cout tiles of this pixel tile; out-of-range offsets are dropped by the hardware
                        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
                            (void*)((T*)p.nl_y + (size_t)n * H0 * W0 * p.C0), 0, H0 * W0 * p.C0 * (int)sizeof(T), 0x00020000);
#pragma unroll
                        for (int i = 0; i < PITEMS; ++i) {
                            const uint4 v = pregs[s][i];
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, ry, (segok && ((ystore >> i) & 1u)) ? poff0[i] : (int)0x80000000,
                                                                   c0 * (int)sizeof(T), 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < PITEMS; ++i) *reinterpret_cast<uint4*>(patch + pdst[i]) = pregs[s][i];
#pragma unroll
            for (int i = 0; i < WITEMS; ++i) {
                const bool full = (i + 1) * TS <= NTAPS;   // every row of this pass is a real tap
                const int dst = (full || wtap0 + i * TS < NTAPS) ? P * kPS + wdst0 + i * WROWS * kPS : dummy;
                *reinterpret_cast<uint4*>(smem + dst) = wregs[s][i];
            }
            __syncthreads();
            if (g.probe && c0 == kbeg) tprobe[2] = wall_clock64();
            if (c0 + PF * CK < kend) load_chunk(c0 + PF * CK, pregs[s], wregs[s]);  // in flight while the MFMAs of PF chunks run
            // fragments of tap t+1 are read while the MFMAs of tap t run (two register sets)
            uint4 wf[2][NJ], xf[2][PT];
            auto read_frags = [&](int tap, uint4 (&w)[NJ], uint4 (&x)[PT]) {
                const int kh = tap / KW, kw = tap % KW;
#pragma unroll
                for (int j = 0; j < NJ; ++j) w[j] = *reinterpret_cast<const uint4*>(wl + (tap * BN + j * 16) * kPS + wbase_l);
#pragma unroll
                for (int i = 0; i < PT; ++i) x[i] = *reinterpret_cast<const uint4*>(patch + xb[i][kw] + kh * khs);
            };
            constexpr int kMfmaPerTap = NJ * PT * (sizeof(T) == 2 ? 1 : 4);
            read_frags(0, wf[0], xf[0]);
            constexpr bool kPin = BN <= 32;   // 64-wide tiles run out of registers under the pinned order (measured slower)
            if (kPin) __builtin_amdgcn_sched_group_barrier(0x100, NJ + PT, 0);
#pragma unroll
            for (int tap = 0; tap < NTAPS; ++tap) {
                if (tap + 1 < NTAPS) read_frags(tap + 1, wf[(tap + 1) & 1], xf[(tap + 1) & 1]);
#pragma unroll
                for (int i = 0; i < PT; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) mma16<T>(acc[i][j], wf[tap & 1][j], xf[tap & 1][i]);
                // pin the software pipeline: the LDS reads of tap t+1 issue before the MFMAs of tap t (left alone, the
                // scheduler serialises read -> wait -> MFMA to save registers)
                if (kPin && tap + 1 < NTAPS) __builtin_amdgcn_sched_group_barrier(0x100, NJ + PT, 0);
                if (kPin) __builtin_amdgcn_sched_group_barrier(0x008, kMfmaPerTap, 0);
            }
        }
    }

    if (g.probe) tprobe[3] = wall_clock64();
    conv_epilogue<T, BN, PT, NW>(p, tw_shift, g.out_nchw, n, h0, w0, n0, grp, acc, smem);
    if (g.probe) {
        __builtin_amdgcn_s_waitcnt(0);  // stores issued and acknowledged
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

// ---- shallow layers: all of Cin in one chunk, Cout <= 16 -----------------------------------------------------------
// The full-resolution decoder layers (-> 16 channels at H x W), the segmentation head and the head's data gradient
// are HBM-bound, with so little matrix work per pixel that the LDS-staged tile kernel above spends its time in
// barriers and load latency.  Here every wave is on its own: it owns a 16-pixel-wide column strip of RH output rows and
// walks down it with a rolling window of input rows held as MFMA B fragments that are loaded STRAIGHT from global memory
// (one 16-byte buffer load per lane: pixel = column + kw - 1, channels 8*(lane>>4)..; padding = out-of-range offset =
// zeros).  Each new output row costs 3 loads (the next input row at the three kw shifts; the 3x reuse across kh lives in
// registers, the rest in L1), 9*NJ MFMAs against the weight slab (staged once per workgroup in LDS) and its stores.  No
// barriers in the loop, many independent waves per SIMD.
struct DirectGeom {
    int strips_w, chunks_h, RH;      // column strips per row, row chunks per image, rows per chunk
    unsigned sw_magic, ch_magic;     // x / strips_w, x / chunks_h by umulhi
    int nwaves, out_nchw;
    int xcd;                         // XCD-local strip order (common.h: xcd_block)
};

// MODE 0: NHWC store in T (optional per-channel affine + ReLU, optional statistics); 1: 2x2 sum-pooled NHWC store (dgrad
// through nearest-x2 upsampling).  (The segmentation head has its own kernel below.)
// Every row iteration issues the same number of loads and (offset-masked, never skipped) buffer stores, so the waits on
// the input ring are counted ones and kD rows stay in flight per wave.
template <typename T, int BN, int MODE>
__global__ __launch_bounds__(256, 4) void conv_direct_kernel(ConvParams p, DirectGeom g) {
    constexpr int EPS = CT<T>::EPS, NJ = BN / 16, NTAPS = 9, kD = (BN == 32 && MODE == 1) ? 2 : ((BN == 16 && MODE == 0) ? 2 : 4);   // input rows in flight
    constexpr int WTOTAL = NTAPS * BN * 4, WITEMS = (WTOTAL + 255) / 256;
    constexpr int kOob = (int)0x80000000;
    __shared__ __attribute__((aligned(16))) char wl[NTAPS * BN * kPS + 64];
    asm volatile("" ::"s"(p.src0), "s"(p.w), "s"(p.out), "s"(p.C0), "s"(p.up0), "s"(p.Hin), "s"(p.Win), "s"(p.Hout), "s"(p.Wout),
                 "s"(p.Cout), "s"(p.scale), "s"(p.shift), "s"(p.relu), "s"(p.stats_partial), "s"(g.strips_w), "s"(g.chunks_h),
                 "s"(g.RH), "s"(g.sw_magic), "s"(g.ch_magic), "s"(g.nwaves));   // one batch of kernel-argument loads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int Cin = p.C0;
    {   // weights: once per workgroup
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.Cout * NTAPS * Cin * (int)sizeof(T));
#pragma unroll
        for (int i = 0; i < WITEMS; ++i) {
            const int item = tid + i * 256;
            const int row = item >> 2, seg = item & 3;
            const int tap = row / BN, nr = row % BN;
            const bool ok = item < WTOTAL && nr < p.Cout && seg * EPS < Cin;
            const uint4 v = bload(rw, ok ? ((nr * NTAPS + tap) * Cin + seg * EPS) * (int)sizeof(T) : -1, 0);
            *reinterpret_cast<uint4*>(wl + (item < WTOTAL ? swz(row, row, seg) : NTAPS * BN * kPS)) = v;
        }
    }
    __syncthreads();
    const int gw = xcd_block(g.xcd) * 4 + wave;             // this wave's strip
    if (gw >= g.nwaves) return;
    const int q = g.strips_w == 1 ? gw : (int)__umulhi((unsigned)gw, g.sw_magic);
    const int ws = gw - q * g.strips_w;
    const int n = g.chunks_h == 1 ? q : (int)__umulhi((unsigned)q, g.ch_magic);
    const int hc = q - n * g.chunks_h;
    const int w0 = ws * 16, h0 = hc * g.RH, h1 = min(p.Hout, h0 + g.RH);
    const int H0 = p.Hin >> p.up0, W0 = p.Win >> p.up0;
    const int rowbytes = W0 * p.C0 * (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc((const T*)p.src0 + (size_t)n * H0 * W0 * p.C0, H0 * rowbytes);
    // per-lane column offsets of the three kw shifts (bytes inside an input row; -1 = zero: padding column / channel tail)
    int coff[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int col = w0 + lr + kw - 1;
        coff[kw] = (col >= 0 && col < p.Win && lq * EPS < Cin) ? ((col >> p.up0) * p.C0 + lq * EPS) * (int)sizeof(T) : -1;
    }
    const int wbase_l = swz(lr, lr, lq);
    auto load_row = [&](int hi, uint4 (&x)[3]) {          // input row hi at the three kw shifts
        const bool ok = hi >= 0 && hi < p.Hin;
        const int soff = ok ? (hi >> p.up0) * rowbytes : 0;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) x[kw] = bload(rx, ok ? coff[kw] : -1, soff);
    };

    // output addressing: one descriptor per image, per-lane byte offset of (row 0, this lane's column, its first cout)
    const int wo = w0 + lr;
    const bool inw = wo < p.Wout;
    __amdgpu_buffer_rsrc_t ro;
    int ooff[NJ], orow;                                      // lane offset inside an output row; bytes per output row
    if constexpr (MODE == 0) {
        ro = make_rsrc((T*)p.out + (size_t)n * p.Hout * p.Wout * p.Cout, p.Hout * p.Wout * p.Cout * (int)sizeof(T));
        orow = p.Wout * p.Cout * (int)sizeof(T);
#pragma unroll
        for (int j = 0; j < NJ; ++j) ooff[j] = (inw && j * 16 + lq * 4 < p.Cout) ? (wo * p.Cout + j * 16 + lq * 4) * (int)sizeof(T) : kOob;
    } else {
        const int Ho = p.Hout >> 1, Wo = p.Wout >> 1;
        ro = make_rsrc((T*)p.out + (size_t)n * Ho * Wo * p.Cout, Ho * Wo * p.Cout * (int)sizeof(T));
        orow = Wo * p.Cout * (int)sizeof(T);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            ooff[j] = (inw && !(lr & 1) && j * 16 + lq * 4 < p.Cout) ? ((wo >> 1) * p.Cout + j * 16 + lq * 4) * (int)sizeof(T) : kOob;
    }
    float4 sc[NJ], sh[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = j * 16 + lq * 4;
        sc[j] = make_float4(1.f, 1.f, 1.f, 1.f); sh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE != 1) {
            float a[4] = {1.f, 1.f, 1.f, 1.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (c + r < p.Cout) { if (p.scale) a[r] = p.scale[c + r]; if (p.shift) b[r] = p.shift[c + r]; }
            sc[j] = make_float4(a[0], a[1], a[2], a[3]); sh[j] = make_float4(b[0], b[1], b[2], b[3]);
        }
    }
    const bool affine = MODE != 1 && (p.scale || p.shift);
    float s1[NJ][4], s2[NJ][4];
    f32x4 prev[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        prev[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[j][r] = s2[j][r] = 0.f;
    }

    // Input-row major: input row hi feeds output rows hi+1 (kh = 0), hi (kh = 1) and hi-1 (kh = 2), whose accumulators
    // a2, a1, a0 roll down by one slot per row; a ring of kD input rows is in flight.
    f32x4 a0[NJ], a1[NJ], a2[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) a0[j] = a1[j] = a2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 ring[kD][3];
#pragma unroll
    for (int u = 0; u < kD; ++u) load_row(h0 - 1 + u, ring[u]);
    for (int base = h0 - 1; base <= h1; base += kD) {
#pragma unroll
        for (int u = 0; u < kD; ++u) {
            const int hi = base + u;
            if (hi > h1) break;
            int wb = wbase_l;
            if (NJ > 1 || MODE != 0) asm volatile("" : "+v"(wb));   // keep the weight fragments in LDS, not in 72 hoisted registers (16 couts: 36 registers, hoisted)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const uint4 w0f = *reinterpret_cast<const uint4*>(wl + ((0 * 3 + kw) * BN + j * 16) * kPS + wb);
                    const uint4 w1f = *reinterpret_cast<const uint4*>(wl + ((1 * 3 + kw) * BN + j * 16) * kPS + wb);
                    const uint4 w2f = *reinterpret_cast<const uint4*>(wl + ((2 * 3 + kw) * BN + j * 16) * kPS + wb);
                    mma16<T>(a2[j], w0f, ring[u][kw]);
                    mma16<T>(a1[j], w1f, ring[u][kw]);
                    mma16<T>(a0[j], w2f, ring[u][kw]);
                }
            load_row(hi + kD, ring[u]);
            const int h = hi - 1;                              // this output row is complete now
            const bool live = h >= h0;                         // (h < h1 always: hi <= h1)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (MODE == 0 && (p.stats_partial || p.stats_bins) && live && inw) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s1[j][r] += a0[j][r]; s2[j][r] += a0[j][r] * a0[j][r]; }
                }
                float v[4] = {a0[j][0], a0[j][1], a0[j][2], a0[j][3]};
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] += prev[j][r]; v[r] += __shfl_xor(v[r], 1, 64); }
                    prev[j] = a0[j];
                    const int off = (live && (h & 1)) ? ooff[j] : kOob;
                    uint2 pk;
                    if constexpr (sizeof(T) == 2) {
                        pk.x = pack2<T>(v[0], v[1]);
                        pk.y = pack2<T>(v[2], v[3]);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned int, pk), ro, off, (h >> 1) * orow, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, ro, off, (h >> 1) * orow, 0);
                    }
                } else {
                    if (affine) {
                        v[0] = v[0] * sc[j].x + sh[j].x; v[1] = v[1] * sc[j].y + sh[j].y;
                        v[2] = v[2] * sc[j].z + sh[j].z; v[3] = v[3] * sc[j].w + sh[j].w;
                    }
                    if (p.relu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    else if (p.relu == 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.f + __expf(-v[r]));
                    }
                    if constexpr (MODE == 0) {
                        const int off = live ? ooff[j] : kOob;
                        if constexpr (sizeof(T) == 2) {
                            uint2 pk;
                            pk.x = pack2<T>(v[0], v[1]);
                            pk.y = pack2<T>(v[2], v[3]);
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned int, pk), ro, off, h * orow, 0);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, ro, off, h * orow, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) { a0[j] = a1[j]; a1[j] = a2[j]; a2[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
    }
    if (MODE == 0 && (p.stats_partial || p.stats_bins)) {   // one partial row per wave: sum / sum of squares over the strip's pixels (or fixed-point bins)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s1[j][r] += __shfl_xor(s1[j][r], o, 64); s2[j][r] += __shfl_xor(s2[j][r], o, 64); }
                const int c = j * 16 + lq * 4 + r;
                if (lr == 0 && c < p.Cout) {
                    if (p.stats_bins) {
                        unsigned long long* b = p.stats_bins + (size_t)(gw & (p.stats_nb - 1)) * 2 * p.Cout + c;
                        atomicAdd(b, (unsigned long long)__double2ll_rn((double)s1[j][r] * kStatScale1));
                        atomicAdd(b + p.Cout, (unsigned long long)__double2ll_rn((double)s2[j][r] * kStatScale2));
                    } else {
                        p.stats_partial[((size_t)gw * 2 + 0) * p.Cout + c] = s1[j][r];
                        p.stats_partial[((size_t)gw * 2 + 1) * p.Cout + c] = s2[j][r];
                    }
                }
            }
    }
}

// ---- the segmentation head's data gradient, straight from the loss gradient's planes ------------------------------------------
// da[y][x][ci] = sum over (window row r, column s, class k) of W[k][r][s][ci] * dl[k][y + 1 - r][x + 1 - s] with dl = dLoss / dlogits as
// autograd hands it over: fp32 NCHW planes of a FEW classes.  The strip kernel above wants it as a 16-channel NHWC tensor - a
// conversion launch (67 MB written, 14 of its 16 channels zero at two classes) on the critical path between the forward and the
// backward pass, then nine K = 32 MFMAs per 16 pixels that multiply mostly zeros.  Here the (tap, class) pairs ARE the K index:
// kk = 9 * k' .. with tap = kk / classes, class = kk % classes, 9 * classes <= 64 -> one or two MFMAs per 16 pixels, the B fragment
// gathered by each lane from the planes (8 scalar loads, L1 / L2 hits: every element is wanted by nine pixels), rounded to the
// storage type exactly as the conversion would have rounded it.  A wave owns a 16-column strip of RH rows, no LDS, no barriers.
// (Reference: the head's share of loss.backward(), vol_seg_2d_trainer.py:429.)
template <typename T, int NK>
__global__ __launch_bounds__(256, 4) void head_dgrad_planes_kernel(const float* __restrict__ dl, const T* __restrict__ w, T* __restrict__ out, int N, int classes,
                                                                  int H, int W, int C, DirectGeom g) {
    constexpr int kOob = (int)0x80000000;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int gw = blockIdx.x * 4 + wave;
    if (gw >= g.nwaves) return;
    const int q = g.strips_w == 1 ? gw : (int)__umulhi((unsigned)gw, g.sw_magic);
    const int ws = gw - q * g.strips_w;
    const int n = g.chunks_h == 1 ? q : (int)__umulhi((unsigned)q, g.ch_magic);
    const int hc = q - n * g.chunks_h;
    const int x = ws * 16 + lr, h0 = hc * g.RH, h1 = min(H, h0 + g.RH);
    const int KK = 9 * classes;
    // A fragments: A[m = ci = lr][kk = 32 qq + 8 lq + j] = W[class][tap][ci] (w: [classes][9][C], the forward's copy)
    uint4 af[NK];
    int poff[NK][8];          // per K slot: offset of its plane element relative to (class plane 0, row y, column x); kOob = no such slot / column outside
    int prow[NK][8];          // its window row r (the row bound check varies with y)
#pragma unroll
    for (int qq = 0; qq < NK; ++qq) {
        unsigned short a8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 32 * qq + 8 * lq + j;
            const bool live = kk < KK;
            const int t = live ? kk / classes : 0, k = live ? kk - t * classes : 0;
            const int r = t / 3, sft = t - 3 * r;
            a8[j] = (live && lr < C) ? __builtin_bit_cast(unsigned short, w[((size_t)k * 9 + t) * C + lr]) : (unsigned short)0;
            const int xx = x + 1 - sft;
            poff[qq][j] = (live && xx >= 0 && xx < W) ? k * H * W + (1 - r) * W + xx : kOob;
            prow[qq][j] = r;
        }
        af[qq] = make_uint4((unsigned)a8[0] | ((unsigned)a8[1] << 16), (unsigned)a8[2] | ((unsigned)a8[3] << 16),
                            (unsigned)a8[4] | ((unsigned)a8[5] << 16), (unsigned)a8[6] | ((unsigned)a8[7] << 16));
    }
    const float* dn = dl + (size_t)n * classes * H * W;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out + (size_t)n * H * W * C, H * W * C * (int)sizeof(T));
    const int ooff = (x < W && lq * 4 < C) ? (x * C + lq * 4) * (int)sizeof(T) : kOob;
    auto gather = [&](int y, float (&v)[NK][8]) {
        bool rok[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { const int yy = y + 1 - r; rok[r] = yy >= 0 && yy < H && y < h1; }
#pragma unroll
        for (int qq = 0; qq < NK; ++qq)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = poff[qq][j] != kOob && rok[prow[qq][j]];
                v[qq][j] = ok ? dn[(size_t)y * W + poff[qq][j]] : 0.f;
            }
    };
    float cur[NK][8], nxt[NK][8];
    gather(h0, cur);
    for (int y = h0; y < h1; ++y) {
        gather(y + 1, nxt);                                  // in flight while this row's MFMAs and store run
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int qq = 0; qq < NK; ++qq) {
            uint4 b;
            b.x = pack2<T>(cur[qq][0], cur[qq][1]); b.y = pack2<T>(cur[qq][2], cur[qq][3]);
            b.z = pack2<T>(cur[qq][4], cur[qq][5]); b.w = pack2<T>(cur[qq][6], cur[qq][7]);
            mma16<T>(acc, af[qq], b);
        }
        uint2 pk;
        pk.x = pack2<T>(acc[0], acc[1]);
        pk.y = pack2<T>(acc[2], acc[3]);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned int, pk), ro, ooff, y * W * C * (int)sizeof(T), 0);
#pragma unroll
        for (int qq = 0; qq < NK; ++qq)
#pragma unroll
            for (int j = 0; j < 8; ++j) cur[qq][j] = nxt[qq][j];
    }
}

// ---- two chained shallow layers in ONE launch (evaluation mode) ---------------------------------------------------------------
// smp's last decoder block at full resolution - Conv2dReLU(up(x), 32 -> 16) then Conv2dReLU(16 -> 16), BatchNorm folded into
// scale / shift (decoders/unet/decoder.py: DecoderBlock.conv1 / conv2, run by model(batch) at vol_seg_2d_predictor.py:44) - moves
// 1.6 + 2.1 GB per 128 x 512^2 batch as two strip-kernel launches, all but 0.5 + 1.1 GB of it the 16-channel tensor between them.
// Here a wave rolls BOTH layers down its strip: every row of the first layer's output goes - affine, ReLU, rounded to the storage
// type exactly as the first launch would have stored it - into a 16-pixel LDS row of the wave's own, from which the second layer
// reads its three column shifts as MFMA B fragments; the intermediate tensor never exists.  Strips are 14 output columns wide: lane
// lr of the 16-lane rows holds column w0 - 1 + lr of the FIRST layer (the one-pixel halo the second layer's 3 x 3 window needs), the
// second layer's lanes 0 and 15 compute nothing that is stored.  Rows / columns of the first layer's output outside the image are
// written as zeros - they are the second layer's zero padding.  Per output value the products and their order (kh-major taps,
// K = 32 MFMA k-steps over the same channel -> k mapping) are those of conv_direct_kernel / conv_igemm_kernel: bit-identical results
// (tests/test_hip_ops.py::test_direct_pair_equals_two_launches_bit_for_bit).  16-bit storage types only.  Measured: 1 034 us against 2 x 562 us
// per 128 x 512^2 batch, 0.8 % of a 512^3 prediction - the strip walk is bound by its MFMA + LDS-fragment issue (18 MFMAs and 22 LDS
// operations per 16-pixel row and wave), not by the 2.1 GB of HBM traffic the fusion removes.
template <typename T>
__global__ __launch_bounds__(256, 4) void conv_direct_pair_kernel(ConvParams p, ConvParams q, DirectGeom g) {
    constexpr int EPS = CT<T>::EPS, NTAPS = 9, BN = 16, kD = 4;
    constexpr int WTOTAL = NTAPS * BN * 4, WITEMS = (WTOTAL + 255) / 256;
    constexpr int kOob = (int)0x80000000;
    constexpr int kRowB = 18 * 32;                                   // one first-layer row of a wave: 16 pixels + a zero pixel each side, 16 channels
    __shared__ __attribute__((aligned(16))) char wl1[NTAPS * BN * kPS + 64];
    __shared__ __attribute__((aligned(16))) char wl2[NTAPS * BN * kPS + 64];
    __shared__ __attribute__((aligned(16))) char ybuf[4 * kRowB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int Cin = p.C0, Cmid = p.Cout;
    {   // both weight slabs: once per workgroup (conv_direct_kernel's staging, rows = taps x 16 couts, 16-byte channel segments)
        const __amdgpu_buffer_rsrc_t rw1 = make_rsrc(p.w, p.Cout * NTAPS * Cin * (int)sizeof(T));
        const __amdgpu_buffer_rsrc_t rw2 = make_rsrc(q.w, q.Cout * NTAPS * Cmid * (int)sizeof(T));
#pragma unroll
        for (int i = 0; i < WITEMS; ++i) {
            const int item = tid + i * 256;
            const int row = item >> 2, seg = item & 3;
            const int tap = row / BN, nr = row % BN;
            const bool ok1 = item < WTOTAL && nr < p.Cout && seg * EPS < Cin, ok2 = item < WTOTAL && nr < q.Cout && seg * EPS < Cmid;
            const uint4 v1 = bload(rw1, ok1 ? ((nr * NTAPS + tap) * Cin + seg * EPS) * (int)sizeof(T) : -1, 0);
            const uint4 v2 = bload(rw2, ok2 ? ((nr * NTAPS + tap) * Cmid + seg * EPS) * (int)sizeof(T) : -1, 0);
            const int dst = item < WTOTAL ? swz(row, row, seg) : NTAPS * BN * kPS;
            *reinterpret_cast<uint4*>(wl1 + dst) = v1;
            *reinterpret_cast<uint4*>(wl2 + dst) = v2;
        }
        char* yb = ybuf + wave * kRowB;                              // the halo pixels of this wave's row stay zero for good
        if (lane < 4) *reinterpret_cast<uint4*>(yb + (lane & 1) * 16 + (lane >> 1) * 17 * 32) = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    const int gw = blockIdx.x * 4 + wave;                            // this wave's strip
    if (gw >= g.nwaves) return;
    const int qq = g.strips_w == 1 ? gw : (int)__umulhi((unsigned)gw, g.sw_magic);
    const int ws = gw - qq * g.strips_w;
    const int n = g.chunks_h == 1 ? qq : (int)__umulhi((unsigned)qq, g.ch_magic);
    const int hc = qq - n * g.chunks_h;
    const int w0 = ws * 14 - 1, h0 = hc * g.RH, h1 = min(q.Hout, h0 + g.RH);      // lane lr: column w0 + lr of BOTH layers' outputs
    const int H0 = p.Hin >> p.up0, W0 = p.Win >> p.up0;
    const int rowbytes = W0 * p.C0 * (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc((const T*)p.src0 + (size_t)n * H0 * W0 * p.C0, H0 * rowbytes);
    int coff[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int col = w0 + lr + kw - 1;
        coff[kw] = (col >= 0 && col < p.Win && lq * EPS < Cin) ? ((col >> p.up0) * p.C0 + lq * EPS) * (int)sizeof(T) : -1;
    }
    const int wbase_l = swz(lr, lr, lq);
    auto load_row = [&](int hi, uint4 (&x)[3]) {
        const bool ok = hi >= 0 && hi < p.Hin;
        const int soff = ok ? (hi >> p.up0) * rowbytes : 0;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) x[kw] = bload(rx, ok ? coff[kw] : -1, soff);
    };
    // the second layer's output: one descriptor per image, lane offset inside a row (lanes 0 / 15 and columns past the image: never stored)
    const int wo = w0 + lr;
    const bool inw = lr >= 1 && lr <= 14 && wo < q.Wout;
    const bool col1 = wo >= 0 && wo < p.Wout;                        // this lane's first-layer column lies inside the image
    const __amdgpu_buffer_rsrc_t ro = make_rsrc((T*)q.out + (size_t)n * q.Hout * q.Wout * q.Cout, q.Hout * q.Wout * q.Cout * (int)sizeof(T));
    const int orow = q.Wout * q.Cout * (int)sizeof(T);
    const int ooff = (inw && lq * 4 < q.Cout) ? (wo * q.Cout + lq * 4) * (int)sizeof(T) : kOob;
    float4 sc1 = make_float4(1.f, 1.f, 1.f, 1.f), sh1 = make_float4(0.f, 0.f, 0.f, 0.f), sc2 = sc1, sh2 = sh1;
    {
        float a[4] = {1.f, 1.f, 1.f, 1.f}, b[4] = {0.f, 0.f, 0.f, 0.f}, c2[4] = {1.f, 1.f, 1.f, 1.f}, d2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = lq * 4 + r;
            if (c < p.Cout) { if (p.scale) a[r] = p.scale[c]; if (p.shift) b[r] = p.shift[c]; }
            if (c < q.Cout) { if (q.scale) c2[r] = q.scale[c]; if (q.shift) d2[r] = q.shift[c]; }
        }
        sc1 = make_float4(a[0], a[1], a[2], a[3]); sh1 = make_float4(b[0], b[1], b[2], b[3]);
        sc2 = make_float4(c2[0], c2[1], c2[2], c2[3]); sh2 = make_float4(d2[0], d2[1], d2[2], d2[3]);
    }
    const bool affine1 = p.scale || p.shift, affine2 = q.scale || q.shift;
    auto act = [](float (&v)[4], int relu) {
        if (relu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        else if (relu == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.f + __expf(-v[r]));
        }
    };
    char* yb = ybuf + wave * kRowB;
    char* ywr = yb + (lr + 1) * 32 + lq * 8;                         // where this lane's 4 couts of pixel lr go
    const char* yrd = yb + lr * 32 + lq * 16;                        // B fragment of column shift kw: + kw * 32 (lq < 2; else zeros)

    // (both layers' weight fragments in registers for the whole strip - 194 VGPRs, two workgroups per CU - were measured: 0.3345 vs 0.3251 s
    // per 8-direction 512^3 prediction with the fragments re-read from LDS every row at four workgroups per CU: not kept)
    // x row hi completes first-layer row r1 = hi - 1, which completes second-layer row r2 = r1 - 1: three accumulators roll per layer
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, b0 = a0, b1 = a0, b2 = a0;
    uint4 ring[kD][3];
    const int hfirst = h0 - 2, hlast = h1 + 1;                       // x rows this strip needs (rows outside the image load as zeros)
#pragma unroll
    for (int u = 0; u < kD; ++u) load_row(hfirst + u, ring[u]);
    for (int base = hfirst; base <= hlast; base += kD) {
#pragma unroll
        for (int u = 0; u < kD; ++u) {
            const int hi = base + u;
            if (hi > hlast) break;
            int wb = wbase_l;
            asm volatile("" : "+v"(wb));   // keep the weight fragments in LDS (see conv_direct_kernel)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const uint4 w0f = *reinterpret_cast<const uint4*>(wl1 + ((0 * 3 + kw) * BN) * kPS + wb);
                const uint4 w1f = *reinterpret_cast<const uint4*>(wl1 + ((1 * 3 + kw) * BN) * kPS + wb);
                const uint4 w2f = *reinterpret_cast<const uint4*>(wl1 + ((2 * 3 + kw) * BN) * kPS + wb);
                mma16<T>(a2, w0f, ring[u][kw]);
                mma16<T>(a1, w1f, ring[u][kw]);
                mma16<T>(a0, w2f, ring[u][kw]);
            }
            load_row(hi + kD, ring[u]);
            // ---- first-layer row r1 is complete: epilogue, then into the wave's LDS row (zeros outside the image) ----
            const int r1 = hi - 1;
            {
                float v[4] = {a0[0], a0[1], a0[2], a0[3]};
                if (affine1) { v[0] = v[0] * sc1.x + sh1.x; v[1] = v[1] * sc1.y + sh1.y; v[2] = v[2] * sc1.z + sh1.z; v[3] = v[3] * sc1.w + sh1.w; }
                act(v, p.relu);
                const bool in1 = col1 && r1 >= 0 && r1 < p.Hout && lq * 4 < Cmid;
                uint2 pk;
                pk.x = in1 ? pack2<T>(v[0], v[1]) : 0u;
                pk.y = in1 ? pack2<T>(v[2], v[3]) : 0u;
                *reinterpret_cast<uint2*>(ywr) = pk;
            }
            a0 = a1; a1 = a2; a2 = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the row is in LDS (one wave: its LDS operations execute in order)
            __builtin_amdgcn_wave_barrier();
            // ---- second layer: first-layer row r1 feeds rows r1 + 1 (kh 0), r1 (kh 1), r1 - 1 (kh 2) ----
            int wb2 = wbase_l;
            asm volatile("" : "+v"(wb2));
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                uint4 yf = make_uint4(0u, 0u, 0u, 0u);
                if (lq < 2) yf = *reinterpret_cast<const uint4*>(yrd + kw * 32);
                const uint4 w0f = *reinterpret_cast<const uint4*>(wl2 + ((0 * 3 + kw) * BN) * kPS + wb2);
                const uint4 w1f = *reinterpret_cast<const uint4*>(wl2 + ((1 * 3 + kw) * BN) * kPS + wb2);
                const uint4 w2f = *reinterpret_cast<const uint4*>(wl2 + ((2 * 3 + kw) * BN) * kPS + wb2);
                mma16<T>(b2, w0f, yf);
                mma16<T>(b1, w1f, yf);
                mma16<T>(b0, w2f, yf);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the fragments are in registers before the next row overwrites the LDS row
            __builtin_amdgcn_wave_barrier();
            const int r2 = r1 - 1;                                   // this second-layer row is complete now
            {
                float v[4] = {b0[0], b0[1], b0[2], b0[3]};
                if (affine2) { v[0] = v[0] * sc2.x + sh2.x; v[1] = v[1] * sc2.y + sh2.y; v[2] = v[2] * sc2.z + sh2.z; v[3] = v[3] * sc2.w + sh2.w; }
                act(v, q.relu);
                const int off = (r2 >= h0 && r2 < h1) ? ooff : kOob;
                uint2 pk;
                pk.x = pack2<T>(v[0], v[1]);
                pk.y = pack2<T>(v[2], v[3]);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned int, pk), ro, off, r2 * orow, 0);
            }
            b0 = b1; b1 = b2; b2 = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// ---- segmentation head: <= 4 classes ---------------------------------------------------------------------------------
// Same strip walk as conv_direct_kernel, but with <= 4 output channels a 16-row MFMA tile would be 3/4 empty and the
// per-pixel epilogue (softmax / arg-max in prediction) would run on lanes that hold nothing: the kernel is VALU-bound.
// So the FOUR output rows h, h+1, h+2, h+3 (h % 4 == 0) share one accumulator tile - row h + q lives in MFMA rows 4q..4q+3,
// i.e. in the lanes of quad q.  An input row hi feeds output rows hi+1 (kh = 0), hi (kh = 1) and hi-1 (kh = 2): the weight
// operand for input-row phase m = hi & 3 carries W[kh = 0] in quad (m+1)&3, W[kh = 1] in quad m, W[kh = 2] in quad (m+3)&3
// and zeros in the fourth quad, so ONE MFMA per kw does what took three (12 phase matrices, built once per workgroup in
// LDS).  A finished row is moved out of its quad with a select; every fourth row all 64 lanes hold one finished pixel each
// and run the epilogue once.  Arithmetic per output value - products, order of accumulation, softmax - is unchanged.
// MODE 2: fp32 NCHW logits + bias (training / plain forward); 3: prediction - no logits leave the kernel: softmax -> first
// arg-max -> fp16 max-prob, cropped and written at the direction's voxel address as label / probability or as a packed key
// through an (order-free) atomic max.
template <typename T, int MODE>
__global__ __launch_bounds__(256, 4) void conv_head_kernel(ConvParams p, DirectGeom g, VolScatter vsc) {
    constexpr int EPS = CT<T>::EPS, kD = 4, NMAT = 12;
    constexpr int kOob = (int)0x80000000;
    __shared__ __attribute__((aligned(16))) char wl[NMAT * 16 * kPS];
    asm volatile("" ::"s"(p.src0), "s"(p.w), "s"(p.out), "s"(p.C0), "s"(p.Hin), "s"(p.Win), "s"(p.Hout), "s"(p.Wout), "s"(p.Cout),
                 "s"(p.shift), "s"(g.strips_w), "s"(g.chunks_h), "s"(g.RH), "s"(g.sw_magic), "s"(g.ch_magic), "s"(g.nwaves));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int Cin = p.C0, K = p.Cout;
    {   // phase matrices: row = (m * 3 + kw) * 16 + 4 * quad + class
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, K * 9 * Cin * (int)sizeof(T));
#pragma unroll
        for (int i = 0; i < NMAT * 16 * 4 / 256; ++i) {
            const int item = tid + i * 256;
            const int row = item >> 2, seg = item & 3;
            const int mat = row >> 4, quad = (row >> 2) & 3, k = row & 3;
            const int m = mat / 3, kw = mat - m * 3;
            const int kh = quad == ((m + 1) & 3) ? 0 : quad == m ? 1 : quad == ((m + 3) & 3) ? 2 : -1;
            const bool ok = kh >= 0 && k < K && seg * EPS < Cin;
            *reinterpret_cast<uint4*>(wl + swz(row, row, seg)) = bload(rw, ok ? ((k * 9 + kh * 3 + kw) * Cin + seg * EPS) * (int)sizeof(T) : -1, 0);
        }
    }
    __syncthreads();
    const int gw = xcd_block(g.xcd) * 4 + wave;             // this wave's strip
    if (gw >= g.nwaves) return;
    const int q = g.strips_w == 1 ? gw : (int)__umulhi((unsigned)gw, g.sw_magic);
    const int ws = gw - q * g.strips_w;
    const int n = g.chunks_h == 1 ? q : (int)__umulhi((unsigned)q, g.ch_magic);
    const int hc = q - n * g.chunks_h;
    const int w0 = ws * 16, h0 = hc * g.RH, h1 = min(p.Hout, h0 + g.RH);   // RH % 4 == 0: h0 & 3 == 0
    const int h1r = (h1 + 3) & ~3;
    const int rowbytes = p.Win * p.C0 * (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc((const T*)p.src0 + (size_t)n * p.Hin * p.Win * p.C0, p.Hin * rowbytes);
    int coff[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int col = w0 + lr + kw - 1;
        coff[kw] = (col >= 0 && col < p.Win && lq * EPS < Cin) ? (col * p.C0 + lq * EPS) * (int)sizeof(T) : -1;
    }
    auto load_row = [&](int hi, uint4 (&x)[3]) {
        const bool ok = hi >= 0 && hi < p.Hin;
        const int soff = ok ? hi * rowbytes : 0;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) x[kw] = bload(rx, ok ? coff[kw] : -1, soff);
    };
    const int wbase_l = swz(lr, lr, lq);
    const int wo = w0 + lr;
    const bool inw = wo < p.Wout;
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = (p.shift && r < K) ? p.shift[r] : 0.f;
    const int plane = p.Hout * p.Wout * 4;
    const __amdgpu_buffer_rsrc_t ro = MODE == 2 ? make_rsrc((float*)p.out + (size_t)n * K * p.Hout * p.Wout, K * plane) : make_rsrc(nullptr, 0);

    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, done = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 ring[kD][3];
#pragma unroll
    for (int u = 0; u < kD; ++u) load_row(h0 - 1 + u, ring[u]);
    for (int base = h0 - 1; base <= h1r; base += kD) {
#pragma unroll
        for (int u = 0; u < kD; ++u) {
            const int hi = base + u;
            if (hi > h1r) break;
            const int m = (u + 3) & 3;                        // hi & 3 (compile time: base & 3 == 3)
            int wb = wbase_l;
            asm volatile("" : "+v"(wb));                      // phase matrices stay in LDS, not in 48 hoisted registers
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                mma16<T>(acc, *reinterpret_cast<const uint4*>(wl + (m * 3 + kw) * 16 * kPS + wb), ring[u][kw]);
            load_row(hi + kD, ring[u]);
            const int qd = (u + 2) & 3;                       // output row hi - 1 is complete; it lives in quad (hi - 1) & 3
            if (lq == qd) { done = acc; acc = f32x4{0.f, 0.f, 0.f, 0.f}; }
            if (qd == 3) {                                    // rows hi-4 .. hi-1 are out: lane (lq, lr) owns row hi-4+lq, column wo
                const int h = hi - 4 + lq;
                const bool live = h >= h0 && h < h1 && inw;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = done[r] + bias[r];
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), ro, (live && r < K) ? (h * p.Wout + wo) * 4 : kOob, r * plane, 0);
                } else {                                      // same arithmetic, in the same order, as logits_to_volume_kernel
                    float mx = v[0];
#pragma unroll
                    for (int r = 1; r < 4; ++r) if (r < K) mx = fmaxf(mx, v[r]);
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (r < K) sum += expf(v[r] - mx);
                    float best = -1.f;
                    int lab = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r < K) { const float pk = __fdiv_rn(expf(v[r] - mx), sum); if (pk > best) { best = pk; lab = r; } }
                    const int rr = h - vsc.m.crop_top, jj = wo - vsc.m.crop_left;
                    if (live && rr >= 0 && rr < vsc.m.h && jj >= 0 && jj < vsc.m.w) {
                        const int64_t addr = vsc.m.base + (int64_t)(vsc.s0 + n) * vsc.m.ss + (int64_t)rr * vsc.m.sh + (int64_t)jj * vsc.m.sw;
                        const __half hv = __float2half_rn(best);
                        const uint32_t hb = __builtin_bit_cast(uint16_t, hv);
                        if (vsc.mode == 0) {
                            if (vsc.labels) vsc.labels[addr] = (uint8_t)lab;
                            if (vsc.probs) vsc.probs[addr] = (uint16_t)hb;
                        } else if (!vsc.stage) {
                            atomicMax(vsc.keys + addr, (hb << 16) | ((uint32_t)(15 - vsc.direction) << 8) | (uint32_t)lab);
                        } else {   // un-cropped slice layout; launch_keys_stage_scatter does the volume addressing
                            vsc.stage[((int64_t)n * p.Hout + h) * p.Wout + wo] = (hb << 16) | ((uint32_t)(15 - vsc.direction) << 8) | (uint32_t)lab;
                        }
                    }
                }
            }
        }
    }
}

// geometry the direct kernel covers
static bool direct_ok(int dtype, const ConvParams& p) {   // p.out_f32: bit 0 = fp32 store, bit 1 = NCHW layout
    const int CK = dtype == VS_F32 ? 16 : 32;
    const bool nchw = (p.out_f32 >> 1) != 0, f32 = (p.out_f32 & 1) != 0;
    const bool head_ok = p.Cout <= 4 && !p.pool0 && !p.scale && !p.relu && !p.up0 && !p.stats_partial;   // conv_head_kernel
    const bool out_ok = p.scatter ? (head_ok && (p.scatter->mode == 0 || (p.scatter->mode == 1 && p.scatter->keys)))
                                  : nchw ? (f32 && head_ok) : (!f32 && !(p.Cout & 3));
    return vs_option("conv_direct") && out_ok && !p.nl_bins && !p.bz && !p.gc && p.dil <= 1 && p.up0 != 2 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.C1 == 0 && p.C0 <= CK &&
           p.Cout <= 16 && !p.residual && !p.out1 && (!p.pool0 || (!(p.Hout & 1) && !(p.Wout & 1) && p.Cout % 4 == 0)) &&
           (long)p.N * p.Hout * p.Wout >= (long)vs_option("conv_direct_min_px") &&
           (double)p.Hout * p.Wout * std::max(p.Cout, 4) * 4.0 < 2.0e9;
}
static DirectGeom direct_geom(const ConvParams& p) {
    DirectGeom g{};
    g.strips_w = cdiv(p.Wout, 16);
    // rows per wave: every wave pays a 2-row warm-up of its input ring; with >= 4 Mi output pixels per launch (prediction batches
    // of 512 x 512 slices) there are waves to spare and longer strips win (512^3 x 12 prediction: 32 rows 0.507 s, 128 rows 0.499 s)
    const bool big = (long)p.N * p.Hout * p.Wout >= (4L << 20) * 4;
    g.RH = std::max(2, vs_option(big ? "conv_direct_rows_big" : "conv_direct_rows") & ~1);
    g.chunks_h = cdiv(p.Hout, g.RH);
    g.sw_magic = 0xffffffffu / (unsigned)g.strips_w + 1u;
    g.ch_magic = 0xffffffffu / (unsigned)g.chunks_h + 1u;
    g.nwaves = p.N * g.strips_w * g.chunks_h;
    g.xcd = 0;      // (an XCD-local strip order was measured: 0.5207 vs 0.5212 s per 512^3 x 12 prediction - not kept)
    return g;
}

template <typename T, int BN>
int launch_direct(const ConvParams& p, int out_nchw, hipStream_t s) {
    DirectGeom g = direct_geom(p);
    g.out_nchw = out_nchw;
    VS_REQUIRE((double)p.Hin * p.Win * p.C0 * sizeof(T) < 2.0e9 && (long)g.nwaves * 16 < (1L << 32), "conv_direct: tensor too large");
    const bool head = p.scatter || (p.Cout & 3) != 0 || out_nchw;
    if (head) {   // four output rows per accumulator tile: strips start at multiples of 4
        g.RH = std::max(4, g.RH & ~3);
        g.chunks_h = cdiv(p.Hout, g.RH);
        g.ch_magic = 0xffffffffu / (unsigned)g.chunks_h + 1u;
        g.nwaves = p.N * g.strips_w * g.chunks_h;
        VS_REQUIRE(!p.up0 && !p.relu && !p.scale && !p.stats_partial, "conv_head: plain 3x3 head only");
    }
    const dim3 grid(cdiv(g.nwaves, 4));
    // (fewer workgroups per CU - enforced with unused dynamic LDS - were measured on the prediction: 2 / 3 per CU 0.559 / 0.542 s against
    // 0.524 s at the kernel's own limit of 4: not kept)
    const size_t pad = 0;
    VolScatter sc{};
    if (p.scatter) {
        sc = *p.scatter;
        ConvParams q = p;
        q.scatter = nullptr;
        hipLaunchKernelGGL((conv_head_kernel<T, 3>), grid, dim3(256), pad, s, q, g, sc);
    } else if (head) {
        VS_REQUIRE(out_nchw && p.Cout <= 4 && BN == 16 && !p.pool0 && !p.scale, "conv_direct: unsupported ragged output");
        hipLaunchKernelGGL((conv_head_kernel<T, 2>), grid, dim3(256), pad, s, p, g, sc);
    } else if (p.pool0) {
        hipLaunchKernelGGL((conv_direct_kernel<T, BN, 1>), grid, dim3(256), pad, s, p, g);
    } else {
        VS_REQUIRE(!p.out_f32, "conv_direct: fp32 NHWC output is not supported");
        hipLaunchKernelGGL((conv_direct_kernel<T, BN, 0>), grid, dim3(256), pad, s, p, g);
    }
    VS_LAUNCH_CHECK();
    return VS_OK;
}

// the pair kernel's geometry: strips of 14 columns
static DirectGeom pair_geom(const ConvParams& q) {
    DirectGeom g{};
    g.strips_w = cdiv(q.Wout, 14);
    const bool big = (long)q.N * q.Hout * q.Wout >= (4L << 20) * 4;
    g.RH = std::max(2, vs_option(big ? "conv_direct_rows_big" : "conv_direct_rows") & ~1);
    g.chunks_h = cdiv(q.Hout, g.RH);
    g.sw_magic = 0xffffffffu / (unsigned)g.strips_w + 1u;
    g.ch_magic = 0xffffffffu / (unsigned)g.chunks_h + 1u;
    g.nwaves = q.N * g.strips_w * g.chunks_h;
    return g;
}

template <typename T, int BN, int PT, int NTAPS, int STRIDE, int NW = 4, int DIL = 1, bool NLOAD = false>
int launch_one(const ConvParams& p, const TileGeom& g, hipStream_t s) {
    static bool attr_set = false;
    auto kern = conv_igemm_kernel<T, BN, PT, NTAPS, STRIDE, NW, DIL, NLOAD>;
    const size_t lds = (size_t)g.PH * g.PW * kPS + (size_t)NTAPS * BN * kPS + 64 + (NLOAD ? (size_t)3 * ((p.C0 + 31) & ~31) * sizeof(float) : 0);
    VS_REQUIRE((double)p.Hin * p.Win * std::max(p.C0, p.C1) * sizeof(T) < 2.0e9 && (double)p.Cout * NTAPS * (p.C0 + p.C1) * sizeof(T) < 2.0e9,
               "conv_igemm: image or weight tensor exceeds the 32-bit staging offsets");
    VS_REQUIRE(lds <= 160 * 1024, "conv_igemm: LDS request %zu too large", lds);
    VS_REQUIRE(g.PH * g.PW * 4 <= patch_items(PT, DIL > 1 ? 2 : STRIDE, NW) * NW * 64, "conv_igemm: patch %dx%d exceeds the staging budget", g.PH, g.PW);
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    VS_REQUIRE(g.tiles_h * g.tiles_w < 65536 && p.N < 65536 && g.PH * g.PW + NW * 64 < 65536, "conv_igemm: tile grid too large");
    TileGeom gg = g;
    gg.groups = p.N * g.tiles_h * g.tiles_w;
    gg.ctiles = cdiv(p.Cout, BN);
    gg.ct_magic = 0xffffffffu / (unsigned)gg.ctiles + 1u;
    gg.ti_magic = 0xffffffffu / (unsigned)(g.tiles_h * g.tiles_w) + 1u;
    const long nwg = (long)cdiv(gg.groups, 8) * 8 * gg.ctiles;
    VS_REQUIRE(nwg < (1L << 31) && (long)gg.groups * (g.tiles_h * g.tiles_w) < (1L << 32), "conv_igemm: tile grid too large");
    if (g.probe) gg.probe = vs_probe_buffer((size_t)nwg);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NW * 64), lds, s, p, gg);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

template <typename T, int BN, int PT>
int launch_tk(const ConvParams& p, const TileGeom& g, hipStream_t s) {
    const int nt = p.KH * p.KW;
    if constexpr (BN >= 32) {
        if (p.dil == 2) return launch_one<T, BN, PT, 9, 1, 4, 2>(p, g, s);
        if (p.dil == 4) return launch_one<T, BN, PT, 9, 1, 4, 4>(p, g, s);
    }
    if constexpr (PT == 1) {
        if (p.stride == 2) return nt == 9 ? launch_one<T, BN, 1, 9, 2>(p, g, s) : launch_one<T, BN, 1, 1, 2>(p, g, s);
    }
    if constexpr (PT == 2 && BN == 64) {   // 8 x 16 output tiles of the stride-2 3x3 layers of a big batch
        if (p.stride == 2) return launch_one<T, 64, 2, 9, 2>(p, g, s);
    }
    if constexpr (std::is_same<T, bf16_t>::value) {
        if (p.nl_bins) return launch_one<T, BN, PT, 9, 1, 4, 1, true>(p, g, s);     // (dispatch checked: stride-1 3x3, no dilation)
    }
    return nt == 9 ? launch_one<T, BN, PT, 9, 1>(p, g, s) : launch_one<T, BN, PT, 1, 1>(p, g, s);
}

struct Pick { int BN, PT, NW; };
// tile width: static per kernel for stride 1 (see conv_igemm_kernel), by output width for stride 2
static int tile_tw(const ConvParams& p, int PT) { return p.stride == 1 ? (PT == 1 ? 8 : 16) : (p.Wout >= 16 ? 16 : 8); }

// One place decides the kernel configuration (cout tile, pixel tiles per wave, waves per workgroup):
//  * 8 waves x 2 pixel tiles = 256-pixel tiles for stride-1 layers with >= 16x16 outputs: half the weight-slab traffic per
//    FLOP of the 128-pixel tile, twice the waves per CU for latency hiding, half the staging registers per thread;
//  * 4 waves x 2 (128 px; also the stride-2 3x3 layers of large launches: 8x16 outputs) or x 1 (64 px: 8x8 images, stride 2)
//    otherwise;
//  * 32-wide cout tiles when 64-wide ones would leave fewer than `conv_min_wgs` workgroups.
Pick pick_cfg(const ConvParams& p) {
    Pick c;
    c.NW = 4;
    c.PT = (p.stride == 1 && p.Hout * p.Wout >= 128 && p.Wout >= 16) ? 2 : 1;
    if (p.stride == 2 && p.KH == 3 && p.Cout >= 64 && !p.gc && p.Wout >= 16 && p.Hout >= 8 &&
        (long)p.N * cdiv(p.Hout, 8) * cdiv(p.Wout, 16) * cdiv(p.Cout, 64) >= vs_option("conv_min_wgs")) c.PT = 2;
    const bool can8 = vs_option("conv_nw8") && p.stride == 1 && p.dil <= 1 && p.Hout >= 16 && p.Wout >= 16;
    auto wgs = [&](int bn, int px) {
        const int tw = tile_tw(p, px == 64 ? 1 : 2), th = px / tw;
        return (long)p.N * cdiv(p.Hout, th) * cdiv(p.Wout, tw) * cdiv(p.Cout, bn);
    };
    int bn = p.Cout >= 64 ? 64 : (p.Cout >= 32 ? 32 : 16);
    if (p.gc) bn = 32;     // grouped: one 32-channel super-group per cout tile
    if (p.out1 && p.split_c % 64 != 0 && bn == 64) bn = 32;   // a split data gradient (decoder concat): no cout tile may straddle the boundary (32 + 40
                                                               // channels under U-Net++ / efficientnet-b3)
    if (p.out1 && p.split_c % 32 != 0 && bn == 32) bn = 16;
    if (can8 && wgs(bn, 256) >= vs_option("conv_nw8_min_wgs")) { c.NW = 8; c.PT = 2; }
    if (bn == 64 && wgs(64, c.NW * c.PT * 16) < vs_option("conv_min_wgs")) bn = 32;
    c.BN = bn;
    return c;
}

// conv_ring_kernel (conv_ring.h) takes the deep stride-1 3x3 layers of small launches: >= 4 chunks of input channels
// (the ring needs a main loop to pay for its longer prologue) and grids of at most a few workgroups per CU, where one
// workgroup per CU with its staging off the registers beats two that stage through them (tools/convlab: layer2 / layer3 /
// layer4 / dec0 / dec1 shapes of the batch-32 step 12 - 22 % faster, the short-K 64- and 32-channel layers slower).
// 0 = no; 1 = 256-pixel tiles x 64 couts; 2 = 256-pixel tiles x 32 couts; 3 = pairs of 8 x 8 images x 32 couts
// conv_stream_kernel (conv_stream.h), the persistent form, takes the evaluation-mode stride-1 3x3 layers of LARGE launches
// (prediction: batches of 512 x 512 slices) - at least `conv_stream_min_tiles` tile jobs per CU-resident workgroup, so that
// the per-tile latencies it removes are what the launch consists of.  0 = no, else the cout tile (64 / 32)
static int stream_mode(int dtype, const ConvParams& p, int out_nchw) {
    if ((dtype != VS_BF16 && dtype != VS_F16) || !vs_option("conv_stream") || !ring::stream_ok(p, out_nchw) || p.C0 + p.C1 > 512 || p.nl_bins) return 0;
    const int bn = (p.Cout % 64 == 0) ? 64 : 32;
    const long jobs = (long)p.N * cdiv(p.Hout, 16) * cdiv(p.Wout, 16) * (p.Cout / bn);
    return jobs >= 256L * vs_option("conv_stream_min_tiles") ? bn : 0;
}

static int ring_mode(int dtype, const ConvParams& p, int out_nchw) {
    if (dtype != VS_BF16 || !vs_option("conv_ring") || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.dil > 1 || p.gc || p.scatter ||
        out_nchw || (p.Cout & 3) || p.out_f32) return 0;
    if (p.nl_bins && ((p.C0 & 31) || p.up0 == 2 || p.C0 > 1024)) return 0;   // normalise-on-load: whole chunks of src0, room for the table
    const int Cin = p.C0 + p.C1;
    if (Cin < 128 || (Cin & 7) || (p.C1 && (p.C0 & 31))) return 0;
    if (p.out1 && (p.split_c & 31)) return 0;
    if ((double)p.Hin * p.Win * std::max(p.C0, p.C1) * 2.0 * 2 >= 4.0e9) return 0;
    if (p.Hout == 8 && p.Wout == 8 && p.up0 == 0) {
        if (p.N % 2 == 0 && !p.pool0 && p.Cout >= 32 && (long)(p.N / 2) * cdiv(p.Cout, 32) <= 1024) return 3;
        return 0;
    }
    if (p.Hout < 16 || p.Wout < 16) return 0;
    const long tiles = (long)p.N * cdiv(p.Hout, 16) * cdiv(p.Wout, 16);
    const long w64 = tiles * cdiv(p.Cout, 64), w32 = tiles * cdiv(p.Cout, 32);
    const long maxw = vs_option("conv_ring_max_wgs");
    if (p.Cout >= 64 && w64 >= 224 && w64 <= maxw && (!p.out1 || p.split_c % 64 == 0)) return 1;
    if (p.Cout >= 32 && w32 >= 128 && w32 <= maxw) return 2;
    return 0;
}

template <typename T>
int dispatch(const ConvParams& p, int out_nchw, hipStream_t s) {
    constexpr int CK = CT<T>::CK, EPS = CT<T>::EPS;
    const int Cin = p.C0 + p.C1;
    VS_REQUIRE(Cin % EPS == 0 && p.C0 % EPS == 0, "conv_igemm: channel counts must be multiples of %d", EPS);
    VS_REQUIRE(p.C1 == 0 || p.C0 % CK == 0, "conv_igemm: concat boundary must be a multiple of %d", CK);
    VS_REQUIRE(p.up0 >= 0 && p.up0 <= 2, "conv_igemm: up0 must be 0, 1 or 2");
    VS_REQUIRE(p.up0 != 2 || (p.C1 == 0 && !(p.Hin & 1) && !(p.Win & 1)), "conv_igemm: a zero-stuffed source has no concat partner and even dims");
    VS_REQUIRE(((p.KH == 3 && p.KW == 3) || (p.KH == 1 && p.KW == 1)) && (p.stride == 1 || p.stride == 2),
               "conv_igemm: unsupported kernel %dx%d stride %d", p.KH, p.KW, p.stride);
    const int dil = p.dil > 1 ? p.dil : 1;
    VS_REQUIRE(dil == 1 || ((dil == 2 || dil == 4) && p.KH == 3 && p.stride == 1 && p.Cout >= 32 && !p.pool0),
               "conv_igemm: dilation 2 / 4 is built for stride-1 3x3 layers of >= 32 channels");
    VS_REQUIRE(p.Hout == (p.Hin + 2 * p.pad - (p.KH - 1) * dil - 1) / p.stride + 1 && p.Wout == (p.Win + 2 * p.pad - (p.KW - 1) * dil - 1) / p.stride + 1,
               "conv_igemm: inconsistent output dims");
    VS_REQUIRE(p.src0 && p.w && (p.out || p.scatter), "conv_igemm: null pointer");
    VS_REQUIRE(p.gc == 0 || (p.gc == 32 && p.C1 == 0 && p.C0 == p.Cout && p.Cout % 32 == 0 && !p.out1),
               "conv_igemm: grouped convolutions run on 32-channel super-groups with as many inputs as outputs");
    VS_REQUIRE(!(out_nchw || (p.Cout & 3)) || (!p.out1), "conv_igemm: ragged / NCHW output cannot be split");
    const Pick cfg = pick_cfg(p);
    const int BN = cfg.BN, PT = cfg.PT, NW = cfg.NW;
    if (p.out1) VS_REQUIRE(p.split_c % BN == 0, "conv_igemm: split_c %d not a multiple of the cout tile %d", p.split_c, BN);
    if (p.bz) {
        VS_REQUIRE(!p.pool0 && !p.out1 && !p.scale && !p.shift && !p.relu && !p.out_f32 && !out_nchw && !(p.Cout & 3) && !p.stats_partial &&
                   p.bmean && p.binvstd && p.bstats_partial && (!p.brelu || p.by || (p.bgamma && p.bbeta)),
                   "conv_igemm: the BN-backward epilogue takes a plain NHWC dgrad output");
    }
    if (p.pool0) {
        VS_REQUIRE(PT >= 2 && p.Wout >= 16 && !(p.Hout & 1) && !(p.Wout & 1) && !p.residual && !p.scale && !p.shift && !out_nchw,
                   "conv_igemm: pooled dgrad epilogue not available for this geometry");
        VS_REQUIRE((p.out1 ? p.split_c : p.Cout) % 4 == 0, "conv_igemm: pooled channel count must be a multiple of 4");
    }
    ConvParams pd = p;
    pd.out_f32 = p.out_f32 | (out_nchw << 1);
    if (p.nl_bins) {
        VS_REQUIRE(conv_igemm_nl_ok(Elem<T>::kDtype, pd) && p.nl_mean && p.nl_invstd && p.nl_gamma && p.nl_beta && p.nl_nb >= 1 && p.nl_rows >= 1,
                   "conv_igemm: normalise-on-load is built for the bf16 stride-1 3x3 layers (ask conv_igemm_nl_ok first)");
    }
    if (direct_ok(CT<T>::CK == 32 ? VS_BF16 : VS_F32, pd))      // (a 32-cout form was measured in round 4: 636 vs 426 us for the tile kernel on the 32 -> 32 layer of a 128 x 512^2 batch - not kept)
        return launch_direct<T, 16>(p, out_nchw, s);
    if constexpr (sizeof(T) == 2) {
        if (const int sm = stream_mode(Elem<T>::kDtype, p, out_nchw)) {
            unsigned long long* probe = vs_probe_buffer(256);
            const int stg = 0;      // (a start delay staggered by workgroup was measured neutral)
            return sm == 64 ? ring::launch_stream<T, 64, 2, 8, 4, 2, 2>(p, probe, s, 256, stg) : ring::launch_stream<T, 32, 2, 8, 4, 2, 2>(p, probe, s, 256, stg);
        }
    }
    if constexpr (std::is_same<T, bf16_t>::value) {
        const int rm = ring_mode(VS_BF16, p, out_nchw);
        if (rm) {
            const long groups = rm == 3 ? p.N / 2 : (long)p.N * cdiv(p.Hout, 16) * cdiv(p.Wout, 16);
            unsigned long long* probe = vs_probe_buffer((size_t)(cdiv((int)groups, 8) * 8) * cdiv(p.Cout, rm == 1 ? 64 : 32));
            if (p.nl_bins) {
                if (rm == 1) return ring::launch_ring<64, 2, 8, 4, 1, 2, 2, true>(p, out_nchw, probe, s);
                if (rm == 2) return ring::launch_ring<32, 2, 8, 4, 1, 4, 2, true>(p, out_nchw, probe, s);
                return ring::launch_ring<32, 2, 4, 3, 2, 1, 2, true>(p, out_nchw, probe, s);
            }
            if (rm == 1) return ring::launch_ring<64, 2, 8, 4, 1, 2, 2>(p, out_nchw, probe, s);
            if (rm == 2) return ring::launch_ring<32, 2, 8, 4, 1, 4, 2>(p, out_nchw, probe, s);
            return ring::launch_ring<32, 2, 4, 3, 2, 1, 2>(p, out_nchw, probe, s);
        }
    }
    VS_REQUIRE(!p.scatter, "conv_igemm: the volume-scatter epilogue needs the direct kernel (check conv_head_scatter_ok first)");
    TileGeom g;
    g.tw_shift = tile_tw(p, PT) == 16 ? 4 : 3;
    const int TW = 1 << g.tw_shift;
    g.TH = NW * 16 * PT / TW;
    g.tiles_h = cdiv(p.Hout, g.TH);
    g.tiles_w = cdiv(p.Wout, TW);
    g.PH = (g.TH - 1) * p.stride + (p.KH - 1) * dil + 1;
    g.PW = (TW - 1) * p.stride + (p.KW - 1) * dil + 1;
    g.out_nchw = out_nchw;
    g.pw_magic = 0xffffffffu / (unsigned)g.PW + 1u;       // exact for x * PW < 2^32
    g.tw_magic = 0xffffffffu / (unsigned)g.tiles_w + 1u;
    g.probe = vs_probe_buffer((size_t)p.N * g.tiles_h * g.tiles_w * cdiv(p.Cout, BN));
    if constexpr (std::is_same<T, bf16_t>::value) {
        if (NW == 8 && p.nl_bins) {
#define VS_CONV8N(bn) if (BN == bn) return launch_one<T, bn, 2, 9, 1, 8, 1, true>(p, g, s)
            VS_CONV8N(64); VS_CONV8N(32); VS_CONV8N(16);
#undef VS_CONV8N
        }
    }
    if (NW == 8) {
        const bool t9 = p.KH * p.KW == 9;
#define VS_CONV8(bn) if (BN == bn) return t9 ? launch_one<T, bn, 2, 9, 1, 8>(p, g, s) : launch_one<T, bn, 2, 1, 1, 8>(p, g, s)
        VS_CONV8(64); VS_CONV8(32); VS_CONV8(16);
#undef VS_CONV8
    }
#define VS_CONV_CASE(bn, pt) if (BN == bn && PT == pt) return launch_tk<T, bn, pt>(p, g, s)
    VS_CONV_CASE(64, 2); VS_CONV_CASE(64, 1);
    VS_CONV_CASE(32, 2); VS_CONV_CASE(32, 1);
    VS_CONV_CASE(16, 2); VS_CONV_CASE(16, 1);
#undef VS_CONV_CASE
    vs_set_error("conv_igemm: no kernel for BN=%d PT=%d NW=%d", BN, PT, NW);
    return VS_ERR_UNSUPPORTED;
}

}  // namespace

bool head_dgrad_planes_ok(int dtype, int classes, int H, int W, int C) {
    return (dtype == VS_BF16 || dtype == VS_F16) && classes >= 1 && 9 * classes <= 64 && C >= 4 && C <= 16 && !(C & 3) &&
           (double)classes * H * W < 2.0e9 && (double)H * W * C * 2.0 < 2.0e9;
}
int launch_head_dgrad_planes(int dtype, const float* dl, const void* w, void* out, int N, int classes, int H, int W, int C, hipStream_t s) {
    VS_REQUIRE(head_dgrad_planes_ok(dtype, classes, H, W, C), "head_dgrad_planes: unsupported shape (classes %d, %d x %d x %d)", classes, H, W, C);
    DirectGeom g{};
    g.strips_w = cdiv(W, 16);
    g.RH = std::max(2, vs_option("conv_direct_rows") & ~1);
    g.chunks_h = cdiv(H, g.RH);
    g.sw_magic = 0xffffffffu / (unsigned)g.strips_w + 1u;
    g.ch_magic = 0xffffffffu / (unsigned)g.chunks_h + 1u;
    g.nwaves = N * g.strips_w * g.chunks_h;
    const dim3 grid(cdiv(g.nwaves, 4));
    const bool two = 9 * classes > 32;
    if (dtype == VS_BF16) {
        if (two) hipLaunchKernelGGL((head_dgrad_planes_kernel<bf16_t, 2>), grid, dim3(256), 0, s, dl, (const bf16_t*)w, (bf16_t*)out, N, classes, H, W, C, g);
        else hipLaunchKernelGGL((head_dgrad_planes_kernel<bf16_t, 1>), grid, dim3(256), 0, s, dl, (const bf16_t*)w, (bf16_t*)out, N, classes, H, W, C, g);
    } else {
        if (two) hipLaunchKernelGGL((head_dgrad_planes_kernel<f16_t, 2>), grid, dim3(256), 0, s, dl, (const f16_t*)w, (f16_t*)out, N, classes, H, W, C, g);
        else hipLaunchKernelGGL((head_dgrad_planes_kernel<f16_t, 1>), grid, dim3(256), 0, s, dl, (const f16_t*)w, (f16_t*)out, N, classes, H, W, C, g);
    }
    VS_LAUNCH_CHECK();
    return VS_OK;
}


bool conv_igemm_can_pool(const ConvParams& p) {
    return pick_cfg(p).PT >= 2 && p.Wout >= 16 && !(p.Hout & 1) && !(p.Wout & 1);   // (the ring kernel's 16 x 16 tiles: the same condition)
}

// instantiation code of the kernel launch_conv_igemm picks: BN*1000 + PT*100 + NTAPS*10 + code
// (code 1 = stride 1, 2 = stride 2, 4 = direct (LDS-free) shallow-layer kernel, 8 = 8-wave 256-pixel tiles)
int conv_igemm_variant(int dtype, const ConvParams& p) {
    if (direct_ok(dtype, p)) return 16 * 1000 + 2 * 100 + 9 * 10 + 4;
    if (const int sm = stream_mode(dtype, p, p.out_f32 >> 1)) return sm * 1000 + 2 * 100 + 9 * 10 + 7;                    // 7 = persistent LDS-DMA ring
    if (const int rm = ring_mode(dtype, p, p.out_f32 >> 1)) return (rm == 1 ? 64 : 32) * 1000 + 2 * 100 + 9 * 10 + 6;   // 6 = LDS-DMA ring
    const Pick c = pick_cfg(p);
    return c.BN * 1000 + c.PT * 100 + (p.KH * p.KW) * 10 + (c.NW == 8 ? 8 : (p.stride == 2 ? 2 : 1));
}

bool conv_head_scatter_ok(int dtype, const ConvParams& p) { return p.scatter && direct_ok(dtype, p); }

// whether the kernel launch_conv_igemm picks for p can put its statistics into fixed-point bins (ConvParams::stats_bins): the
// kernels that end in conv_epilogue (tile and ring kernels) and the direct shallow-layer kernel (one atomic pair per wave and cout)
bool conv_igemm_bins_ok(int dtype, const ConvParams& p) { return dtype == VS_BF16; }

// whether launch_conv_igemm can normalise p.src0 while loading it (ConvParams::nl_*): the register-staged tile kernel's bf16
// stride-1 3x3 instantiations (p as the layer will be launched, nl_* set or not)
bool conv_igemm_nl_ok(int dtype, const ConvParams& p) {
    ConvParams q = p;
    q.nl_bins = nullptr;
    if (dtype != VS_BF16 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.dil > 1 || p.gc || p.scatter || p.up0 == 2 || p.bz ||
        (p.C0 & 7) || p.C0 > 2048) return false;
    if (direct_ok(dtype, q)) return false;    // the strip kernels have no such loader (yet)
    return true;
}

int conv_igemm_stat_rows(int dtype, const ConvParams& p) {
    if (direct_ok(dtype, p)) return direct_geom(p).nwaves;   // one partial row per wave
    if (const int rm = ring_mode(dtype, p, p.out_f32 >> 1)) return rm == 3 ? p.N / 2 : p.N * cdiv(p.Hout, 16) * cdiv(p.Wout, 16);
    const Pick c = pick_cfg(p);
    const int TW = tile_tw(p, c.PT), TH = c.NW * 16 * c.PT / TW;
    return p.N * cdiv(p.Hout, TH) * cdiv(p.Wout, TW);
}

int launch_conv_igemm(int dtype, const ConvParams& p, hipStream_t s) {
    const int nchw = p.out_f32 >> 1;  // out_f32: bit0 = fp32 store, bit1 = NCHW layout
    ConvParams q = p;
    q.out_f32 = p.out_f32 & 1;
    if (dtype == VS_BF16) return dispatch<bf16_t>(q, nchw, s);
    if (dtype == VS_F32) return dispatch<float>(q, nchw, s);
    if (dtype == VS_F16) {
        VS_REQUIRE(!q.bz && !q.stats_partial && !q.pool0, "conv_igemm: fp16 is the inference precision (no training epilogues)");
        return dispatch<f16_t>(q, nchw, s);
    }
    vs_set_error("conv_igemm: bad dtype %d", dtype);
    return VS_ERR_INVALID;
}

// Whether two chained evaluation-mode layers - p's output (NHWC, storage type) read by q and by nothing else - can run as ONE launch of
// conv_direct_pair_kernel: both are strip-kernel layers on their own (direct_ok), 16-bit storage, at most 16 channels between them.
bool conv_pair_ok(int dtype, const ConvParams& p, const ConvParams& q) {
    if ((dtype != VS_BF16 && dtype != VS_F16) || !vs_option("conv_pair")) return false;
    if (!direct_ok(dtype, p) || !direct_ok(dtype, q)) return false;
    if (p.pool0 || q.pool0 || p.scatter || q.scatter || p.out_f32 || q.out_f32 || p.stats_partial || p.stats_bins || q.stats_partial || q.stats_bins ||
        q.up0 || q.C1 || p.Cout > 16 || q.Cout > 16 || (p.Cout & 7) || (q.Cout & 3)) return false;
    return q.C0 == p.Cout && q.N == p.N && q.Hin == p.Hout && q.Win == p.Wout && q.Hout == p.Hout && q.Wout == p.Wout;
}

int launch_conv_pair(int dtype, const ConvParams& p, const ConvParams& q, hipStream_t s) {
    VS_REQUIRE(conv_pair_ok(dtype, p, q), "conv_pair: the two layers cannot share a launch (ask conv_pair_ok first)");
    VS_REQUIRE(p.src0 && p.w && q.w && q.out, "conv_pair: null pointer");
    VS_REQUIRE((double)p.Hin * p.Win * p.C0 * 2.0 < 2.0e9 && (double)q.Hout * q.Wout * q.Cout * 2.0 < 2.0e9, "conv_pair: tensor too large");
    const DirectGeom g = pair_geom(q);
    VS_REQUIRE((long)g.nwaves * 16 < (1L << 32), "conv_pair: grid too large");
    const dim3 grid(cdiv(g.nwaves, 4));
    if (dtype == VS_BF16) hipLaunchKernelGGL((conv_direct_pair_kernel<bf16_t>), grid, dim3(256), 0, s, p, q, g);
    else hipLaunchKernelGGL((conv_direct_pair_kernel<f16_t>), grid, dim3(256), 0, s, p, q, g);
    VS_LAUNCH_CHECK();
    return VS_OK;
}
