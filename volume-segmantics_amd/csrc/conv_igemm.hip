// Implicit-GEMM NHWC convolution on MFMA (gfx950), im2col-free.
//
// One workgroup (4 waves) computes a TH x TW patch of output pixels of one image times BN output channels.
// For every chunk of CK input channels (64 bytes per pixel: 32 bf16 / 16 f32) the input halo patch and the
// BN x taps weight slab are staged in LDS once and reused by all KH*KW taps: a tap is just a constant byte
// offset into the staged patch, so there is no im2col buffer anywhere.  MFMA orientation: A = weights
// (rows = cout), B = pixels (cols), so each lane ends up with 4 consecutive output channels of one pixel
// -> 8/16-byte NHWC stores.
//
// Pipeline: the 16-byte global loads of chunk c+1 are all issued (into registers) BEFORE the MFMAs of
// chunk c and written to LDS after them, so HBM/L2 latency hides under the matrix work (one LDS buffer,
// two barriers per chunk; >= 2 workgroups per CU cover the barrier bubbles).
//
// bf16: v_mfma_f32_16x16x32_bf16 (one per 32-channel chunk-tap); f32: 4 x v_mfma_f32_16x16x4_f32 on the
// same 16-byte fragments (exact fp32 FMA chain - the parity path).
//
// Serves every 3x3 / 1x1 convolution of smp.Unet(resnet34) forward (reference call sites
// vol_seg_2d_trainer.py:424, vol_seg_2d_predictor.py:44), with the decoder's nearest-x2 upsample + concat
// folded into the patch loader, and - fed with flipped/transposed weights - their dgrad.
#include <algorithm>
#include <cstdlib>

#include "conv_common.h"
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
    unsigned long long* probe;  // phase timestamps (tools/conv_probe.py); null in normal operation
};

// 16-byte patch items per thread (NW = waves per workgroup; the tile has NW*PT*16 pixels)
constexpr int patch_items(int pt, int stride, int nw = 4) {
    return nw == 8 ? 3 : (stride == 2 ? 5 : (pt == 4 ? 6 : (pt == 2 ? 3 : 2)));
}

template <typename T, int BN, int PT, int NTAPS, int STRIDE, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : 1) void conv_igemm_kernel(ConvParams p, TileGeom g) {
    constexpr int NT = NW * 64;
    constexpr int CK = CT<T>::CK, EPS = CT<T>::EPS;
    constexpr int NJ = BN / 16, KW = NTAPS == 9 ? 3 : 1;
    constexpr int PITEMS = patch_items(PT, STRIDE, NW);
    constexpr int WROWS = NT / 4;                 // weight rows one pass of the workgroup stages
    constexpr int TS = WROWS / BN;                // taps per pass
    static_assert(WROWS % BN == 0, "cout tile must divide the rows of a staging pass");
    constexpr int WITEMS = (NTAPS + TS - 1) / TS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    // stride-1 kernels have a compile-time tile (16 wide; 8 wide for the 64-pixel tiles of small images), so patch
    // row offsets are instruction immediates; stride-2 kernels take it from the launch
    constexpr bool kStatic = STRIDE == 1;
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
    unsigned long long tprobe[5];
    if (g.probe) tprobe[0] = wall_clock64();

    const int ty = g.tiles_w == 1 ? (int)blockIdx.x : (int)__umulhi(blockIdx.x, g.tw_magic);
    const int tx = (int)blockIdx.x - ty * g.tiles_w;
    const int n = blockIdx.z;
    const int h0 = ty * TH, w0 = tx * TW;
    const int n0 = blockIdx.y * BN;
    const int hbase = h0 * STRIDE - p.pad, wbase = w0 * STRIDE - p.pad;
    const int H0 = p.Hin >> p.up0, W0 = p.Win >> p.up0;
    // Staging loads are raw buffer loads: one descriptor per source (base = this image, so per-lane offsets are 32-bit
    // byte offsets), the chunk's channel offset rides in the scalar offset, and offset -1 (out of range) returns zeros -
    // zero padding, ragged tiles and channel tails cost no branches.
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)p.src0 + (size_t)n * H0 * W0 * p.C0), 0, H0 * W0 * p.C0 * (int)sizeof(T), 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)p.src1 + (size_t)n * p.Hin * p.Win * p.C1), 0, p.src1 ? p.Hin * p.Win * p.C1 * (int)sizeof(T) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.Cout * NTAPS * Cin * (int)sizeof(T), 0x00020000);
    const int dummy = (P + NTAPS * BN) * kPS;    // 64 spare bytes behind the staged tiles: target of the stores of idle items

    // ---- chunk-invariant staging addresses (byte offsets; -1 = zero fill) ----
    int poff0[PITEMS], poff1[PITEMS], pdst[PITEMS];
#pragma unroll
    for (int i = 0; i < PITEMS; ++i) {
        const int item = tid + i * NT;
        const int pp = item >> 2, seg = item & 3;
        const int ph = kStatic ? pp / PW : (int)__umulhi((unsigned)pp, g.pw_magic), pw = pp - ph * PW;
        const int hi = hbase + ph, wi = wbase + pw;
        const bool ok = pp < P && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win;
        poff0[i] = ok ? (((hi >> p.up0) * W0 + (wi >> p.up0)) * p.C0 + seg * EPS) * (int)sizeof(T) : -1;
        poff1[i] = ok ? ((hi * p.Win + wi) * p.C1 + seg * EPS) * (int)sizeof(T) : -1;
        pdst[i] = pp < P ? swz(pp, pw, seg) : dummy;
    }
    // weight staging: pass i covers rows i*WROWS + (tid >> 2); row = tap*BN + nr, so a pass advances TS taps
    const int wrow0 = tid >> 2, wseg = tid & 3;
    const int wnr = wrow0 % BN, wtap0 = wrow0 / BN;
    const bool wok = n0 + wnr < p.Cout;
    const int woff0 = (((n0 + wnr) * NTAPS + wtap0) * Cin + wseg * EPS) * (int)sizeof(T);
    const int wdst0 = swz(wrow0, wrow0, wseg);   // i*WROWS is a multiple of 16 rows: same swizzle in every pass

    // Prefetch depth: PF chunks are in flight (in registers) while one is being multiplied.  Kernels that run with one or
    // two workgroups per CU and have registers to spare use 2, so a staging load gets two chunk periods to land.
    constexpr int PF = ((NW == 8 && BN <= 32) || (NW == 4 && PT == 1 && STRIDE == 1)) ? 2 : 1;
    uint4 pregs[PF][PITEMS], wregs[PF][WITEMS];
    auto load_chunk = [&](int c0, uint4 (&preg)[PITEMS], uint4 (&wreg)[WITEMS]) {
        const bool from0 = c0 < p.C0;
        const int cs = from0 ? p.C0 : p.C1;
        const int cb = from0 ? c0 : c0 - p.C0;
        const int nseg = (cs - cb) / EPS;         // valid 16-byte segments of this chunk (< 4 only in a ragged channel tail)
        const int nsegw = (Cin - c0) / EPS;
#pragma unroll
        for (int i = 0; i < PITEMS; ++i) {
            int off = from0 ? poff0[i] : poff1[i];
            if (((tid + i * NT) & 3) >= nseg) off = -1;
            preg[i] = from0 ? bload(r0, off, cb * (int)sizeof(T)) : bload(r1, off, cb * (int)sizeof(T));
        }
#pragma unroll
        for (int i = 0; i < WITEMS; ++i) {
            int off = woff0 + i * TS * Cin * (int)sizeof(T);
            if (!wok || wtap0 + i * TS >= NTAPS || wseg >= nsegw) off = -1;
            wreg[i] = bload(rw, off, c0 * (int)sizeof(T));
        }
    };

    // per-lane LDS read bases: pixel tile i, tap column kw (tap row kh adds the constant kh * PW * kPS)
    int xb[PT][KW];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int pl = wave * (PT * 16) + i * 16 + lr;
        const int th = pl >> tw_shift, tw = pl & (TW - 1);
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) xb[i][kw] = swz((th * STRIDE) * PW + tw * STRIDE + kw, tw * STRIDE + kw, lq);
    }
    const int wbase_l = swz(lr, lr, lq);                  // (tap*BN + 16j) is a multiple of 16: it does not change the swizzle
    const int khs = PW * kPS;

    f32x4 acc[PT][NJ];
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_chunk(0, pregs[0], wregs[0]);
    if (PF == 2 && CK < Cin) load_chunk(CK, pregs[PF - 1], wregs[PF - 1]);
    if (g.probe) tprobe[1] = wall_clock64();
    for (int cbase = 0; cbase < Cin; cbase += PF * CK) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int c0 = cbase + s * CK;
            if (c0 >= Cin) break;
            __syncthreads();  // every wave is done reading the previous chunk
#pragma unroll
            for (int i = 0; i < PITEMS; ++i) *reinterpret_cast<uint4*>(patch + pdst[i]) = pregs[s][i];
#pragma unroll
            for (int i = 0; i < WITEMS; ++i) {
                const bool full = (i + 1) * TS <= NTAPS;   // every row of this pass is a real tap
                const int dst = (full || wtap0 + i * TS < NTAPS) ? P * kPS + wdst0 + i * WROWS * kPS : dummy;
                *reinterpret_cast<uint4*>(smem + dst) = wregs[s][i];
            }
            __syncthreads();
            if (g.probe && c0 == 0) tprobe[2] = wall_clock64();
            if (c0 + PF * CK < Cin) load_chunk(c0 + PF * CK, pregs[s], wregs[s]);  // in flight while the MFMAs of PF chunks run
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
    conv_epilogue<T, BN, PT, NW>(p, tw_shift, g.out_nchw, n, h0, w0, n0, (int)(blockIdx.z * gridDim.x + blockIdx.x), acc, smem);
    if (g.probe) {
        __builtin_amdgcn_s_waitcnt(0);  // stores issued and acknowledged
        tprobe[4] = wall_clock64();
        if (tid == 0) {
            unsigned long long* o = g.probe + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;
            for (int i = 0; i < 5; ++i) o[i] = tprobe[i];
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            o[5] = hw; o[6] = xcc; o[7] = 0;
        }
    }
}

template <typename T, int BN, int PT, int NTAPS, int STRIDE, int NW = 4>
int launch_one(const ConvParams& p, const TileGeom& g, hipStream_t s) {
    static bool attr_set = false;
    auto kern = conv_igemm_kernel<T, BN, PT, NTAPS, STRIDE, NW>;
    const size_t lds = (size_t)g.PH * g.PW * kPS + (size_t)NTAPS * BN * kPS + 64;
    VS_REQUIRE((double)p.Hin * p.Win * std::max(p.C0, p.C1) * sizeof(T) < 2.0e9 && (double)p.Cout * NTAPS * (p.C0 + p.C1) * sizeof(T) < 2.0e9,
               "conv_igemm: image or weight tensor exceeds the 32-bit staging offsets");
    VS_REQUIRE(lds <= 160 * 1024, "conv_igemm: LDS request %zu too large", lds);
    VS_REQUIRE(g.PH * g.PW * 4 <= patch_items(PT, STRIDE, NW) * NW * 64, "conv_igemm: patch %dx%d exceeds the staging budget", g.PH, g.PW);
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    VS_REQUIRE(g.tiles_h * g.tiles_w < 65536 && p.N < 65536 && g.PH * g.PW + NW * 64 < 65536, "conv_igemm: tile grid too large");
    dim3 grid((unsigned)(g.tiles_h * g.tiles_w), (unsigned)cdiv(p.Cout, BN), (unsigned)p.N);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, p, g);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

template <typename T, int BN, int PT>
int launch_tk(const ConvParams& p, const TileGeom& g, hipStream_t s) {
    const int nt = p.KH * p.KW;
    if constexpr (PT == 1) {
        if (p.stride == 2) return nt == 9 ? launch_one<T, BN, 1, 9, 2>(p, g, s) : launch_one<T, BN, 1, 1, 2>(p, g, s);
    }
    return nt == 9 ? launch_one<T, BN, PT, 9, 1>(p, g, s) : launch_one<T, BN, PT, 1, 1>(p, g, s);
}

struct Pick { int BN, PT, NW; };
// tile width: static per kernel for stride 1 (see conv_igemm_kernel), by output width for stride 2
static int tile_tw(const ConvParams& p, int PT) { return p.stride == 1 ? (PT == 1 ? 8 : 16) : (p.Wout >= 16 ? 16 : 8); }
static int g_dtype_hint = VS_BF16;

// One place decides the kernel configuration (cout tile, pixel tiles per wave, waves per workgroup):
//  * 8 waves x 2 pixel tiles = 256-pixel tiles for stride-1 layers with >= 16x16 outputs: half the weight-slab traffic per
//    FLOP of the 128-pixel tile, twice the waves per CU for latency hiding, half the staging registers per thread;
//  * 4 waves x 2 (128 px) or x 1 (64 px: 8x8 images, stride 2) otherwise;
//  * 32-wide cout tiles when 64-wide ones would leave fewer than `conv_min_wgs` workgroups.
Pick pick_cfg(const ConvParams& p) {
    Pick c;
    c.NW = 4;
    c.PT = (p.stride == 1 && p.Hout * p.Wout >= 128 && p.Wout >= 16) ? 2 : 1;
    const bool can8 = vs_option("conv_nw8") && p.stride == 1 && p.Hout >= 16 && p.Wout >= 16;
    auto wgs = [&](int bn, int px) {
        const int tw = tile_tw(p, px == 64 ? 1 : 2), th = px / tw;
        return (long)p.N * cdiv(p.Hout, th) * cdiv(p.Wout, tw) * cdiv(p.Cout, bn);
    };
    int bn = p.Cout >= 64 ? 64 : (p.Cout >= 32 ? 32 : 16);
    if (can8 && wgs(bn, 256) >= vs_option("conv_nw8_min_wgs")) { c.NW = 8; c.PT = 2; }
    if (bn == 64 && wgs(64, c.NW * c.PT * 16) < vs_option("conv_min_wgs")) bn = 32;
    c.BN = bn;
    if (conv_igemm_dma_ok(g_dtype_hint, p, bn)) { c.NW = 4; c.PT = 2; }
    return c;
}

template <typename T>
int dispatch(const ConvParams& p, int out_nchw, hipStream_t s) {
    constexpr int CK = CT<T>::CK, EPS = CT<T>::EPS;
    const int Cin = p.C0 + p.C1;
    VS_REQUIRE(Cin % EPS == 0 && p.C0 % EPS == 0, "conv_igemm: channel counts must be multiples of %d", EPS);
    VS_REQUIRE(p.C1 == 0 || p.C0 % CK == 0, "conv_igemm: concat boundary must be a multiple of %d", CK);
    VS_REQUIRE(p.up0 == 0 || p.up0 == 1, "conv_igemm: up0 must be 0 or 1");
    VS_REQUIRE(((p.KH == 3 && p.KW == 3) || (p.KH == 1 && p.KW == 1)) && (p.stride == 1 || p.stride == 2),
               "conv_igemm: unsupported kernel %dx%d stride %d", p.KH, p.KW, p.stride);
    VS_REQUIRE(p.Hout == (p.Hin + 2 * p.pad - p.KH) / p.stride + 1 && p.Wout == (p.Win + 2 * p.pad - p.KW) / p.stride + 1,
               "conv_igemm: inconsistent output dims");
    VS_REQUIRE(p.src0 && p.w && p.out, "conv_igemm: null pointer");
    VS_REQUIRE(!(out_nchw || (p.Cout & 3)) || (!p.out1), "conv_igemm: ragged / NCHW output cannot be split");
    const Pick cfg = pick_cfg(p);
    const int BN = cfg.BN, PT = cfg.PT, NW = cfg.NW;
    if (p.out1) VS_REQUIRE(p.split_c % BN == 0, "conv_igemm: split_c %d not a multiple of the cout tile %d", p.split_c, BN);
    if (p.pool0) {
        VS_REQUIRE(PT >= 2 && p.Wout >= 16 && !(p.Hout & 1) && !(p.Wout & 1) && !p.residual && !p.scale && !p.shift && !out_nchw,
                   "conv_igemm: pooled dgrad epilogue not available for this geometry");
        VS_REQUIRE((p.out1 ? p.split_c : p.Cout) % 4 == 0, "conv_igemm: pooled channel count must be a multiple of 4");
    }
    if (conv_igemm_dma_ok(CT<T>::CK == 32 ? VS_BF16 : VS_F32, p, BN))
        return launch_conv_igemm_dma(CT<T>::CK == 32 ? VS_BF16 : VS_F32, p, BN, out_nchw, s);
    TileGeom g;
    g.tw_shift = tile_tw(p, PT) == 16 ? 4 : 3;
    const int TW = 1 << g.tw_shift;
    g.TH = NW * 16 * PT / TW;
    g.tiles_h = cdiv(p.Hout, g.TH);
    g.tiles_w = cdiv(p.Wout, TW);
    g.PH = (g.TH - 1) * p.stride + p.KH;
    g.PW = (TW - 1) * p.stride + p.KW;
    g.out_nchw = out_nchw;
    g.pw_magic = 0xffffffffu / (unsigned)g.PW + 1u;       // exact for x * PW < 2^32
    g.tw_magic = 0xffffffffu / (unsigned)g.tiles_w + 1u;
    g.probe = vs_probe_buffer((size_t)p.N * g.tiles_h * g.tiles_w * cdiv(p.Cout, BN));
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

bool conv_igemm_can_pool(const ConvParams& p) {
    return pick_cfg(p).PT >= 2 && p.Wout >= 16 && !(p.Hout & 1) && !(p.Wout & 1);
}

// instantiation code of the kernel launch_conv_igemm picks: BN*1000 + PT*100 + NTAPS*10 + code
// (code 1 = stride 1, 2 = stride 2, 3 = LDS-DMA ring kernel, 8 = 8-wave 256-pixel tiles)
int conv_igemm_variant(int dtype, const ConvParams& p) {
    g_dtype_hint = dtype;
    const Pick c = pick_cfg(p);
    if (conv_igemm_dma_ok(dtype, p, c.BN)) return c.BN * 1000 + 2 * 100 + 9 * 10 + 3;
    return c.BN * 1000 + c.PT * 100 + (p.KH * p.KW) * 10 + (c.NW == 8 ? 8 : ((c.PT == 1 && p.stride == 2) ? 2 : 1));
}

int conv_igemm_stat_rows(const ConvParams& p) {
    const Pick c = pick_cfg(p);
    const int TW = tile_tw(p, c.PT), TH = c.NW * 16 * c.PT / TW;
    return p.N * cdiv(p.Hout, TH) * cdiv(p.Wout, TW);
}

int launch_conv_igemm(int dtype, const ConvParams& p, hipStream_t s) {
    g_dtype_hint = dtype;
    const int nchw = p.out_f32 >> 1;  // out_f32: bit0 = fp32 store, bit1 = NCHW layout
    ConvParams q = p;
    q.out_f32 = p.out_f32 & 1;
    if (dtype == VS_BF16) return dispatch<bf16_t>(q, nchw, s);
    if (dtype == VS_F32) return dispatch<float>(q, nchw, s);
    vs_set_error("conv_igemm: bad dtype %d", dtype);
    return VS_ERR_INVALID;
}
