// Implicit-GEMM 3x3 stride-1 convolution with an LDS ring filled by direct-to-LDS loads (gfx950).
//
// Same math, tiling (8x16 output pixels x BN couts, 4 waves) and epilogue as conv_igemm.hip, different pipeline:
// every 32-channel (bf16) / 16-channel (f32) stage - the 10x18 input halo patch plus the 9*BN weight rows, 64 bytes
// each - is fetched by `global_load_lds_dwordx4` straight into one slot of an NSTAGE-deep LDS ring.  No staging
// registers, no ds_write pass, and NSTAGE-1 stages (not one) are in flight while the MFMAs of the current stage run,
// which is what it takes to cover L2/HBM latency with ~0.5 us of matrix work per stage.
//   * LDS image: unpadded 64-byte rows, 16-byte slot = seg ^ ((row >> 1) & 3).  An LDS-DMA wave-instruction writes
//     64 lanes x 16 B linearly, so the swizzle is applied to the per-lane SOURCE address and again on the read; with
//     gfx950's ds_read_b128 lane groups this XOR is conflict-free for every row alignment (tools/lds_bank_sim.py).
//   * zero padding / ragged channels: the lane's source pointer is redirected to a 64-byte zero page.
//   * sync per stage: s_waitcnt vmcnt(<stages still allowed in flight>), one raw s_barrier, issue the stage that
//     reuses the slot everyone just left, compute.
#include "conv_common.h"
#include "prof.h"

namespace {

__device__ __attribute__((aligned(64))) unsigned char g_zero_page[64];

constexpr int kPH = 10, kPW = 18, kP = kPH * kPW;   // 8x16 tile, 3x3, stride 1
constexpr int kPatchRows = 192;                      // 12 wave-instructions of 16 rows
constexpr int kPatchBytes = kPatchRows * 64;

template <int BN> struct DmaCfg;
template <> struct DmaCfg<64> { static constexpr int WI = 9, NSTAGE = 3; };   // 36 weight instructions / 4 waves
template <> struct DmaCfg<32> { static constexpr int WI = 5, NSTAGE = 4; };   // 18 -> padded to 20

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int BN>
__global__ __launch_bounds__(256, 1) void conv_igemm_dma_kernel(ConvParams p, int tiles_h, int tiles_w, int out_nchw) {
    constexpr int CK = CT<T>::CK, EPS = CT<T>::EPS, NJ = BN / 16, PT = 2;
    constexpr int WI = DmaCfg<BN>::WI, NSTAGE = DmaCfg<BN>::NSTAGE, NI = 3 + WI;
    constexpr int kWBytes = WI * 4 * 1024, kStage = kPatchBytes + kWBytes;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int Cin = p.C0 + p.C1;
    const int nchunks = (Cin + CK - 1) / CK;

    int bid = blockIdx.x;
    const int tx = bid % tiles_w; bid /= tiles_w;
    const int ty = bid % tiles_h;
    const int n = bid / tiles_h;
    const int h0 = ty * 8, w0 = tx * 16;
    const int n0 = blockIdx.y * BN;
    const int H0 = p.Hin >> p.up0, W0 = p.Win >> p.up0;

    // ---- per-lane DMA sources ----
    // patch instruction i of this wave covers rows (wave + 4i)*16 .. +15; lane -> row = that + (lane>>2), physical slot lane&3
    int poff0[3], poff1[3], pseg[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int row = (wave + 4 * i) * 16 + (lane >> 2);
        const int seg = (lane & 3) ^ ((row >> 1) & 3);
        const int ph = row / kPW, pw = row - ph * kPW;
        const int hi = h0 - 1 + ph, wi = w0 - 1 + pw;
        const bool ok = row < kP && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win;
        pseg[i] = seg * EPS;
        poff0[i] = ok ? ((n * H0 + (hi >> p.up0)) * W0 + (wi >> p.up0)) * p.C0 + seg * EPS : -1;
        poff1[i] = ok ? ((n * p.Hin + hi) * p.Win + wi) * p.C1 + seg * EPS : -1;
    }
    // weight instruction i of this wave covers rows (wave + 4i)*16 .. ; row = tap*BN + nr
    const int wrow0 = wave * 16 + (lane >> 2);
    const int wseg = ((lane & 3) ^ ((wrow0 >> 1) & 3)) * EPS;   // (64*i) >> 1 is a multiple of 4: same swizzle for every i
    const int w_nr = wrow0 % BN;
    const int w_tap0 = wrow0 / BN, w_tapstep = 64 / BN;          // tap of instruction i = w_tap0 + i * w_tapstep
    const bool w_ok = n0 + w_nr < p.Cout;
    const int wsrc0 = ((n0 + w_nr) * 9 + w_tap0) * Cin + wseg;

    auto issue = [&](int c) {
        const int c0 = c * CK;
        char* st = smem + (c % NSTAGE) * kStage;
        const bool from0 = c0 < p.C0;
        const T* src = from0 ? (const T*)p.src0 : (const T*)p.src1;
        const int cs = from0 ? p.C0 : p.C1;
        const int cb = from0 ? c0 : c0 - p.C0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int off = from0 ? poff0[i] : poff1[i];
            const void* g = (off >= 0 && cb + pseg[i] < cs) ? (const void*)(src + off + cb) : (const void*)g_zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(st + (wave + 4 * i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const bool ok = w_ok && (w_tap0 + i * w_tapstep) < 9 && c0 + wseg < Cin;
            const void* g = ok ? (const void*)((const T*)p.w + wsrc0 + i * w_tapstep * Cin + c0) : (const void*)g_zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(st + kPatchBytes + (wave + 4 * i) * 1024), 16, 0, 0);
        }
    };

    // ---- per-lane LDS read addressing ----
    int rbase[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) rbase[i] = (wave * PT + i) * kPW + lr;    // patch row of (th, tw) for tap (0,0)
    const int wread = kPatchBytes + lr * 64 + ((lq ^ ((lr >> 1) & 3)) << 4);  // + (tap*BN + 16j) * 64

    f32x4 acc[PT][NJ];
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < NSTAGE - 1 && c < nchunks; ++c) issue(c);

    for (int c = 0; c < nchunks; ++c) {
        // stage c has landed once at most `ahead` younger stages of this wave are still in flight
        const int ahead = min(NSTAGE - 2, nchunks - 1 - c);
        if (ahead >= 2) wait_vmcnt<2 * NI>();
        else if (ahead == 1) wait_vmcnt<NI>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // every wave's share of stage c is in LDS; every wave left stage c-1
        if (c + NSTAGE - 1 < nchunks) issue(c + NSTAGE - 1);
        const char* st = smem + (c % NSTAGE) * kStage;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
            uint4 wf[NJ], xf[PT];
#pragma unroll
            for (int j = 0; j < NJ; ++j) wf[j] = *reinterpret_cast<const uint4*>(st + wread + (tap * BN + j * 16) * 64);
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int r = rbase[i] + kh * kPW + kw;
                xf[i] = *reinterpret_cast<const uint4*>(st + r * 64 + ((lq ^ ((r >> 1) & 3)) << 4));
            }
#pragma unroll
            for (int i = 0; i < PT; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) mma16<T>(acc[i][j], wf[j], xf[i]);
        }
    }
    __syncthreads();   // ring is dead: the epilogue may use it as scratch
    conv_epilogue<T, BN, PT>(p, 4, out_nchw, n, h0, w0, n0, (int)blockIdx.x, acc, smem);
}

template <typename T, int BN>
int launch_dma(const ConvParams& p, int out_nchw, hipStream_t s) {
    static bool attr_set = false;
    auto kern = conv_igemm_dma_kernel<T, BN>;
    constexpr size_t lds = (size_t)DmaCfg<BN>::NSTAGE * (kPatchBytes + DmaCfg<BN>::WI * 4 * 1024);
    static_assert(lds <= 160 * 1024, "ring does not fit LDS");
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int tiles_h = cdiv(p.Hout, 8), tiles_w = cdiv(p.Wout, 16);
    dim3 grid((unsigned)(p.N * tiles_h * tiles_w), (unsigned)cdiv(p.Cout, BN));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p, tiles_h, tiles_w, out_nchw);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

}  // namespace

// geometry the ring kernel covers: 3x3, stride 1, pad 1, >= 8x16 outputs, cout tile 64 or 32, >= 2 stages of channels
bool conv_igemm_dma_ok(int dtype, const ConvParams& p, int BN) {
    const int CK = dtype == VS_BF16 ? 32 : 16;
    return vs_option("conv_dma") && !p.bz && p.up0 != 2 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.Wout >= 16 &&
           p.Hout * p.Wout >= 128 && (BN == 64 || BN == 32) && (p.C0 + p.C1) >= 2 * CK;
}

int launch_conv_igemm_dma(int dtype, const ConvParams& p, int BN, int out_nchw, hipStream_t s) {
    if (dtype == VS_BF16) return BN == 64 ? launch_dma<bf16_t, 64>(p, out_nchw, s) : launch_dma<bf16_t, 32>(p, out_nchw, s);
    return BN == 64 ? launch_dma<float, 64>(p, out_nchw, s) : launch_dma<float, 32>(p, out_nchw, s);
}
