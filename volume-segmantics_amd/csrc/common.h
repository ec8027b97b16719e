// Shared device/host helpers for libvolseg_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/volseg_hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) short short8v;

typedef uint16_t bf16_t;  // storage type for bf16 tensors
typedef _Float16 f16_t;   // storage type for fp16 tensors (VS_F16: the inference-only precision)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// ---- error plumbing ------------------------------------------------------------------------------
void vs_set_error(const char* fmt, ...);
#define VS_CHECK_HIP(expr)                                                                    \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            vs_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return VS_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)
#define VS_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            vs_set_error(__VA_ARGS__);        \
            return VS_ERR_INVALID;            \
        }                                     \
    } while (0)
#define VS_LAUNCH_CHECK() VS_CHECK_HIP(hipGetLastError())

// ---- bf16 <-> f32 --------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN preserving) on gfx950
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kDtype = VS_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int kDtype = VS_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

template <> struct Elem<f16_t> {
    static constexpr int kDtype = VS_F16;
    __device__ static __forceinline__ float ld(const f16_t* p) { return (float)*p; }
    __device__ static __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }
};
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {       // v_cvt_pkrtz would truncate: two RNE conversions
    f16x2 h = {(f16_t)a, (f16_t)b};
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float2 unpack_f16(uint32_t u) {
    const f16x2 h = __builtin_bit_cast(f16x2, u);
    return make_float2((float)h.x, (float)h.y);
}
// two values -> one 32-bit word of T's storage format (16-bit types only)
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float a, float b) { return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16); }
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float a, float b) { return pack_f16(a, b); }

// 4 consecutive elements <-> float4
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ float4 ld4(const f16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    const float2 a = unpack_f16(u.x), b = unpack_f16(u.y);
    return make_float4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ void st4(f16_t* p, float4 v) { *reinterpret_cast<uint2*>(p) = make_uint2(pack_f16(v.x, v.y), pack_f16(v.z, v.w)); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    uint2 u;
    u.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    u.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    *reinterpret_cast<uint2*>(p) = u;
}
// 4 consecutive elements kept in their storage format (2 registers for bf16) until they are used
template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef float4 type; };
template <> struct Raw4<bf16_t> { typedef uint2 type; };
struct f16raw4 { uint2 u; };
template <> struct Raw4<f16_t> { typedef f16raw4 type; };
__device__ __forceinline__ f16raw4 ld4raw(const f16_t* p) { return f16raw4{*reinterpret_cast<const uint2*>(p)}; }
__device__ __forceinline__ float4 unpack4(f16raw4 r) {
    const float2 a = unpack_f16(r.u.x), b = unpack_f16(r.u.y);
    return make_float4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ float4 ld4raw(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ uint2 ld4raw(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ float4 unpack4(float4 v) { return v; }
__device__ __forceinline__ float4 unpack4(uint2 u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
// 8 consecutive elements
__device__ __forceinline__ void ld8(const bf16_t* p, float* o) {
    uint4 u = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
    o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
    o[4] = __uint_as_float(u.z << 16); o[5] = __uint_as_float(u.z & 0xffff0000u);
    o[6] = __uint_as_float(u.w << 16); o[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ void ld8(const f16_t* p, float* o) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const float2 a = unpack_f16(u.x), b = unpack_f16(u.y), c = unpack_f16(u.z), d = unpack_f16(u.w);
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; o[4] = c.x; o[5] = c.y; o[6] = d.x; o[7] = d.y;
}
__device__ __forceinline__ void st8(f16_t* p, const float* v) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_f16(v[0], v[1]), pack_f16(v[2], v[3]), pack_f16(v[4], v[5]), pack_f16(v[6], v[7]));
}
__device__ __forceinline__ void ld8(const float* p, float* o) {
    float4 a = ld4(p), b = ld4(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void st8(bf16_t* p, const float* v) {
    uint4 u;
    u.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    u.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    u.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
    u.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = u;
}
__device__ __forceinline__ void st8(float* p, const float* v) {
    st4(p, make_float4(v[0], v[1], v[2], v[3]));
    st4(p + 4, make_float4(v[4], v[5], v[6], v[7]));
}

// normalise-on-load of one 16-byte item (8 bf16 channels): the arithmetic of bn_apply_kernel, element for element
__device__ __forceinline__ uint4 nl_apply8(const uint4 v, const bool live, const float (&m)[8], const float (&a)[8], const float (&b)[8]) {
    float x[8];
    x[0] = __uint_as_float(v.x << 16); x[1] = __uint_as_float(v.x & 0xffff0000u);
    x[2] = __uint_as_float(v.y << 16); x[3] = __uint_as_float(v.y & 0xffff0000u);
    x[4] = __uint_as_float(v.z << 16); x[5] = __uint_as_float(v.z & 0xffff0000u);
    x[6] = __uint_as_float(v.w << 16); x[7] = __uint_as_float(v.w & 0xffff0000u);
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = fmaxf((x[k] - m[k]) * a[k] + b[k], 0.f);
    uint4 r;
    r.x = pack2<bf16_t>(x[0], x[1]); r.y = pack2<bf16_t>(x[2], x[3]); r.z = pack2<bf16_t>(x[4], x[5]); r.w = pack2<bf16_t>(x[6], x[7]);
    return live ? r : make_uint4(0u, 0u, 0u, 0u);   // padding / ragged edges stay zero
}

// ---- raw buffer loads ---------------------------------------------------------------------------------
// 16-byte load through a buffer descriptor: `voff` is a per-lane byte offset, `soff` a uniform one; an offset outside
// [0, bytes) - use -1 - returns zeros, so padding and ragged edges need no branches.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ uint4 bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Workgroup index with XCD locality: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs (id & 7), each with
// its own L2.  Kernels whose neighbouring workgroups share input halos (strips of a convolution, pooling windows) take this
// index instead of blockIdx.x: XCD k then works on the k-th contiguous eighth of the grid, so a halo is fetched into one L2
// instead of two.  Bijective for every grid size (the first n % 8 XCDs take one workgroup more).  `on` = 0: identity.
__device__ __forceinline__ int xcd_block(int on) {
    const int b = blockIdx.x, n = gridDim.x;
    if (!on || n < 16) return b;
    const int xcd = b & 7, j = b >> 3, q = n >> 3, r = n & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline size_t dtype_size(int dt) { return dt == VS_F32 ? 4 : 2; }

// One launcher body per storage type.  VS_F16 is the inference precision: the operators of an evaluation-mode forward pass
// dispatch through VS_FOR_T; the training-only operators keep their two-way dispatch behind VS_NO_F16.
#define VS_FOR_T(dtype, ...)                                          \
    do {                                                                   \
        if ((dtype) == VS_BF16) { typedef bf16_t T; __VA_ARGS__; }         \
        else if ((dtype) == VS_F16) { typedef f16_t T; __VA_ARGS__; }      \
        else { typedef float T; __VA_ARGS__; }                             \
    } while (0)
#define VS_NO_F16(dtype, what) VS_REQUIRE((dtype) != VS_F16, "%s: fp16 is the inference precision - this operator is built for fp32 / bf16", what)

// ---- internal op launchers (one per .hip file) ------------------------------------------------------
struct ConvParams {
    const void* src0; const void* src1;  // virtual input = cat(upsample(src0, 2^up0), src1) along C
    int C0, C1, up0;
    int N, Hin, Win;                     // virtual input spatial dims
    int Hout, Wout, stride, pad, KH, KW;
    const void* w;                       // [Cout][KH*KW][C0+C1]; grouped (gc = 32): [Cout][KH*KW][32], see vs_weights_prepare_grouped
    int Cout;
    void* out;                           // [N][Hout][Wout][Cout or split_c]
    void* out1; int split_c;             // optional: couts >= split_c go to out1 ([..][Cout-split_c])
    const float* scale; const float* shift;  // per-cout affine applied to the accumulator (may be null)
    const void* residual;                // same shape as out, added after the affine (may be null)
    int relu;
    int out_f32;                         // store fp32 regardless of the compute dtype
    int pool0;                           // dgrad through nearest-x2 upsampling: channels < (out1 ? split_c : Cout) are summed
                                         // over 2x2 pixel blocks and written to `out` at half resolution
    float* stats_partial;                // optional [tiles][2][Cout]: per-tile sum / sum of squares of the raw accumulators
    // Optional BatchNorm-backward reduction fused into a dgrad epilogue (the output IS the gradient w.r.t. the activation of a
    // conv+BN(+ReLU) unit, complete after this launch): the epilogue applies that unit's ReLU mask, stores the masked
    // gradient g and writes per-tile partials of sum(g) and sum(g * xhat) - the first sweep of BN backward never runs.
    const void* bz;                      // the unit's pre-BN conv output z (same shape / dtype as `out`); null = feature off
    const void* by;                      // its activation y for the mask (units with a residual input); null: recompute from z
    const float* bmean; const float* binvstd; const float* bgamma; const float* bbeta;
    float* bstats_partial;               // [tiles][2][Cout]
    int brelu;
    int dil;                             // 0 / 1 = none; 2 = dilation 2 of a stride-1 3x3 kernel (pad 2)
    int gc;                              // 0 = dense; 32 = grouped convolution on 32-channel super-groups (C0 == Cout, C1 == 0)
    // Optional, instead of stats_partial's one row per tile: the per-tile sums go, as 64-bit FIXED-POINT integers, into
    // stats_nb (power of two) rows of bins by atomic add (row = tile & (stats_nb - 1); layout [row][2][Cout]; sum x scaled by
    // 2^24, sum x^2 by 2^16).  Integer adds commute: the totals are bit-reproducible whatever order the workgroups arrive in,
    // and the consumer finalises stats_nb rows inline instead of waiting for a finalize launch over hundreds (norm.hip).
    unsigned long long* stats_bins; int stats_nb;
    // Normalise-on-load of src0 (training forward of a conv -> BN -> ReLU -> conv pair; nl_bins != null = on): src0 holds the
    // producer's PRE-norm output z and no normalisation sweep ran.  Every workgroup of THIS launch sums the producer's nl_nb rows of
    // fixed-point statistics bins itself in its prologue (the arithmetic of bn_apply_inline_kernel: fp64 mean / variance; workgroup 0
    // also publishes mean / invstd for the backward pass and updates the running statistics), then applies
    // y = max((z - mean) * (invstd * gamma) + beta, 0), rounded to the storage type exactly as the sweep would have stored it, to
    // every item between its buffer load and its LDS store (zero padding stays zero).  The workgroups of the first cout tile also
    // store the normalised interior of their patch to nl_y - the activation tensor the weight gradient reads comes out as a
    // by-product of the forward convolution, and the sweep (launch, one tensor read) is gone.
    const unsigned long long* nl_bins; int nl_nb;
    long long nl_rows; float nl_eps, nl_mom;
    float* nl_mean; float* nl_invstd; float* nl_rm; float* nl_rv;   // outputs of workgroup 0 ([C0]; rm / rv may be null)
    const float* nl_gamma; const float* nl_beta;
    void* nl_y;                          // [N][Hin >> up0][Win >> up0][C0] in the storage type
    const struct VolScatter* scatter;    // HOST pointer, optional (segmentation head in prediction): instead of storing logits,
                                         // softmax -> arg-max -> (label, fp16 max-prob) goes straight to the volume (see predict.hip)
};
// where the head's pixels land in the output volume(s): the epilogue form of vs_logits_to_volume (modes 0 and 1)
struct VolScatter {
    vs_dirmap m;
    int s0, direction, mode;             // first slice of the batch; direction index (key mode); 0 = labels/probs, 1 = packed keys
    uint8_t* labels; uint16_t* probs; uint32_t* keys;
    uint32_t* stage;                     // key mode, optional: keys go to stage[n][hp][wp] instead (launch_keys_stage_scatter follows)
};
// stage[n][hp][wp] packed keys -> cropped, max-merged into the key volume with the slice index as the fastest-moving lane
// index: for directions whose slices are the volume's contiguous axis the head's own scatter would touch one line per voxel
int launch_keys_stage_scatter(const uint32_t* stage, int nb, const vs_dirmap& m, int s0, uint32_t* keys, hipStream_t s);
bool conv_head_scatter_ok(int dtype, const ConvParams& p);
bool conv_igemm_bins_ok(int dtype, const ConvParams& p);     // whether p's kernel honours ConvParams::stats_bins   // whether launch_conv_igemm can honour p.scatter for this layer
bool conv_igemm_nl_ok(int dtype, const ConvParams& p);       // whether launch_conv_igemm can honour p.nl_* (normalise src0 on load)
int launch_conv_igemm(int dtype, const ConvParams& p, hipStream_t s);
// two chained evaluation-mode shallow layers (q reads p's output and nothing else does) as one launch: the tensor between them never exists
// the segmentation head's data gradient from dLoss / dlogits as fp32 NCHW planes (w: the forward's [classes][9][C] copy; out: [N][H][W][C])
bool head_dgrad_planes_ok(int dtype, int classes, int H, int W, int C);
int launch_head_dgrad_planes(int dtype, const float* dl, const void* w, void* out, int N, int classes, int H, int W, int C, hipStream_t s);
bool conv_pair_ok(int dtype, const ConvParams& p, const ConvParams& q);
int launch_conv_pair(int dtype, const ConvParams& p, const ConvParams& q, hipStream_t s);
bool conv_igemm_can_pool(const ConvParams& p);      // whether pool0 is supported for this geometry
int conv_igemm_stat_rows(int dtype, const ConvParams& p);   // number of partial rows stats_partial receives
int conv_igemm_variant(int dtype, const ConvParams& p);      // BN*1000 + PT*100 + taps*10 + code of the chosen instantiation
// Cross-rank BatchNorm statistics (SyncBatchNorm under data parallelism; vs_unet_set_stats_hook): `hook` sums `count` device values in
// place over the ranks, stream-ordered on `stream` (kind 0: 64-bit integers - the fixed-point statistics bins, whose sum is exact and
// the same bits on every rank; kind 1: fp32).  world = ranks contributing equal shares: the statistics then describe rows * world rows.
typedef int (*vs_stats_hook_fn)(void* user, void* values, int64_t count, int kind, void* stream);
struct BnSync { vs_stats_hook_fn hook; void* user; int world; float* scratch; };   // scratch: 2 * c device floats of the caller's
int launch_bn_bwd_from_partials(int dtype, const void* g, const void* x, const float* mean, const float* invstd, const float* gamma,
                                void* dx, void* dres, float* dgamma, float* dbeta, int64_t rows, int c, const float* partial,
                                int nparts, hipStream_t s, const BnSync* sync = nullptr);
int launch_bn_apply_from_partials(int dtype, const void* x, const float* partial, int nparts, float eps, float momentum, float* mean,
                                  float* invstd, float* running_mean, float* running_var, const float* gamma, const float* beta,
                                  const void* residual, int relu, void* y, int64_t rows, int c, hipStream_t s);
int launch_bn_finalize_partials(const float* partial, int nparts, int c, int64_t rows, float eps, float momentum,
                                float* mean, float* invstd, float* running_mean, float* running_var, hipStream_t s);
// train-mode BatchNorm whose sums sit in fixed-point bins (ConvParams::stats_bins): finalise the nb rows inline, normalise
constexpr double kStatScale1 = 16777216.0, kStatScale2 = 65536.0;     // 2^24, 2^16
constexpr double kBwdStatScale = 68719476736.0;                        // 2^36: gradient sums (|sum| < 1.3e8, resolution 1.5e-11)
int launch_bn_apply_from_bins(int dtype, const void* x, const unsigned long long* bins, int nb, float eps, float momentum, float* mean,
                              float* invstd, float* running_mean, float* running_var, const float* gamma, const float* beta,
                              const void* residual, int relu, void* y, int64_t rows, int c, hipStream_t s, int64_t stat_rows = 0);
int launch_zero_u64(unsigned long long* p, size_t n, hipStream_t s);
bool stem_fwd_bins_ok(int dtype);
int launch_stem_fwd_bins(const float* x, const float* w, void* z, int n, int h, int w_, unsigned long long* bins, int nb, hipStream_t s);

// BatchNorm backward (dx, dres, dgamma, dbeta): three launches, or ONE with a grid barrier for tensors of a few MB (norm.hip).
// ctl: 16 zeroed bytes of barrier counters that re-arm themselves (null: the end of `workspace`, zeroed by a memset first)
int bn_bwd_dispatch(int dtype, const void* dy, const void* y, const void* x, const float* mean, const float* invstd, const float* gamma,
                    const float* beta, int relu, void* dx, void* dres, float* dgamma, float* dbeta, int64_t rows, int c, float* workspace,
                    size_t workspace_bytes, unsigned* ctl, hipStream_t s, const BnSync* sync = nullptr);

int launch_upsample2x_bwd(int dtype, const void* dy, void* dx, int n, int h, int w, int c, int accumulate, hipStream_t stream);

struct WgradParams {
    const void* src0; const void* src1; int C0, C1, up0;  // the forward conv's virtual input
    int N, Hin, Win, Hout, Wout, stride, pad, KH, KW;
    const void* dy; int Cout;             // [N][Hout][Wout][Cout]
    float* dw;                            // [Cout][KH*KW][Cin] fp32 (overwritten)
    float* partials; size_t partial_bytes;  // workspace
    int dil;                              // 0 / 1 = none; 2 = the forward convolution was dilated by 2 (stride 1, 3x3)
    int cg;                               // 0 = dense; else channels per group of a grouped convolution (4 / 8 / 16 / 32, C0 == Cout):
                                          // dw is [Cout][KH*KW][cg]
    int dy_planes;                        // 0: dy is [N][Hout][Wout][Cout]; else dy is fp32 NCHW planes of dy_planes (= cout_live) channels -
                                          // only where conv_wgrad_takes_planes says so (the segmentation head on the row-streaming kernel)
    int cout_live;                        // 0 = all; else only the first cout_live gradient channels are real (the head's classes in its
                                          // 16-channel gradient) and, IF conv_wgrad_honours_cout_live says so, dw is [cout_live][KH*KW][Cin]
};
bool conv_wgrad_honours_cout_live(int dtype, const WgradParams& p);
bool conv_wgrad_takes_planes(int dtype, const WgradParams& p);   // (ask with dy_planes = the class count)
size_t wgrad_workspace_bytes(int dtype, const WgradParams& p);
int launch_conv_wgrad(int dtype, const WgradParams& p, hipStream_t s);
// dw[i] = sum_k partials[k*n + i], fixed summation order
int launch_slab_reduce(const float* partials, float* dw, size_t n, int nparts, hipStream_t s);
// the lane-group count G of the slab_reduce4_kernel<G> launch_slab_reduce picks for (n, nparts) - 1, 4 or 16 - or 0 when it takes the
// scalar kernel (n not a multiple of 4, unaligned buffers): a launch that sums several layers' slabs at once uses the same body
int slab_reduce_groups(const float* partials, const float* dw, size_t n, int nparts);
// dw = sum over the split-K slabs, 16 bytes per lane: thread (j, g) of a block sums slabs g, g + G, g + 2 G, .. of four
// consecutive outputs (two accumulators, four loads in flight), the G groups of an output meet in LDS in a fixed order -
// bitwise reproducible.  G widens the grid for the small tensors (64 x 9 x 64 outputs summed over 128 slabs).  `bid` = the block's
// index among the cdiv(n4, 256 / G) blocks of this tensor (a per-layer launch passes blockIdx.x, the per-group launch its offset).
template <int G>
__device__ __forceinline__ void slab_reduce4_body(const float4* __restrict__ partials, float4* __restrict__ dw, size_t n4, int nparts, unsigned bid) {
    constexpr int J = 256 / G;
    __shared__ float4 red[G][J];
    const int j = threadIdx.x % J, g = threadIdx.x / J;
    const size_t i = (size_t)bid * J + j;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (i < n4) {
        for (int k = g; k < nparts; k += 4 * G) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (k + u * G < nparts) ? partials[(size_t)(k + u * G) * n4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
            s0.x += v[0].x; s0.y += v[0].y; s0.z += v[0].z; s0.w += v[0].w;
            s1.x += v[1].x; s1.y += v[1].y; s1.z += v[1].z; s1.w += v[1].w;
            s0.x += v[2].x; s0.y += v[2].y; s0.z += v[2].z; s0.w += v[2].w;
            s1.x += v[3].x; s1.y += v[3].y; s1.z += v[3].z; s1.w += v[3].w;
        }
    }
    const float4 t = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    if constexpr (G == 1) {
        if (i < n4) dw[i] = t;
    } else {
        red[g][j] = t;
        __syncthreads();
        if (g == 0 && i < n4) {
            float4 a = red[0][j];
#pragma unroll
            for (int q = 1; q < G; ++q) { a.x += red[q][j].x; a.y += red[q][j].y; a.z += red[q][j].z; a.w += red[q][j].w; }
            dw[i] = a;
        }
    }
}
// ---- optimiser pieces used by the fused backward (optim.hip) -------------------------------------------------------------
struct AdamwRanges { int n; long off[160]; long len[160]; };   // passed to the kernel by value
int launch_adamw_slice(const vs_adamw_args& a, const float* grads, int64_t off, int64_t n, hipStream_t s);
int launch_adamw_ranges(const vs_adamw_args& a, const float* grads, const AdamwRanges& r, hipStream_t s);
int launch_weight_prepare(int dtype, const float* w, void* wc, void* wt, int cout, int taps, int cin, int cout_pad, hipStream_t s);
// grouped convolution (cg channels per group, cin == cout): fp32 [cout][taps][cg] -> the block-expanded [cout][taps][32] copy
// and its flipped / transposed twin [cin][taps reversed][32] (32-channel super-groups; zeros outside a group's own block)
int launch_weight_prepare_grouped(int dtype, const float* w, void* wc, void* wt, int cout, int taps, int cg, hipStream_t s);
// ConvTranspose2d(4, stride 2, padding 1) = 3x3 convolution onto 4 * cout channels + pixel shuffle (vs_depth_to_space2):
// w fp32 [cin][cout][4][4] -> wc [4 * cout][9][cin] and wt [cin][9 reversed][4 * cout]; and the way back for the gradient
int launch_convt_weight_prepare(int dtype, const float* w, void* wc, void* wt, int cin, int cout, hipStream_t s);
int launch_convt_wgrad_gather(const float* dense, float* dw, int cin, int cout, hipStream_t s);
int launch_two_group_wgrad_extract(const float* dense, float* dw, int cout, int taps, int cin, hipStream_t s);
