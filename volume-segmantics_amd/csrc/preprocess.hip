// Volume pre-processing on the device: the statistics and the clip-to-uint8 map of BaseDataManager._preprocess_data
// (volume_segmantics/data/base_data_manager.py:29-42) and utilities.clip_to_uint8 (base_data_utils.py:243-287).
//
// The reference computes np.nanmean / np.nanstd and then rescales with NumPy element-wise arithmetic in the volume's own
// float type (float32 stays float32; integers go through float64).  The clip bounds come out of those statistics, and a
// bound that differs in its last bit moves voxels across a truncation boundary - so the sums here follow NumPy's add.reduce
// order exactly (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum, and the 8192-element buffered outer loop):
//
//     total = 0;  for every chunk of 8192 consecutive elements:  total += pairwise(chunk)
//     pairwise(a, n):  n < 8      -> sequential
//                      n <= 128   -> 8 interleaved accumulators r[j] += a[8i + j], ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), tail
//                      otherwise  -> pairwise(a, n2) + pairwise(a + n2, n - n2),  n2 = n/2 rounded down to a multiple of 8
//
// One workgroup per chunk: coalesced loads, element transform (NaN -> 0, or squared deviation from the mean) into LDS,
// 512 accumulator chains in parallel, then the fixed combine tree; a one-thread kernel adds the chunk sums in order.
// All in the accumulation type NumPy uses (float for float32 volumes, double for float64 and - after its cast - integers),
// with separately rounded operations (no contraction).  HBM-bound: 1 read of the volume per pass.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "common.h"

namespace {

constexpr int kChunk = 8192, kLeaf = 128, kPad = 8;   // LDS index of element e: e + (e >> 7) * kPad (conflict-free chains)
__device__ __forceinline__ int lidx(int e) { return e + (e >> 7) * kPad; }
constexpr int kVLen = kChunk + (kChunk / kLeaf) * kPad;
template <typename C> constexpr size_t chunk_lds() { return (size_t)(kVLen + kChunk / 16) * sizeof(C) + 16; }

template <typename C> __device__ __forceinline__ C addr(C a, C b);
template <> __device__ __forceinline__ float addr<float>(float a, float b) { return __fadd_rn(a, b); }
template <> __device__ __forceinline__ double addr<double>(double a, double b) { return __dadd_rn(a, b); }
template <typename C> __device__ __forceinline__ C subr(C a, C b);
template <> __device__ __forceinline__ float subr<float>(float a, float b) { return __fsub_rn(a, b); }
template <> __device__ __forceinline__ double subr<double>(double a, double b) { return __dsub_rn(a, b); }
template <typename C> __device__ __forceinline__ C mulr(C a, C b);
template <> __device__ __forceinline__ float mulr<float>(float a, float b) { return __fmul_rn(a, b); }
template <> __device__ __forceinline__ double mulr<double>(double a, double b) { return __dmul_rn(a, b); }
template <typename C> __device__ __forceinline__ C divr(C a, C b);
template <> __device__ __forceinline__ float divr<float>(float a, float b) { return __fdiv_rn(a, b); }
template <> __device__ __forceinline__ double divr<double>(double a, double b) { return __ddiv_rn(a, b); }

template <typename TIn> __device__ __forceinline__ bool is_nan(TIn v) { return false; }
template <> __device__ __forceinline__ bool is_nan<float>(float v) { return v != v; }
template <> __device__ __forceinline__ bool is_nan<double>(double v) { return v != v; }

// pairwise(a, n) of the LDS-resident, already transformed chunk, n <= 128
template <typename C>
__device__ C leaf_sum(const C* v, int off, int n) {
    if (n < 8) {
        C res = 0;
        for (int i = 0; i < n; ++i) res = addr<C>(res, v[lidx(off + i)]);
        return res;
    }
    C r[8];
    for (int j = 0; j < 8; ++j) r[j] = v[lidx(off + j)];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] = addr<C>(r[j], v[lidx(off + i + j)]);
    C res = addr<C>(addr<C>(addr<C>(r[0], r[1]), addr<C>(r[2], r[3])), addr<C>(addr<C>(r[4], r[5]), addr<C>(r[6], r[7])));
    for (; i < n; ++i) res = addr<C>(res, v[lidx(off + i)]);
    return res;
}

// pairwise(a, n) for any n <= 8192 on one thread (the ragged last chunk): explicit stack instead of recursion
template <typename C>
__device__ C pairwise_serial(const C* v, int n) {
    struct Frame { int off, n, stage; C left; };
    Frame st[16];
    int sp = 0;
    st[0] = Frame{0, n, 0, 0};
    C ret = 0;
    while (sp >= 0) {
        Frame& f = st[sp];
        if (f.n <= kLeaf) { ret = leaf_sum<C>(v, f.off, f.n); --sp; continue; }
        int n2 = f.n / 2;
        n2 -= n2 % 8;
        if (f.stage == 0) { f.stage = 1; st[sp + 1] = Frame{f.off, n2, 0, 0}; ++sp; }
        else if (f.stage == 1) { f.left = ret; f.stage = 2; st[sp + 1] = Frame{f.off + n2, f.n - n2, 0, 0}; ++sp; }
        else { ret = addr<C>(f.left, ret); --sp; }
    }
    return ret;
}

// OP 0: x with NaN -> 0 (and the NaN count);  OP 1: (x - avg)^2 with NaN -> 0
template <typename TIn, typename C, int OP>
__global__ __launch_bounds__(256) void chunk_sums_kernel(const TIn* __restrict__ data, int64_t n, C avg, C* __restrict__ sums,
                                                       int* __restrict__ nan_counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 72 KB with C = double: dynamic
    C* v = reinterpret_cast<C*>(smem);                            // the transformed chunk
    C* part = v + kVLen;                                          // 512 chain sums
    int* nan_part = reinterpret_cast<int*>(part + kChunk / 16);
    const int tid = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * kChunk;
    const int m = (int)((n - c0) < kChunk ? (n - c0) : kChunk);
    int nans = 0;
    for (int e = tid; e < m; e += 256) {
        const TIn x = data[c0 + e];
        const bool bad = is_nan<TIn>(x);
        nans += bad;
        C val;
        if (OP == 0) val = bad ? (C)0 : (C)x;
        else { const C d = subr<C>((C)x, avg); val = bad ? (C)0 : mulr<C>(d, d); }
        v[lidx(e)] = val;
    }
    if (OP == 0) {
        for (int o = 32; o; o >>= 1) nans += __shfl_xor(nans, o, 64);
        if ((tid & 63) == 0) nan_part[tid >> 6] = nans;
    }
    __syncthreads();
    if (m == kChunk) {
        // 64 leaves x 8 chains; chain c = leaf * 8 + j adds elements leaf*128 + 8i + j, i = 0..15, in order
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = tid + k * 256;
            const int base = (c >> 3) * kLeaf + (c & 7);
            C r = v[lidx(base)];
#pragma unroll
            for (int i = 1; i < kLeaf / 8; ++i) r = addr<C>(r, v[lidx(base + 8 * i)]);
            part[c] = r;
        }
        __syncthreads();
        if (tid < 64) {   // leaf sums, then the balanced tree over the 64 leaves (each level pairs neighbours: left + right)
            const C* r = part + tid * 8;
            C s = addr<C>(addr<C>(addr<C>(r[0], r[1]), addr<C>(r[2], r[3])), addr<C>(addr<C>(r[4], r[5]), addr<C>(r[6], r[7])));
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const C other = __shfl_xor(s, o, 64);
                s = (tid & o) ? addr<C>(other, s) : addr<C>(s, other);   // IEEE addition commutes: both lanes hold left + right
            }
            if (tid == 0) sums[blockIdx.x] = s;
        }
    } else if (tid == 0) {
        sums[blockIdx.x] = pairwise_serial<C>(v, m);
    }
    if (OP == 0 && tid == 0) nan_counts[blockIdx.x] = nan_part[0] + nan_part[1] + nan_part[2] + nan_part[3];
}

// total = ((0 + s0) + s1) + ...  (NumPy's outer loop over the 8192-element buffers), NaN count alongside.  The additions
// are a dependent chain on one lane; the workgroup only stages the chunk sums in LDS so that chain never waits for HBM.
template <typename C>
__global__ __launch_bounds__(256) void sequential_sum_kernel(const C* __restrict__ sums, const int* __restrict__ nan_counts,
                                                           int64_t nchunks, double* __restrict__ out) {
    constexpr int kTile = 4096;
    __shared__ C tile[kTile];
    __shared__ unsigned long long nan_total;
    const int tid = threadIdx.x;
    if (tid == 0) nan_total = 0;
    C acc = 0;
    unsigned long long nans = 0;
    for (int64_t t0 = 0; t0 < nchunks; t0 += kTile) {
        const int m = (int)((nchunks - t0) < kTile ? (nchunks - t0) : kTile);
        __syncthreads();
        for (int k = tid; k < m; k += 256) {
            tile[k] = sums[t0 + k];
            if (nan_counts) nans += (unsigned)nan_counts[t0 + k];
        }
        __syncthreads();
        if (tid == 0)
            for (int k = 0; k < m; ++k) acc = addr<C>(acc, tile[k]);
    }
    if (nans) atomicAdd(&nan_total, nans);
    __syncthreads();
    if (tid == 0) {
        out[0] = (double)acc;          // a float is exactly representable as a double
        out[1] = (double)nan_total;
    }
}

template <typename TIn, typename C>
int run_sum(const TIn* data, int64_t n, int op, double avg, void* ws, double* out, hipStream_t s) {
    const int64_t nchunks = (n + kChunk - 1) / kChunk;
    C* sums = (C*)ws;
    int* nans = (int*)((char*)ws + nchunks * sizeof(double));
    static bool attr_set = false;   // per instantiation
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void*)chunk_sums_kernel<TIn, C, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chunk_lds<C>()));
        VS_CHECK_HIP(hipFuncSetAttribute((const void*)chunk_sums_kernel<TIn, C, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chunk_lds<C>()));
        attr_set = true;
    }
    if (op == 0) hipLaunchKernelGGL((chunk_sums_kernel<TIn, C, 0>), dim3((unsigned)nchunks), dim3(256), chunk_lds<C>(), s, data, n, (C)0, sums, nans);
    else hipLaunchKernelGGL((chunk_sums_kernel<TIn, C, 1>), dim3((unsigned)nchunks), dim3(256), chunk_lds<C>(), s, data, n, (C)avg, sums, nans);
    VS_LAUNCH_CHECK();
    hipLaunchKernelGGL((sequential_sum_kernel<C>), dim3(1), dim3(256), 0, s, sums, op == 0 ? nans : nullptr, nchunks, out);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

// clip -> subtract -> divide -> clip(0, 1) -> * 255 -> truncate, each step rounded in C as NumPy's in-place ufuncs do
// (base_data_utils.py:270-287); NaN voxels become the mean first (np.nan_to_num(nan=data_mean)); counts of voxels outside
// the bounds are what the reference logs (:259-268)
template <typename TIn, typename C>
__global__ __launch_bounds__(256) void clip_to_uint8_kernel(const TIn* __restrict__ data, int64_t n, C nan_fill, C lower, C upper,
                                                          uint8_t* __restrict__ out, unsigned long long* __restrict__ counts) {
    const C range = subr<C>(upper, lower);
    unsigned above = 0, below = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const TIn raw = data[i];
        C x = (C)raw;
        above += x > upper;      // comparisons with NaN are false, as in NumPy
        below += x < lower;
        if (is_nan<TIn>(raw)) x = nan_fill;
        x = x < lower ? lower : (x > upper ? upper : x);
        x = subr<C>(x, lower);
        x = divr<C>(x, range);
        x = x < (C)0 ? (C)0 : (x > (C)1 ? (C)1 : x);
        x = mulr<C>(x, (C)255);
        out[i] = (uint8_t)x;
    }
    if (counts) {
        for (int o = 32; o; o >>= 1) { above += __shfl_xor(above, o, 64); below += __shfl_xor(below, o, 64); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(counts, (unsigned long long)above); atomicAdd(counts + 1, (unsigned long long)below); }
    }
}

template <typename TIn, typename C>
int run_clip(const TIn* data, int64_t n, double nan_fill, double lower, double upper, uint8_t* out, unsigned long long* counts,
             hipStream_t s) {
    int64_t g = (n + 255) / 256;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL((clip_to_uint8_kernel<TIn, C>), dim3((unsigned)g), dim3(256), 0, s, data, n, (C)nan_fill, (C)lower, (C)upper, out, counts);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

// 2 x 2 x 2 block mean with zero padding at odd edges (skimage.measure.block_reduce(data, (2, 2, 2), np.nanmean) as the reference
// calls it, utilities/base_data_utils.py:161-163): out[z][y][x] = (sum of the block's 8 voxels, missing ones = 0) / 8 in float64.
// Integer volumes only: their block sums are exact in float64 in any order and no voxel is NaN, so this IS NumPy's result.
template <typename TIn, typename C>
__global__ void downsample2x_kernel(const TIn* __restrict__ in, double* __restrict__ out, int D, int H, int W, int d2, int h2, int w2) {
    const int64_t total = (int64_t)d2 * h2 * w2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w2);
        const int64_t r = i / w2;
        const int y = (int)(r % h2), z = (int)(r / h2);
        double acc = 0.0;
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int zz = 2 * z + dz, yy = 2 * y + dy, xx = 2 * x + dx;
                    if (zz < D && yy < H && xx < W) acc += (double)in[((int64_t)zz * H + yy) * W + xx];
                }
        out[i] = acc / 8.0;
    }
}

template <typename TIn, typename C>
int run_downsample(const TIn* data, double* out, int D, int H, int W, hipStream_t s) {
    const int d2 = (D + 1) / 2, h2 = (H + 1) / 2, w2 = (W + 1) / 2;
    int64_t g = ((int64_t)d2 * h2 * w2 + 255) / 256;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL((downsample2x_kernel<TIn, C>), dim3((unsigned)g), dim3(256), 0, s, data, out, D, H, W, d2, h2, w2);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

}  // namespace

extern "C" size_t vs_volume_sum_workspace(int64_t n) {
    const int64_t nchunks = (n + kChunk - 1) / kChunk;
    return (size_t)nchunks * (sizeof(double) + sizeof(int)) + 64;
}

#define VS_VOLUME_DISPATCH(FN, ...)                                                                  \
    switch (vtype) {                                                                                 \
    case VS_VOL_F32: return FN<float, float>((const float*)data, __VA_ARGS__);                       \
    case VS_VOL_F64: return FN<double, double>((const double*)data, __VA_ARGS__);                    \
    case VS_VOL_U8: return FN<uint8_t, double>((const uint8_t*)data, __VA_ARGS__);                   \
    case VS_VOL_I8: return FN<int8_t, double>((const int8_t*)data, __VA_ARGS__);                     \
    case VS_VOL_U16: return FN<uint16_t, double>((const uint16_t*)data, __VA_ARGS__);                \
    case VS_VOL_I16: return FN<int16_t, double>((const int16_t*)data, __VA_ARGS__);                  \
    case VS_VOL_U32: return FN<uint32_t, double>((const uint32_t*)data, __VA_ARGS__);                \
    case VS_VOL_I32: return FN<int32_t, double>((const int32_t*)data, __VA_ARGS__);                  \
    case VS_VOL_I64: return FN<int64_t, double>((const int64_t*)data, __VA_ARGS__);                  \
    case VS_VOL_U64: return FN<uint64_t, double>((const uint64_t*)data, __VA_ARGS__);                \
    default: vs_set_error("volume dtype %d not supported", vtype); return VS_ERR_UNSUPPORTED;        \
    }

extern "C" int vs_volume_sum(int vtype, const void* data, int64_t n, int op, double avg, void* workspace, size_t workspace_bytes,
                             double* out, void* stream) {
    VS_REQUIRE(data && workspace && out && n >= 1, "volume_sum: bad arguments");
    VS_REQUIRE(op == 0 || op == 1, "volume_sum: op must be 0 (sum) or 1 (sum of squared deviations)");
    VS_REQUIRE(workspace_bytes >= vs_volume_sum_workspace(n), "volume_sum: workspace too small");
    VS_REQUIRE((n + kChunk - 1) / kChunk < (1LL << 31), "volume_sum: volume too large");
    hipStream_t s = (hipStream_t)stream;
    VS_VOLUME_DISPATCH(run_sum, n, op, avg, workspace, out, s)
}

extern "C" int vs_clip_to_uint8(int vtype, const void* data, int64_t n, double nan_fill, double lower, double upper, uint8_t* out,
                                uint64_t* counts, void* stream) {
    VS_REQUIRE(data && out && n >= 1, "clip_to_uint8: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* cnt = (unsigned long long*)counts;
    VS_VOLUME_DISPATCH(run_clip, n, nan_fill, lower, upper, out, cnt, s)
}

extern "C" int vs_downsample2x_mean(int vtype, const void* data, double* out, int d, int h, int w, void* stream) {
    VS_REQUIRE(data && out && d >= 1 && h >= 1 && w >= 1, "downsample2x_mean: bad arguments");
    VS_REQUIRE(vtype != VS_VOL_F32 && vtype != VS_VOL_F64, "downsample2x_mean: integer volumes only (float block means depend on NumPy's summation order and NaN handling: host path)");
    hipStream_t s = (hipStream_t)stream;
    VS_VOLUME_DISPATCH(run_downsample, out, d, h, w, s)
}
