// Prediction-side kernels (gfx950, HBM-bound byte/integer work):
//  * slice gather + reflect-101 pad + ImageNet normalisation straight from the resident uint8 volume
//    (VolSeg2dPredictionDataset.__getitem__, data/datasets.py:120-142; augmentations.py:46-65);
//  * softmax -> first-argmax -> max-prob(fp16 RNE) -> centre-crop -> scatter to the voxel address of the
//    (axis, rotation) direction (vol_seg_2d_predictor.py:45-64 and base_data_utils.py:125-138);
//  * the max-probability merge, both in the reference's pairwise form (_merge_vols_in_mem, :90-98) and as
//    the packed (prob, direction, label) key maximum that is order-independent and all-reducible.
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    const int m = 2 * (len - 1);
    p %= m;
    if (p < 0) p += m;
    return p >= len ? m - p : p;
}

// VolSeg2dPredictionDataset.__getitem__ (data/datasets.py:120-142) in numpy's own arithmetic: integer volumes of ANY width are
// converted to float32 and divided by 255 (so a uint16 volume that was not clipped to bytes gives inputs far above 1, as in
// the reference), then - 0.449, / 0.226 in float32; float32 volumes skip the / 255; float64 volumes are normalised in float64
// (the network input is rounded to float32 afterwards - the reference would hand smp a float64 batch).
template <typename VT> __device__ __forceinline__ float normalise_voxel(VT v) {
    return __fdiv_rn(__fsub_rn(__fdiv_rn((float)v, 255.0f), 0.449f), 0.226f);    // (u / 255 - 0.449) / 0.226, every step rounded
}
template <> __device__ __forceinline__ float normalise_voxel<float>(float v) { return __fdiv_rn(__fsub_rn(v, 0.449f), 0.226f); }
template <> __device__ __forceinline__ float normalise_voxel<double>(double v) { return (float)__ddiv_rn(__dsub_rn(v, 0.449), 0.226); }

template <typename VT>
__global__ void slices_gather_kernel(const VT* __restrict__ vol, vs_dirmap m, int s0, int nb, float* __restrict__ x) {
    const int64_t total = (int64_t)nb * m.hp * m.wp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int j = t % m.wp; t /= m.wp;
        const int r = t % m.hp;
        const int b = t / m.hp;
        const int hh = reflect101(r - m.pad_top, m.h), ww = reflect101(j - m.pad_left, m.w);
        x[i] = normalise_voxel<VT>(vol[m.base + (int64_t)(s0 + b) * m.ss + (int64_t)hh * m.sh + (int64_t)ww * m.sw]);
    }
}

__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
    const __half hv = __float2half_rn(f);
    return __builtin_bit_cast(uint16_t, hv);
}

template <int MODE>
__global__ void logits_to_volume_kernel(const float* __restrict__ logits, int classes, vs_dirmap m, int s0, int nb,
                                        int direction, uint8_t* __restrict__ labels, uint16_t* __restrict__ probs,
                                        uint32_t* __restrict__ keys, uint8_t* __restrict__ votes, int64_t nvox) {
    const int64_t total = (int64_t)nb * m.h * m.w;
    const size_t plane = (size_t)m.hp * m.wp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int j = t % m.w; t /= m.w;
        const int r = t % m.h;
        const int b = t / m.h;
        const float* lp = logits + (size_t)b * classes * plane + (size_t)(r + m.crop_top) * m.wp + (j + m.crop_left);
        float mx = lp[0];
        for (int k = 1; k < classes; ++k) mx = fmaxf(mx, lp[k * plane]);
        float sum = 0.f;
        for (int k = 0; k < classes; ++k) sum += expf(lp[k * plane] - mx);
        // argmax over the *probabilities* (first max wins), as torch.argmax(softmax(x)) does
        float best = -1.f;
        int lab = 0;
        for (int k = 0; k < classes; ++k) {
            const float pk = __fdiv_rn(expf(lp[k * plane] - mx), sum);
            if (pk > best) { best = pk; lab = k; }
        }
        const int64_t addr = m.base + (int64_t)(s0 + b) * m.ss + (int64_t)r * m.sh + (int64_t)j * m.sw;
        if (MODE == 0) {
            if (labels) labels[addr] = (uint8_t)lab;
            if (probs) probs[addr] = f32_to_f16_bits(best);
        } else if (MODE == 1) {
            const uint32_t key = ((uint32_t)f32_to_f16_bits(best) << 16) | ((uint32_t)(15 - direction) << 8) | (uint32_t)lab;
            const uint32_t old = keys[addr];
            if (key > old) keys[addr] = key;
        } else {
            votes[(int64_t)lab * nvox + addr] += (MODE == 3 ? 2 : 1);    // (3: this direction stands for itself and its exact repeat)
        }
    }
}

// 64 voxels per thread and trip: 16-byte label loads / stores, 2 x 16-byte probability loads / stores per operand and
// 16 voxels, the 24 loads of a trip issued before the first use (a pure streaming kernel: what it needs is bytes in flight;
// one 16-voxel vector per trip left it at 3.7 TB/s), non-temporal (every byte is touched once).
// Algorithmic traffic 9 B / voxel (read 2 x (u8 + f16), write u8 + f16) - HBM-bound.
__device__ __forceinline__ uint4 ldnt(const uint4* p) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stnt(uint4* p, const uint4& v) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    __builtin_nontemporal_store(u4{v.x, v.y, v.z, v.w}, reinterpret_cast<u4*>(p));
}
// one 32-bit word of two fp16 probabilities: slot 1 wins only when strictly greater (np.argmax: ties keep slot 0);
// returns the selection as a byte mask pair (0xff per winning half)
__device__ __forceinline__ uint32_t merge_pair(uint32_t& pa, uint32_t pb) {
    const float a0 = __half2float(__builtin_bit_cast(__half, (uint16_t)(pa & 0xffff))), a1 = __half2float(__builtin_bit_cast(__half, (uint16_t)(pa >> 16)));
    const float b0 = __half2float(__builtin_bit_cast(__half, (uint16_t)(pb & 0xffff))), b1 = __half2float(__builtin_bit_cast(__half, (uint16_t)(pb >> 16)));
    const uint32_t m = (b0 > a0 ? 0x0000ffffu : 0u) | (b1 > a1 ? 0xffff0000u : 0u);
    pa = (pa & ~m) | (pb & m);
    return (b0 > a0 ? 0x00ffu : 0u) | (b1 > a1 ? 0xff00u : 0u);
}
__global__ __launch_bounds__(256) void merge_maxprob_kernel(uint8_t* __restrict__ l0, uint16_t* __restrict__ p0,
                                                          const uint8_t* __restrict__ l1, const uint16_t* __restrict__ p1,
                                                          int64_t n) {
    constexpr int U = 4;
    const int64_t nvec = n / 16;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v0 < nvec; v0 += U * stride) {
        uint4 la[U], lb[U], pa[U][2], pb[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = v0 + u * stride;
            if (v < nvec) {
                la[u] = ldnt(reinterpret_cast<const uint4*>(l0) + v);
                lb[u] = ldnt(reinterpret_cast<const uint4*>(l1) + v);
                pa[u][0] = ldnt(reinterpret_cast<const uint4*>(p0) + 2 * v); pa[u][1] = ldnt(reinterpret_cast<const uint4*>(p0) + 2 * v + 1);
                pb[u][0] = ldnt(reinterpret_cast<const uint4*>(p1) + 2 * v); pb[u][1] = ldnt(reinterpret_cast<const uint4*>(p1) + 2 * v + 1);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = v0 + u * stride;
            if (v >= nvec) continue;
            uint32_t* law = reinterpret_cast<uint32_t*>(&la[u]);
            const uint32_t* lbw = reinterpret_cast<const uint32_t*>(&lb[u]);
            uint32_t* paw = reinterpret_cast<uint32_t*>(&pa[u][0]);
            const uint32_t* pbw = reinterpret_cast<const uint32_t*>(&pb[u][0]);
#pragma unroll
            for (int w = 0; w < 4; ++w) {             // label word w = voxels 4 w .. 4 w + 3 = probability words 2 w, 2 w + 1
                const uint32_t m = merge_pair(paw[2 * w], pbw[2 * w]) | (merge_pair(paw[2 * w + 1], pbw[2 * w + 1]) << 16);
                law[w] = (law[w] & ~m) | (lbw[w] & m);
            }
            stnt(reinterpret_cast<uint4*>(l0) + v, la[u]);
            stnt(reinterpret_cast<uint4*>(p0) + 2 * v, pa[u][0]);
            stnt(reinterpret_cast<uint4*>(p0) + 2 * v + 1, pa[u][1]);
        }
    }
    // ragged tail
    for (int64_t i = nvec * 16 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float a = __half2float(__builtin_bit_cast(__half, p0[i]));
        const float b = __half2float(__builtin_bit_cast(__half, p1[i]));
        if (b > a) { p0[i] = p1[i]; l0[i] = l1[i]; }
    }
}

__global__ __launch_bounds__(256) void keys_unpack_kernel(const uint32_t* __restrict__ keys, uint8_t* __restrict__ labels,
                                                        uint16_t* __restrict__ probs, int64_t n) {
    const int64_t nvec = n / 4;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        const uint4 k = reinterpret_cast<const uint4*>(keys)[v];
        if (labels)
            reinterpret_cast<uint32_t*>(labels)[v] = (k.x & 0xff) | ((k.y & 0xff) << 8) | ((k.z & 0xff) << 16) | ((k.w & 0xff) << 24);
        if (probs) reinterpret_cast<uint2*>(probs)[v] = make_uint2((k.x >> 16) | (k.y & 0xffff0000u), (k.z >> 16) | (k.w & 0xffff0000u));
    }
    for (int64_t i = nvec * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t k = keys[i];
        if (labels) labels[i] = (uint8_t)(k & 0xff);
        if (probs) probs[i] = (uint16_t)(k >> 16);
    }
}

// Keys of one batch of slices, staged by the head kernel as stage[n][hp][wp], into the key volume.  For a direction whose
// slice index is the volume's contiguous axis, one slice's pixels are a plane stride apart - written from the head kernel
// every voxel is its own memory transaction.  Here a workgroup transposes a 64-pixel x 64-slice tile through LDS: reads are
// contiguous along the pixels of a slice, the max-merge runs along the slices (lane = slice: 256 contiguous bytes per pixel).
__global__ __launch_bounds__(256) void keys_stage_scatter_kernel(const uint32_t* __restrict__ stage, int nb, vs_dirmap m, int s0,
                                                               uint32_t* __restrict__ keys) {
    __shared__ uint32_t tile[64][65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int npx = m.h * m.w;
    const int64_t plane = (int64_t)m.hp * m.wp;
    for (int t0 = blockIdx.x * 64; t0 < npx; t0 += gridDim.x * 64) {
        for (int n0 = 0; n0 < nb; n0 += 64) {
            const int px = t0 + lane;
            const int r = px / m.w, j = px - r * m.w;
            const int64_t src = (int64_t)(r + m.crop_top) * m.wp + (j + m.crop_left);
            __syncthreads();
            for (int k = wave; k < 64; k += 4)
                if (n0 + k < nb && px < npx) tile[k][lane] = stage[(n0 + k) * plane + src];
            __syncthreads();
            for (int k = wave; k < 64; k += 4) {
                const int q = t0 + k;
                if (q >= npx || n0 + lane >= nb) continue;
                const int rq = q / m.w, jq = q - rq * m.w;
                const int64_t addr = m.base + (int64_t)(s0 + n0 + lane) * m.ss + (int64_t)rq * m.sh + (int64_t)jq * m.sw;
                const uint32_t key = tile[lane][k], old = keys[addr];
                if (key > old) keys[addr] = key;     // one lane per voxel and launch; launches are ordered by the stream
            }
        }
    }
}

inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

int check_map(const vs_dirmap* m, int s0, int nb) {
    VS_REQUIRE(m && m->depth > 0 && m->h > 0 && m->w > 0 && m->hp >= m->h && m->wp >= m->w, "dirmap: bad dims");
    VS_REQUIRE(s0 >= 0 && nb > 0 && s0 + nb <= m->depth, "dirmap: slice range [%d,%d) outside depth %d", s0, s0 + nb, m->depth);
    VS_REQUIRE(m->crop_top >= 0 && m->crop_top + m->h <= m->hp && m->crop_left >= 0 && m->crop_left + m->w <= m->wp,
               "dirmap: crop window outside the padded slice");
    return VS_OK;
}

}  // namespace

extern "C" int vs_slices_gather_typed(int vtype, const void* vol, const vs_dirmap* m, int s0, int nb, float* x, void* stream) {
    int rc = check_map(m, s0, nb);
    if (rc) return rc;
    VS_REQUIRE(vol && x, "slices_gather: null pointer");
    const int64_t total = (int64_t)nb * m->hp * m->wp;
    const dim3 grid(grid_for(total));
    hipStream_t s = (hipStream_t)stream;
#define VS_GATHER(T) hipLaunchKernelGGL(slices_gather_kernel<T>, grid, dim3(256), 0, s, (const T*)vol, *m, s0, nb, x)
    switch (vtype) {
    case VS_VOL_U8: VS_GATHER(uint8_t); break;
    case VS_VOL_I8: VS_GATHER(int8_t); break;
    case VS_VOL_U16: VS_GATHER(uint16_t); break;
    case VS_VOL_I16: VS_GATHER(int16_t); break;
    case VS_VOL_U32: VS_GATHER(uint32_t); break;
    case VS_VOL_I32: VS_GATHER(int32_t); break;
    case VS_VOL_I64: VS_GATHER(int64_t); break;
    case VS_VOL_U64: VS_GATHER(uint64_t); break;
    case VS_VOL_F32: VS_GATHER(float); break;
    case VS_VOL_F64: VS_GATHER(double); break;
    default: vs_set_error("slices_gather: unsupported volume type %d", vtype); return VS_ERR_UNSUPPORTED;
    }
#undef VS_GATHER
    VS_LAUNCH_CHECK();
    return VS_OK;
}

extern "C" int vs_slices_gather(const uint8_t* vol, const vs_dirmap* m, int s0, int nb, float* x, void* stream) {
    return vs_slices_gather_typed(VS_VOL_U8, vol, m, s0, nb, x, stream);
}

extern "C" int vs_logits_to_volume(const float* logits, int classes, const vs_dirmap* m, int s0, int nb, int mode,
                                   int direction, uint8_t* labels, uint16_t* probs, uint32_t* keys, uint8_t* votes,
                                   int64_t nvox, void* stream) {
    int rc = check_map(m, s0, nb);
    if (rc) return rc;
    VS_REQUIRE(logits && classes >= 1 && classes <= 255, "logits_to_volume: bad arguments");
    VS_REQUIRE(direction >= 0 && direction < 16, "logits_to_volume: direction %d out of range", direction);
    const int64_t total = (int64_t)nb * m->h * m->w;
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) {
        hipLaunchKernelGGL(logits_to_volume_kernel<0>, dim3(grid_for(total)), dim3(256), 0, s, logits, classes, *m, s0, nb,
                           direction, labels, probs, keys, votes, nvox);
    } else if (mode == 1) {
        VS_REQUIRE(keys, "logits_to_volume: mode 1 needs a key volume");
        hipLaunchKernelGGL(logits_to_volume_kernel<1>, dim3(grid_for(total)), dim3(256), 0, s, logits, classes, *m, s0, nb,
                           direction, labels, probs, keys, votes, nvox);
    } else if (mode == 2) {
        VS_REQUIRE(votes && nvox > 0, "logits_to_volume: mode 2 needs a vote volume");
        hipLaunchKernelGGL(logits_to_volume_kernel<2>, dim3(grid_for(total)), dim3(256), 0, s, logits, classes, *m, s0, nb,
                           direction, labels, probs, keys, votes, nvox);
    } else if (mode == 3) {     // two votes: a direction of the 12-way scheme that another one repeats exactly (not run)
        VS_REQUIRE(votes && nvox > 0, "logits_to_volume: mode 3 needs a vote volume");
        hipLaunchKernelGGL(logits_to_volume_kernel<3>, dim3(grid_for(total)), dim3(256), 0, s, logits, classes, *m, s0, nb,
                           direction, labels, probs, keys, votes, nvox);
    } else {
        vs_set_error("logits_to_volume: bad mode %d", mode);
        return VS_ERR_INVALID;
    }
    VS_LAUNCH_CHECK();
    return VS_OK;
}

int launch_keys_stage_scatter(const uint32_t* stage, int nb, const vs_dirmap& m, int s0, uint32_t* keys, hipStream_t s) {
    VS_REQUIRE(stage && keys && nb >= 1, "keys_stage_scatter: bad arguments");
    const int tiles = (m.h * m.w + 63) / 64;
    hipLaunchKernelGGL(keys_stage_scatter_kernel, dim3(tiles > 8192 ? 8192 : tiles), dim3(256), 0, s, stage, nb, m, s0, keys);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

extern "C" int vs_merge_maxprob(uint8_t* label0, uint16_t* prob0, const uint8_t* label1, const uint16_t* prob1,
                                int64_t n, void* stream) {
    VS_REQUIRE(label0 && prob0 && label1 && prob1 && n >= 0, "merge_maxprob: bad arguments");
    if (n == 0) return VS_OK;
    VS_REQUIRE(((uintptr_t)label0 | (uintptr_t)label1 | (uintptr_t)prob0 | (uintptr_t)prob1) % 16 == 0,
               "merge_maxprob: volumes must be 16-byte aligned");
    hipLaunchKernelGGL(merge_maxprob_kernel, dim3(grid_for(n / 64 + 1)), dim3(256), 0, (hipStream_t)stream, label0, prob0, label1, prob1, n);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

extern "C" int vs_keys_unpack(const uint32_t* keys, uint8_t* labels, uint16_t* probs, int64_t n, void* stream) {
    VS_REQUIRE(keys && n >= 0, "keys_unpack: bad arguments");
    if (n == 0) return VS_OK;
    hipLaunchKernelGGL(keys_unpack_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, keys, labels, probs, n);
    VS_LAUNCH_CHECK();
    return VS_OK;
}
