// DiceLoss(normalization="none") of the reference (volume_segmantics/data/pytorch3dunet_losses.py:15-41,89-135; selected
// at vol_seg_2d_trainer.py:133-135) and its gradient, fused: per class c over all (n, pixel):
//   I_c = sum x t,  D_c = sum x^2 + sum t^2,  loss = 1 - mean_c 2 I_c / max(D_c, eps)
//   dloss/dx = -(2/K) * (t D_c - 2 x I_c) / D_c^2      (D_c > eps;  -(2/K) t / eps otherwise)
// x: logits (N, K, H, W) fp32 NCHW (what vs_unet_forward returns), t: one-hot targets (N, K, H, W) uint8 or fp32.
// Two HBM sweeps (reduce, then gradient) instead of ~20 elementwise torch kernels; sums finalised in fp64, fixed order.
#include <algorithm>

#include "common.h"

namespace {

constexpr int kBlocks = 512;

template <typename TT> __device__ __forceinline__ float tval(const TT* t, size_t i) { return (float)t[i]; }

template <typename TT>
__global__ __launch_bounds__(256) void dice_partial_kernel(const float* __restrict__ x, const TT* __restrict__ t, int n, int k,
                                                         int64_t hw, float* __restrict__ partial) {
    __shared__ float red[3][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = 0; c < k; ++c) {
        float si = 0.f, sx = 0.f, st = 0.f;
        for (int b = 0; b < n; ++b) {
            const size_t base = ((size_t)b * k + c) * hw;
            for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
                const float xv = x[base + i], tv = tval(t, base + i);
                si += xv * tv; sx += xv * xv; st += tv * tv;
            }
        }
        si = wave_sum(si); sx = wave_sum(sx); st = wave_sum(st);
        if (lane == 0) { red[0][wave] = si; red[1][wave] = sx; red[2][wave] = st; }
        __syncthreads();
        if (threadIdx.x < 3)
            partial[((size_t)blockIdx.x * k + c) * 3 + threadIdx.x] =
                (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
        __syncthreads();
    }
}

// The same sums with 16-byte loads, four (logits, targets) pairs in flight per thread (hw % 4 == 0): the scalar form above is one
// dependent 4-byte load pair per (sample, class) and thread - 64 round trips, 31 us for the 17 MB of a batch-32 step; this one is
// one trip per class.
template <typename TT> struct TV4;
template <> struct TV4<float> { typedef float4 type; static __device__ __forceinline__ float4 f(float4 v) { return v; } };
template <> struct TV4<uint8_t> {
    typedef uchar4 type;
    static __device__ __forceinline__ float4 f(uchar4 v) { return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w); }
};
template <typename TT>
__global__ __launch_bounds__(256) void dice_partial4_kernel(const float* __restrict__ x, const TT* __restrict__ t, int n, int k,
                                                          int64_t hw, float* __restrict__ partial) {
    __shared__ float red[3][4];
    typedef typename TV4<TT>::type TQ;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nv = hw >> 2, total = (int64_t)n * nv, stride = (int64_t)gridDim.x * 256;
    for (int c = 0; c < k; ++c) {
        float si = 0.f, sx = 0.f, st = 0.f;
        for (int64_t v0 = (int64_t)blockIdx.x * 256 + threadIdx.x; v0 < total; v0 += 4 * stride) {
            float4 xv[4], tv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t v = v0 + u * stride;
                const bool ok = v < total;
                const int64_t b = ok ? v / nv : 0, i = ok ? v - b * nv : 0;
                const size_t at = ((size_t)b * k + c) * hw + (size_t)i * 4;
                xv[u] = ok ? *reinterpret_cast<const float4*>(x + at) : make_float4(0.f, 0.f, 0.f, 0.f);
                tv[u] = ok ? TV4<TT>::f(*reinterpret_cast<const TQ*>(t + at)) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                si += xv[u].x * tv[u].x + xv[u].y * tv[u].y + xv[u].z * tv[u].z + xv[u].w * tv[u].w;
                sx += xv[u].x * xv[u].x + xv[u].y * xv[u].y + xv[u].z * xv[u].z + xv[u].w * xv[u].w;
                st += tv[u].x * tv[u].x + tv[u].y * tv[u].y + tv[u].z * tv[u].z + tv[u].w * tv[u].w;
            }
        }
        si = wave_sum(si); sx = wave_sum(sx); st = wave_sum(st);
        if (lane == 0) { red[0][wave] = si; red[1][wave] = sx; red[2][wave] = st; }
        __syncthreads();
        if (threadIdx.x < 3)
            partial[((size_t)blockIdx.x * k + c) * 3 + threadIdx.x] =
                (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
        __syncthreads();
    }
}

// one wave per class; stats[c] = {I_c, D_c}; loss written by class 0's wave after all classes are known -> single block
__global__ __launch_bounds__(64) void dice_finalize_kernel(const float* __restrict__ partial, int nblocks, int k, float eps,
                                                         float* __restrict__ stats, float* __restrict__ loss) {
    const int lane = threadIdx.x;
    double acc = 0.0;
    for (int c = 0; c < k; ++c) {
        double si = 0.0, sx = 0.0, st = 0.0;
        for (int b = lane; b < nblocks; b += 64) {
            const float* p = partial + ((size_t)b * k + c) * 3;
            si += p[0]; sx += p[1]; st += p[2];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { si += __shfl_xor(si, o, 64); sx += __shfl_xor(sx, o, 64); st += __shfl_xor(st, o, 64); }
        const double d = sx + st;
        if (lane == 0) { stats[2 * c] = (float)si; stats[2 * c + 1] = (float)d; }
        acc += 2.0 * si / (d > (double)eps ? d : (double)eps);
    }
    if (lane == 0) *loss = (float)(1.0 - acc / k);
}

template <typename TT>
__global__ __launch_bounds__(256) void dice_grad_kernel(const float* __restrict__ x, const TT* __restrict__ t,
                                                      const float* __restrict__ stats, const float* __restrict__ gout,
                                                      float eps, int n, int k, int64_t hw, float* __restrict__ dx) {
    const int64_t total = (int64_t)n * k * hw;
    const float g = (gout ? *gout : 1.f) * (-2.f / (float)k);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i / hw) % k);
        const float I = stats[2 * c], D = stats[2 * c + 1];
        const float xv = x[i], tv = tval(t, i);
        dx[i] = D > eps ? g * (tv * D - 2.f * xv * I) / (D * D) : g * tv / eps;
    }
}


// 16 bytes per lane, two vectors in flight (hw % 4 == 0: a vector never straddles two class planes)
template <typename TT>
__global__ __launch_bounds__(256) void dice_grad4_kernel(const float* __restrict__ x, const TT* __restrict__ t,
                                                       const float* __restrict__ stats, const float* __restrict__ gout,
                                                       float eps, int n, int k, int64_t hw, float* __restrict__ dx) {
    typedef typename TV4<TT>::type TQ;
    const int64_t nv = hw >> 2, total = (int64_t)n * k * nv, stride = (int64_t)gridDim.x * 256;
    const float g = (gout ? *gout : 1.f) * (-2.f / (float)k);
    for (int64_t v0 = (int64_t)blockIdx.x * 256 + threadIdx.x; v0 < total; v0 += 2 * stride) {
        float4 xv[2], tv[2];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t v = v0 + u * stride;
            ok[u] = v < total;
            xv[u] = ok[u] ? *reinterpret_cast<const float4*>(x + v * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            tv[u] = ok[u] ? TV4<TT>::f(*reinterpret_cast<const TQ*>(t + v * 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!ok[u]) continue;
            const int64_t v = v0 + u * stride;
            const int c = (int)((v / nv) % k);
            const float I = stats[2 * c], D = stats[2 * c + 1];
            float4 o;
            if (D > eps) {
                const float dd = D * D;
                o.x = g * (tv[u].x * D - 2.f * xv[u].x * I) / dd; o.y = g * (tv[u].y * D - 2.f * xv[u].y * I) / dd;
                o.z = g * (tv[u].z * D - 2.f * xv[u].z * I) / dd; o.w = g * (tv[u].w * D - 2.f * xv[u].w * I) / dd;
            } else {
                o.x = g * tv[u].x / eps; o.y = g * tv[u].y / eps; o.z = g * tv[u].z / eps; o.w = g * tv[u].w / eps;
            }
            *reinterpret_cast<float4*>(dx + v * 4) = o;
        }
    }
}


// ---- MeanIoU (data/pytorch3dunet_metrics.py:34-106; validation metric of vol_seg_2d_trainer.py:150-161,243) ----------------
// Per sample: prediction = one-hot of the first arg-max over channels (input > 0.5 for a single channel), per class
// |P & T| / max(|P | T|, 1e-8) on the byte-converted target, mean over classes, then over samples.  One sweep: integer
// counts per (sample, class) through wave reductions and integer atomics (exact, order-free), then one tiny finalise.
template <typename TT>
__global__ __launch_bounds__(256) void mean_iou_count_kernel(const float* __restrict__ x, const TT* __restrict__ t, int c, int64_t hw,
                                                           int from_logits, int* __restrict__ counts) {
    constexpr int kMax = 16;
    const int b = blockIdx.y;
    int inter[kMax], uni[kMax];
#pragma unroll
    for (int k = 0; k < kMax; ++k) inter[k] = uni[k] = 0;
    const size_t base = (size_t)b * c * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
        int arg = 0;
        if (c == 1) {
            arg = x[base + i] > 0.5f ? 0 : -1;
        } else if (!from_logits) {
            float best = x[base + i];
            for (int k = 1; k < c; ++k) {
                const float v = x[base + (size_t)k * hw + i];
                if (v > best) { best = v; arg = k; }     // strict: the first maximum wins
            }
        } else {
            // the trainer hands the metric softmax(logits, dim=1): two distinct logits may round to the SAME probability,
            // and then the first one wins - so the arg-max is taken over fp32 probabilities, as in vs_logits_to_volume
            float m = x[base + i];
            for (int k = 1; k < c; ++k) m = fmaxf(m, x[base + (size_t)k * hw + i]);
            float sum = 0.f;
            for (int k = 0; k < c; ++k) sum += expf(x[base + (size_t)k * hw + i] - m);
            float best = -1.f;
            for (int k = 0; k < c; ++k) {
                const float pr = __fdiv_rn(expf(x[base + (size_t)k * hw + i] - m), sum);
                if (pr > best) { best = pr; arg = k; }
            }
        }
#pragma unroll
        for (int k = 0; k < kMax; ++k) {
            if (k < c) {
                const int p = k == arg ? 1 : 0;
                const int tb = (int)(unsigned char)t[base + (size_t)k * hw + i];   // `_target.byte()`
                inter[k] += p & tb;
                uni[k] += p | tb;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kMax; ++k) {
        int a = inter[k], u = uni[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); u += __shfl_xor(u, o, 64); }
        if (k < c && (threadIdx.x & 63) == 0) {
            atomicAdd(counts + ((size_t)b * c + k) * 2 + 0, a);
            atomicAdd(counts + ((size_t)b * c + k) * 2 + 1, u);
        }
    }
}

__global__ void mean_iou_finalize_kernel(const int* __restrict__ counts, int n, int c, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float total = 0.f;
    for (int b = 0; b < n; ++b) {
        float s = 0.f;
        for (int k = 0; k < c; ++k)
            s += (float)counts[((size_t)b * c + k) * 2] / fmaxf((float)counts[((size_t)b * c + k) * 2 + 1], 1e-8f);
        total += s / (float)c;
    }
    *out = total / (float)n;
}


// labels (n, hw) uint8 -> one-hot (n, K, hw) uint8 (prepare_training_batch, utilities/base_data_utils.py:150-158)
__global__ void onehot_kernel(const uint8_t* __restrict__ lab, int k, int64_t hw, uint8_t* __restrict__ out) {
    const int b = blockIdx.y;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < hw; i += (int64_t)gridDim.x * blockDim.x * 4) {
        if (i + 4 <= hw && (hw & 3) == 0) {
            const uchar4 l = *reinterpret_cast<const uchar4*>(lab + (size_t)b * hw + i);
            for (int c = 0; c < k; ++c)
                *reinterpret_cast<uchar4*>(out + ((size_t)b * k + c) * hw + i) =
                    make_uchar4(l.x == c, l.y == c, l.z == c, l.w == c);
        } else {
            for (int64_t j = i; j < min(hw, i + 4); ++j)
                for (int c = 0; c < k; ++c) out[((size_t)b * k + c) * hw + j] = lab[(size_t)b * hw + j] == c;
        }
    }
}

}  // namespace

extern "C" size_t vs_dice_workspace(int classes) { return ((size_t)kBlocks * classes * 3 + 2 * (size_t)classes) * sizeof(float); }

// loss (1 float, device) and per-class stats are produced by the forward; the backward multiplies by *grad_out (device
// scalar, may be null = 1).  target_is_f32: targets are fp32 (the reference passes targets.float()) instead of uint8.
extern "C" int vs_dice_loss_fwd(const float* logits, const void* targets, int target_is_f32, int n, int classes, int64_t hw,
                                float eps, float* loss, float* workspace, size_t workspace_bytes, void* stream) {
    VS_REQUIRE(logits && targets && loss && workspace && workspace_bytes >= vs_dice_workspace(classes) && classes >= 1,
               "dice_loss_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    float* stats = workspace + (size_t)kBlocks * classes * 3;
    const bool vec4 = (hw & 3) == 0 && ((uintptr_t)logits & 15) == 0 && ((uintptr_t)targets & (target_is_f32 ? 15 : 3)) == 0;
    if (vec4 && target_is_f32)
        hipLaunchKernelGGL(dice_partial4_kernel<float>, dim3(kBlocks), dim3(256), 0, s, logits, (const float*)targets, n, classes, hw, workspace);
    else if (vec4)
        hipLaunchKernelGGL(dice_partial4_kernel<uint8_t>, dim3(kBlocks), dim3(256), 0, s, logits, (const uint8_t*)targets, n, classes, hw, workspace);
    else if (target_is_f32)
        hipLaunchKernelGGL(dice_partial_kernel<float>, dim3(kBlocks), dim3(256), 0, s, logits, (const float*)targets, n, classes, hw, workspace);
    else
        hipLaunchKernelGGL(dice_partial_kernel<uint8_t>, dim3(kBlocks), dim3(256), 0, s, logits, (const uint8_t*)targets, n, classes, hw, workspace);
    VS_LAUNCH_CHECK();
    hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(64), 0, s, workspace, kBlocks, classes, eps, stats, loss);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

extern "C" int vs_dice_loss_bwd(const float* logits, const void* targets, int target_is_f32, const float* grad_out, int n,
                                int classes, int64_t hw, float eps, const float* workspace, float* dlogits, void* stream) {
    VS_REQUIRE(logits && targets && workspace && dlogits, "dice_loss_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const float* stats = workspace + (size_t)kBlocks * classes * 3;
    const int64_t total = (int64_t)n * classes * hw;
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    const bool vec4 = (hw & 3) == 0 && ((uintptr_t)logits & 15) == 0 && ((uintptr_t)dlogits & 15) == 0 && ((uintptr_t)targets & (target_is_f32 ? 15 : 3)) == 0;
    const int grid4 = (int)std::min<int64_t>(4096, (total / 4 + 511) / 512);
    if (vec4 && target_is_f32)
        hipLaunchKernelGGL(dice_grad4_kernel<float>, dim3(grid4), dim3(256), 0, s, logits, (const float*)targets, stats, grad_out, eps, n, classes, hw, dlogits);
    else if (vec4)
        hipLaunchKernelGGL(dice_grad4_kernel<uint8_t>, dim3(grid4), dim3(256), 0, s, logits, (const uint8_t*)targets, stats, grad_out, eps, n, classes, hw, dlogits);
    else if (target_is_f32)
        hipLaunchKernelGGL(dice_grad_kernel<float>, dim3(grid), dim3(256), 0, s, logits, (const float*)targets, stats, grad_out, eps, n, classes, hw, dlogits);
    else
        hipLaunchKernelGGL(dice_grad_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, logits, (const uint8_t*)targets, stats, grad_out, eps, n, classes, hw, dlogits);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

extern "C" size_t vs_mean_iou_workspace(int n, int classes) { return (size_t)n * classes * 2 * sizeof(int); }

extern "C" int vs_mean_iou(const float* input, const void* targets, int target_is_f32, int from_logits, int n, int classes, int64_t hw,
                           float* out, void* workspace, size_t workspace_bytes, void* stream) {
    VS_REQUIRE(input && targets && out && n >= 1 && classes >= 1 && classes <= 16 && hw >= 1, "mean_iou: bad arguments");
    VS_REQUIRE(workspace && workspace_bytes >= vs_mean_iou_workspace(n, classes), "mean_iou: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    VS_CHECK_HIP(hipMemsetAsync(workspace, 0, vs_mean_iou_workspace(n, classes), s));
    const dim3 grid((unsigned)std::min<int64_t>(64, (hw + 255) / 256), (unsigned)n);
    if (target_is_f32)
        hipLaunchKernelGGL(mean_iou_count_kernel<float>, grid, dim3(256), 0, s, input, (const float*)targets, classes, hw, from_logits, (int*)workspace);
    else
        hipLaunchKernelGGL(mean_iou_count_kernel<uint8_t>, grid, dim3(256), 0, s, input, (const uint8_t*)targets, classes, hw, from_logits, (int*)workspace);
    VS_LAUNCH_CHECK();
    hipLaunchKernelGGL(mean_iou_finalize_kernel, dim3(1), dim3(64), 0, s, (const int*)workspace, n, classes, out);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

extern "C" int vs_onehot_u8(const uint8_t* labels, int n, int classes, int64_t hw, uint8_t* onehot, void* stream) {
    VS_REQUIRE(labels && onehot && n >= 1 && classes >= 1 && classes <= 255 && hw >= 1, "onehot: bad arguments");
    const dim3 grid((unsigned)std::min<int64_t>(256, (hw / 4 + 255) / 256 + 1), (unsigned)n);
    hipLaunchKernelGGL(onehot_kernel, grid, dim3(256), 0, (hipStream_t)stream, labels, classes, hw, onehot);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

// ---- the other selectable criteria, fused (vol_seg_2d_trainer.py:124-148) ------------------------------------------------------
// kind 1: BCEDiceLoss(alpha, beta) = alpha * BCEWithLogitsLoss + beta * DiceLoss(normalization = sigmoid)  (pytorch3dunet_losses.py:171-184)
// kind 2: BCEWithLogitsLoss (mean over all elements)
// kind 3: CrossEntropyLoss over the channel dimension, mean over pixels; the class index is the position of the 1 in the one-hot target
// kind 4: GeneralizedDiceLoss(normalization = sigmoid, epsilon)  (pytorch3dunet_losses.py:138-169; one channel -> (p, 1-p) / (t, 1-t))
// Same structure as the Dice kernels above: ONE reduction sweep over logits + targets (per-class sums of p t, p^2, t^2, p, t with
// p = sigmoid(x), plus the BCE / CE sum), a one-block finalise in fp64 that also leaves per-class gradient coefficients, and ONE
// gradient sweep: dL/dx = (a_c t + b_c p + g_c) p (1 - p) + s (p - t)  (kinds 1, 2, 4) or s (softmax - t) (kind 3).
namespace {

constexpr int kSegBlocks = 512, kSegSums = 6;   // per class: S_pt, S_pp, S_tt, S_p, S_t, BCE or CE term

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// BCEWithLogits element: max(x, 0) - x t + log(1 + exp(-|x|))   (torch's stable form)
__device__ __forceinline__ float bce_elem(float x, float t) { return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))); }

template <typename TT>
__global__ __launch_bounds__(256) void seg_partial_kernel(const float* __restrict__ x, const TT* __restrict__ t, int kind, int n, int k,
                                                        int64_t hw, float* __restrict__ partial) {
    __shared__ float red[kSegSums][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = 0; c < k; ++c) {
        float s[kSegSums] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < n; ++b) {
            const size_t base = ((size_t)b * k + c) * hw;
            for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
                const float xv = x[base + i], tv = tval(t, base + i);
                if (kind == 3) {
                    if (c == 0) {   // cross entropy is per pixel across the classes: class 0's pass carries it
                        const size_t p0 = (size_t)b * k * hw + i;
                        float m = x[p0], xt = 0.f;
                        for (int q = 1; q < k; ++q) m = fmaxf(m, x[p0 + (size_t)q * hw]);
                        float se = 0.f;
                        bool found = false;
                        for (int q = 0; q < k; ++q) {
                            const float xq = x[p0 + (size_t)q * hw];
                            se += expf(xq - m);
                            if (!found && tval(t, p0 + (size_t)q * hw) != 0.f) { xt = xq; found = true; }   // first 1 = argmax of the one-hot
                        }
                        if (!found) xt = x[p0];                                                                   // all-zero column: argmax = 0
                        s[5] += (m + logf(se)) - xt;
                    }
                } else {
                    const float p = sigmoidf_(xv);
                    s[0] += p * tv; s[1] += p * p; s[2] += tv * tv; s[3] += p; s[4] += tv;
                    if (kind <= 2) s[5] += bce_elem(xv, tv);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < kSegSums; ++q) {
            s[q] = wave_sum(s[q]);
            if (lane == 0) red[q][wave] = s[q];
        }
        __syncthreads();
        if (threadIdx.x < kSegSums)
            partial[((size_t)blockIdx.x * k + c) * kSegSums + threadIdx.x] =
                (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
        __syncthreads();
    }
}

// one wave: per-class sums in fp64 (fixed order), the loss, and coef[c] = {a_c, b_c, g_c} + scal = {s}
__global__ __launch_bounds__(64) void seg_finalize_kernel(const float* __restrict__ partial, int nblocks, int kind, int k, double m_elems,
                                                        double n_pix, float alpha, float beta, float eps, float* __restrict__ coef,
                                                        float* __restrict__ loss) {
    const int lane = threadIdx.x;
    __shared__ double S[16][kSegSums];
    for (int c = 0; c < k; ++c) {
        double s[kSegSums] = {0, 0, 0, 0, 0, 0};
        for (int b = lane; b < nblocks; b += 64) {
            const float* p = partial + ((size_t)b * k + c) * kSegSums;
#pragma unroll
            for (int q = 0; q < kSegSums; ++q) s[q] += p[q];
        }
#pragma unroll
        for (int q = 0; q < kSegSums; ++q) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s[q] += __shfl_xor(s[q], o, 64);
            if (lane == 0) S[c][q] = s[q];
        }
    }
    __syncthreads();
    if (lane != 0) return;
    double total = 0.0, bce = 0.0;
    for (int c = 0; c < k; ++c) bce += S[c][5];
    float* scal = coef + 16 * 4;
    scal[0] = 0.f;
    for (int c = 0; c < 16; ++c) coef[c * 4 + 0] = coef[c * 4 + 1] = coef[c * 4 + 2] = coef[c * 4 + 3] = 0.f;
    if (kind == 1 || kind == 2) {
        const double wb = kind == 1 ? (double)alpha : 1.0;
        total += wb * bce / m_elems;
        scal[0] = (float)(wb / m_elems);                         // d/dx of the mean BCE: (p - t) / M
        if (kind == 1) {
            double acc = 0.0;
            for (int c = 0; c < k; ++c) {
                const double I = S[c][0], D = S[c][1] + S[c][2];
                const bool big = D > (double)eps;
                const double Dc = big ? D : (double)eps;
                acc += 2.0 * I / Dc;
                // d(1 - mean_c 2 I / D)/dp = -(2/K) (t D - 2 p I) / D^2 ; with D clamped: -(2/K) t / eps
                coef[c * 4 + 0] = (float)(beta * (-2.0 / k) / Dc);
                coef[c * 4 + 1] = big ? (float)(beta * (4.0 / k) * I / (D * D)) : 0.f;
            }
            total += beta * (1.0 - acc / k);
        }
    } else if (kind == 3) {
        total = bce / n_pix;
        scal[0] = (float)(1.0 / n_pix);
    } else {   // generalised Dice
        const int kk = k == 1 ? 2 : k;
        double Spt[16], Sp[16], St[16];
        for (int c = 0; c < k; ++c) { Spt[c] = S[c][0]; Sp[c] = S[c][3]; St[c] = S[c][4]; }
        if (k == 1) { Spt[1] = n_pix - Sp[0] - St[0] + Spt[0]; Sp[1] = n_pix - Sp[0]; St[1] = n_pix - St[0]; }
        double inter = 0.0, denom = 0.0, w[16];
        bool live[16];
        for (int c = 0; c < kk; ++c) {
            const double v = St[c] * St[c];
            w[c] = 1.0 / (v > (double)eps ? v : (double)eps);
            inter += Spt[c] * w[c];
            const double d = (Sp[c] + St[c]) * w[c];
            live[c] = d > (double)eps;
            denom += live[c] ? d : (double)eps;
        }
        total = 1.0 - 2.0 * inter / denom;
        // dL/dp_c = -2 [ w_c t denom - inter w_c 1{live} ] / denom^2 = a_c t + g_c
        double a[16], gg[16];
        for (int c = 0; c < kk; ++c) { a[c] = -2.0 * w[c] / denom; gg[c] = live[c] ? 2.0 * inter * w[c] / (denom * denom) : 0.0; }
        if (k == 1) {   // p' = 1 - p, t' = 1 - t:  dL/dp = (a0 t + g0) - (a1 (1 - t) + g1) = (a0 + a1) t + (g0 - a1 - g1)
            coef[0] = (float)(a[0] + a[1]); coef[2] = (float)(gg[0] - a[1] - gg[1]);
        } else {
            for (int c = 0; c < k; ++c) { coef[c * 4 + 0] = (float)a[c]; coef[c * 4 + 2] = (float)gg[c]; }
        }
    }
    *loss = (float)total;
}

template <typename TT>
__global__ __launch_bounds__(256) void seg_grad_kernel(const float* __restrict__ x, const TT* __restrict__ t, const float* __restrict__ coef,
                                                     const float* __restrict__ gout, int kind, int n, int k, int64_t hw,
                                                     float* __restrict__ dx) {
    const float g = gout ? *gout : 1.f;
    const float sc = coef[16 * 4];
    if (kind == 3) {
        const int64_t total = (int64_t)n * hw;
        for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < total; j += (int64_t)gridDim.x * 256) {
            const int64_t b = j / hw, i = j - b * hw;
            const size_t p0 = (size_t)b * k * hw + i;
            float m = x[p0];
            for (int q = 1; q < k; ++q) m = fmaxf(m, x[p0 + (size_t)q * hw]);
            float se = 0.f;
            int cls = 0;
            bool found = false;
            for (int q = 0; q < k; ++q) {
                se += expf(x[p0 + (size_t)q * hw] - m);
                if (!found && tval(t, p0 + (size_t)q * hw) != 0.f) { cls = q; found = true; }
            }
            for (int q = 0; q < k; ++q)
                dx[p0 + (size_t)q * hw] = g * sc * (expf(x[p0 + (size_t)q * hw] - m) / se - (q == cls ? 1.f : 0.f));
        }
        return;
    }
    const int64_t total = (int64_t)n * k * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i / hw) % k);
        const float xv = x[i], tv = tval(t, i);
        const float p = sigmoidf_(xv);
        const float dldp = coef[c * 4 + 0] * tv + coef[c * 4 + 1] * p + coef[c * 4 + 2];
        dx[i] = g * (dldp * p * (1.f - p) + sc * (p - tv));
    }
}

}  // namespace

extern "C" size_t vs_seg_loss_workspace(int classes) { return ((size_t)kSegBlocks * classes * kSegSums + 16 * 4 + 4) * sizeof(float); }

extern "C" int vs_seg_loss_fwd(int kind, const float* logits, const void* targets, int target_is_f32, int n, int classes, int64_t hw,
                               float alpha, float beta, float eps, float* loss, float* workspace, size_t workspace_bytes, void* stream) {
    VS_REQUIRE(kind >= 1 && kind <= 4, "seg_loss: kind must be 1 (BCE-Dice), 2 (BCE), 3 (cross entropy) or 4 (generalised Dice)");
    VS_REQUIRE(logits && targets && loss && workspace && workspace_bytes >= vs_seg_loss_workspace(classes) && classes >= 1 && classes <= 16,
               "seg_loss_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    float* coef = workspace + (size_t)kSegBlocks * classes * kSegSums;
    if (target_is_f32)
        hipLaunchKernelGGL(seg_partial_kernel<float>, dim3(kSegBlocks), dim3(256), 0, s, logits, (const float*)targets, kind, n, classes, hw, workspace);
    else
        hipLaunchKernelGGL(seg_partial_kernel<uint8_t>, dim3(kSegBlocks), dim3(256), 0, s, logits, (const uint8_t*)targets, kind, n, classes, hw, workspace);
    VS_LAUNCH_CHECK();
    hipLaunchKernelGGL(seg_finalize_kernel, dim3(1), dim3(64), 0, s, workspace, kSegBlocks, kind, classes, (double)n * classes * (double)hw,
                       (double)n * (double)hw, alpha, beta, eps, coef, loss);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

extern "C" int vs_seg_loss_bwd(int kind, const float* logits, const void* targets, int target_is_f32, const float* grad_out, int n,
                               int classes, int64_t hw, const float* workspace, float* dlogits, void* stream) {
    VS_REQUIRE(kind >= 1 && kind <= 4 && logits && targets && workspace && dlogits, "seg_loss_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const float* coef = workspace + (size_t)kSegBlocks * classes * kSegSums;
    const int64_t total = kind == 3 ? (int64_t)n * hw : (int64_t)n * classes * hw;
    const int grid = (int)std::min<int64_t>(8192, (total + 255) / 256);
    if (target_is_f32)
        hipLaunchKernelGGL(seg_grad_kernel<float>, dim3(grid), dim3(256), 0, s, logits, (const float*)targets, coef, grad_out, kind, n, classes, hw, dlogits);
    else
        hipLaunchKernelGGL(seg_grad_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, logits, (const uint8_t*)targets, coef, grad_out, kind, n, classes, hw, dlogits);
    VS_LAUNCH_CHECK();
    return VS_OK;
}
