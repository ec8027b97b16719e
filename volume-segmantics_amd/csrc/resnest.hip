// What timm's ResNeSt blocks (models/resnest.py ResNestBottleneck, models/layers/split_attn.py SplitAttnConv2d; timm 0.4.12 behind smp's
// timm-resnest50d / timm-resnest101e encoders) add to the existing operators (gfx950):
//   * RadixSoftmax for radix 2, cardinality 1: the attention logits [n][2 C] hold the two splits' values C apart; a = softmax over the pair
//   * everything else of the block is composed from existing kernels: the radix-2 grouped 3x3 convolution as a dense convolution with
//     block-expanded weight copies (optim.hip: two_groups), the sum of the splits and its adjoint as channel-slice copies, the gated sum
//     as vs_channel_gate + that fold, avd / avg_down pools as depthwise convolutions with constant taps (effnet.hip)
#include "common.h"

namespace {

template <typename T>
__global__ void radix2_softmax_kernel(const T* __restrict__ z, T* __restrict__ a, int n, int c) {
    const int total = n * c;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / c, ch = i % c;
        const float z0 = Elem<T>::ld(z + (size_t)b * 2 * c + ch), z1 = Elem<T>::ld(z + (size_t)b * 2 * c + c + ch);
        const float a0 = 1.f / (1.f + __expf(z1 - z0));        // softmax over two values
        Elem<T>::st(a + (size_t)b * 2 * c + ch, a0);
        Elem<T>::st(a + (size_t)b * 2 * c + c + ch, 1.f - a0);
    }
}
// dz_r = a_r (da_r - sum_r' a_r' da_r')
template <typename T>
__global__ void radix2_softmax_bwd_kernel(const T* __restrict__ da, const T* __restrict__ a, T* __restrict__ dz, int n, int c) {
    const int total = n * c;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / c, ch = i % c;
        const size_t o0 = (size_t)b * 2 * c + ch, o1 = o0 + c;
        const float a0 = Elem<T>::ld(a + o0), a1 = Elem<T>::ld(a + o1), g0 = Elem<T>::ld(da + o0), g1 = Elem<T>::ld(da + o1);
        const float dot = a0 * g0 + a1 * g1;
        Elem<T>::st(dz + o0, a0 * (g0 - dot));
        Elem<T>::st(dz + o1, a1 * (g1 - dot));
    }
}

constexpr int kVec = 8;
inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
// the attention-weighted sum of the two radix splits: out[n][p][ch] = x[n][p][ch] a[n][ch] + x[n][p][c + ch] a[n][c + ch]  (x: 2 c channels)
template <typename T>
__global__ void radix2_gated_sum_kernel(const T* __restrict__ x, const T* __restrict__ a, T* __restrict__ out, int n, int64_t hw, int c) {
    const int cv = c / kVec;
    const int64_t total = (int64_t)n * hw * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % cv);
        const int64_t row = i / cv;
        const int b = (int)(row / hw);
        float x0[kVec], x1[kVec], a0[kVec], a1[kVec], o[kVec];
        ld8(x + row * 2 * c + cg * kVec, x0);
        ld8(x + row * 2 * c + c + cg * kVec, x1);
        ld8(a + (size_t)b * 2 * c + cg * kVec, a0);
        ld8(a + (size_t)b * 2 * c + c + cg * kVec, a1);
#pragma unroll
        for (int k = 0; k < kVec; ++k) o[k] = x0[k] * a0[k] + x1[k] * a1[k];
        st8(out + i * kVec, o);
    }
}
// its data gradient: dx[n][p][r c + ch] = dout[n][p][ch] a[n][r c + ch]
template <typename T>
__global__ void radix2_gated_sum_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ a, T* __restrict__ dx, int n, int64_t hw, int c) {
    const int cv = c / kVec;
    const int64_t total = (int64_t)n * hw * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % cv);
        const int64_t row = i / cv;
        const int b = (int)(row / hw);
        float g[kVec], a0[kVec], a1[kVec], d0[kVec], d1[kVec];
        ld8(dout + i * kVec, g);
        ld8(a + (size_t)b * 2 * c + cg * kVec, a0);
        ld8(a + (size_t)b * 2 * c + c + cg * kVec, a1);
#pragma unroll
        for (int k = 0; k < kVec; ++k) { d0[k] = g[k] * a0[k]; d1[k] = g[k] * a1[k]; }
        st8(dx + row * 2 * c + cg * kVec, d0);
        st8(dx + row * 2 * c + c + cg * kVec, d1);
    }
}

}  // namespace

// z, a [n][2 c]: a[b][r c + ch] = softmax over r of z[b][r c + ch] (timm's RadixSoftmax(radix 2, cardinality 1)); bwd: dz from da and a
extern "C" int vs_radix2_softmax(int dtype, const void* z, void* a, int n, int c, void* stream) {
    VS_REQUIRE(z && a && n > 0 && c > 0, "radix2_softmax: bad arguments");
    const int blocks = (n * c + 255) / 256;
    VS_FOR_T(dtype, hipLaunchKernelGGL(radix2_softmax_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T*)z, (T*)a, n, c));
    VS_LAUNCH_CHECK();
    return VS_OK;
}
extern "C" int vs_radix2_softmax_bwd(int dtype, const void* da, const void* a, void* dz, int n, int c, void* stream) {
    VS_REQUIRE(da && a && dz && n > 0 && c > 0, "radix2_softmax_bwd: bad arguments");
    const int blocks = (n * c + 255) / 256;
    VS_FOR_T(dtype, hipLaunchKernelGGL(radix2_softmax_bwd_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T*)da, (const T*)a, (T*)dz, n, c));
    VS_LAUNCH_CHECK();
    return VS_OK;
}

// SplitAttnConv2d's last step: out [n][hw][c] = sum over the two radix splits of x [n][hw][2 c] weighted by the attention a [n][2 c]; bwd: the
// data gradient dx [n][hw][2 c] from dout (the attention's gradient is vs_sample_rowsum_b: sums of x * dout with dout repeated per split)
extern "C" int vs_radix2_gated_sum(int dtype, const void* x, const void* a, void* out, int n, int64_t hw, int c, void* stream) {
    VS_REQUIRE(x && a && out && n > 0 && hw > 0 && c > 0 && c % kVec == 0, "radix2_gated_sum: channels must be a multiple of 8");
    const dim3 grid(grid_for((int64_t)n * hw * (c / kVec)));
    VS_FOR_T(dtype, hipLaunchKernelGGL(radix2_gated_sum_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)a, (T*)out, n, hw, c));
    VS_LAUNCH_CHECK();
    return VS_OK;
}
extern "C" int vs_radix2_gated_sum_bwd(int dtype, const void* dout, const void* a, void* dx, int n, int64_t hw, int c, void* stream) {
    VS_REQUIRE(dout && a && dx && n > 0 && hw > 0 && c > 0 && c % kVec == 0, "radix2_gated_sum_bwd: channels must be a multiple of 8");
    const dim3 grid(grid_for((int64_t)n * hw * (c / kVec)));
    VS_FOR_T(dtype, hipLaunchKernelGGL(radix2_gated_sum_bwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)dout, (const T*)a, (T*)dx, n, hw, c));
    VS_LAUNCH_CHECK();
    return VS_OK;
}
