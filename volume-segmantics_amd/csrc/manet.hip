// The attention operators of smp.MAnet's decoder (segmentation-models-pytorch 0.2.1, decoders/manet/decoder.py) on NHWC tensors (gfx950).
//   * PAB (position-wise attention at the deepest level): sp = softmax over ALL hw x hw entries of center^T top, out = sp bottom,
//     then smp's `reshape(b, C, h, w)` of the [b][hw][C] product WITHOUT a transpose (the memory is reinterpreted), added to x
//   * MFAB's squeeze-excitation gates: AdaptiveAvgPool2d(1) -> Conv1x1(C -> C/16) -> ReLU -> Conv1x1(-> C) -> Sigmoid on the skip and
//     on the upsampled input, summed, multiplied onto the input
// Tiny problems (hw <= 1024 positions, <= 512 channels): plain fixed-order kernels, fp32 accumulation, no atomics.
#include <algorithm>

#include "common.h"

namespace {

constexpr int kVec = 8;
inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// S[b][i][j] = sum_k A[b][i][k] * B[b][j][k]   (A, B: [b][m][K] / [b][n][K] in T; S fp32 [b][m][n])
template <typename T>
__global__ void bmm_nt_kernel(const T* __restrict__ A, const T* __restrict__ B, float* __restrict__ S, int m, int n, int K) {
    const int b = blockIdx.y;
    const int64_t total = (int64_t)m * n;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(o / n), j = (int)(o % n);
        const T* a = A + ((size_t)b * m + i) * K;
        const T* bb = B + ((size_t)b * n + j) * K;
        float acc = 0.f;
        for (int k = 0; k < K; k += kVec) {
            float x[kVec], y[kVec];
            ld8(a + k, x); ld8(bb + k, y);
#pragma unroll
            for (int q = 0; q < kVec; ++q) acc += x[q] * y[q];
        }
        S[(size_t)b * total + o] = acc;
    }
}
// in-place softmax over all `len` entries of each row b (one block per row)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ S, int64_t len) {
    __shared__ float red[256];
    float* s = S + (size_t)blockIdx.x * len;
    float mx = -3.0e38f;
    for (int64_t i = threadIdx.x; i < len; i += 256) mx = fmaxf(mx, s[i]);
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int64_t i = threadIdx.x; i < len; i += 256) { const float e = __expf(s[i] - mx); s[i] = e; sum += e; }
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const float inv = 1.f / red[0];
    for (int64_t i = threadIdx.x; i < len; i += 256) s[i] *= inv;
}
// O[b][i][c] = sum_j S[b][i][j] * V[b][j][c]   (S fp32 [b][m][n], V [b][n][C] in T, O [b][m][C] fp32)
template <typename T>
__global__ void bmm_sv_kernel(const float* __restrict__ S, const T* __restrict__ V, float* __restrict__ O, int m, int n, int C) {
    const int b = blockIdx.y, cv = C / kVec;
    const int64_t total = (int64_t)m * cv;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(o / cv), cg = (int)(o % cv);
        const float* s = S + ((size_t)b * m + i) * n;
        float acc[kVec];
#pragma unroll
        for (int q = 0; q < kVec; ++q) acc[q] = 0.f;
        for (int j = 0; j < n; ++j) {
            float v[kVec];
            ld8(V + ((size_t)b * n + j) * C + cg * kVec, v);
            const float w = s[j];
#pragma unroll
            for (int q = 0; q < kVec; ++q) acc[q] += w * v[q];
        }
        st8(O + ((size_t)b * m + i) * C + cg * kVec, acc);
    }
}
// y[b][p][c] = x[b][p][c] + flat[b][c * hw + p], flat = O[b] read as a plain [hw * C] array: smp reshapes the [b][hw][C] product to
// (b, C, h, w) without transposing.  inverse: dO[b][q] = dy[b][p][c] for q = c * hw + p (the gradient of that reinterpretation).
template <typename T>
__global__ void pab_scramble_add_kernel(const T* __restrict__ x, const float* __restrict__ O, T* __restrict__ y, int hw, int C, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int p = (int)(i / C % hw);
        const int64_t b = i / C / hw;
        Elem<T>::st(y + i, Elem<T>::ld(x + i) + O[(size_t)b * hw * C + (size_t)c * hw + p]);
    }
}
template <typename T>
__global__ void pab_unscramble_kernel(const T* __restrict__ dy, float* __restrict__ dO, int hw, int C, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int p = (int)(i / C % hw);
        const int64_t b = i / C / hw;
        dO[(size_t)b * hw * C + (size_t)c * hw + p] = Elem<T>::ld(dy + i);
    }
}
// dS[b][i][j] = sum_c dO[b][i][c] * V[b][j][c]  (fp32 x T)
template <typename T>
__global__ void bmm_dov_kernel(const float* __restrict__ dO, const T* __restrict__ V, float* __restrict__ dS, int m, int n, int C) {
    const int b = blockIdx.y;
    const int64_t total = (int64_t)m * n;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(o / n), j = (int)(o % n);
        const float* a = dO + ((size_t)b * m + i) * C;
        const T* v = V + ((size_t)b * n + j) * C;
        float acc = 0.f;
        for (int k = 0; k < C; k += kVec) {
            float y[kVec];
            ld8(v + k, y);
#pragma unroll
            for (int q = 0; q < kVec; ++q) acc += a[k + q] * y[q];
        }
        dS[(size_t)b * total + o] = acc;
    }
}
// dV[b][j][c] = sum_i S[b][i][j] * dO[b][i][c]   (written in T)
template <typename T>
__global__ void bmm_stdo_kernel(const float* __restrict__ S, const float* __restrict__ dO, T* __restrict__ dV, int m, int n, int C) {
    const int b = blockIdx.y, cv = C / kVec;
    const int64_t total = (int64_t)n * cv;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(o / cv), cg = (int)(o % cv);
        float acc[kVec];
#pragma unroll
        for (int q = 0; q < kVec; ++q) acc[q] = 0.f;
        for (int i = 0; i < m; ++i) {
            const float w = S[((size_t)b * m + i) * n + j];
            const float* d = dO + ((size_t)b * m + i) * C + cg * kVec;
#pragma unroll
            for (int q = 0; q < kVec; ++q) acc[q] += w * d[q];
        }
        st8(dV + ((size_t)b * n + j) * C + cg * kVec, acc);
    }
}
// softmax backward over each whole row b: dS <- S * (dS - sum(S * dS))
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ S, float* __restrict__ dS, int64_t len) {
    __shared__ float red[256];
    const float* s = S + (size_t)blockIdx.x * len;
    float* d = dS + (size_t)blockIdx.x * len;
    float dot = 0.f;
    for (int64_t i = threadIdx.x; i < len; i += 256) dot += s[i] * d[i];
    red[threadIdx.x] = dot;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    dot = red[0];
    for (int64_t i = threadIdx.x; i < len; i += 256) d[i] = s[i] * (d[i] - dot);
}
// dA[b][i][k] = sum_j dP[b][i][j] * B[b][j][k]  (trans = 0)   or   dB[b][j][k] = sum_i dP[b][i][j] * A[b][i][k]  (trans = 1); K small
template <typename T>
__global__ void bmm_dp_kernel(const float* __restrict__ dP, const T* __restrict__ M, T* __restrict__ out, int m, int n, int K, int trans) {
    const int b = blockIdx.y, kv = K / kVec;
    const int rows = trans ? n : m, inner = trans ? m : n;
    const int64_t total = (int64_t)rows * kv;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(o / kv), kg = (int)(o % kv);
        float acc[kVec];
#pragma unroll
        for (int q = 0; q < kVec; ++q) acc[q] = 0.f;
        for (int t = 0; t < inner; ++t) {
            const float w = trans ? dP[((size_t)b * m + t) * n + r] : dP[((size_t)b * m + r) * n + t];
            float v[kVec];
            ld8(M + ((size_t)b * inner + t) * K + kg * kVec, v);
#pragma unroll
            for (int q = 0; q < kVec; ++q) acc[q] += w * v[q];
        }
        st8(out + ((size_t)b * rows + r) * K + kg * kVec, acc);
    }
}

// ---- squeeze-excitation gate on pooled features p [n][C] (T): a = sigmoid(W2 act(W1 p + b1) + b2), W1 [R][C], W2 [C][R] fp32; act = ReLU
// (MFAB) or swish (efficientnet-pytorch).  Workgroups per (sample, 4 hidden units) / (sample, 256 outputs): every first-layer dot product is a wave's coalesced sweep + butterfly.
__device__ __forceinline__ float se_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// forward, layer 1: hid[b][r] = act-input of the hidden unit (swish: the pre-activation; ReLU: the activation) - one wave per (sample, r), the
// C products spread over its lanes (coalesced rows of W1); blockIdx = (sample, group of 4 hidden units)
template <typename T>
__global__ __launch_bounds__(256) void se_hidden_kernel(const T* __restrict__ p, const float* __restrict__ w1, const float* __restrict__ b1,
                                                      float* __restrict__ hid, int C, int R, int swish) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const T* ps = p + (size_t)b * C;
    float acc = 0.f;
#pragma unroll 8
    for (int c = lane; c < C; c += 64) acc += w1[(size_t)r * C + c] * Elem<T>::ld(ps + c);
    acc = se_wave_sum(acc) + b1[r];
    if (lane == 0) hid[(size_t)b * R + r] = swish ? acc : fmaxf(acc, 0.f);
}
// forward, layer 2: a[b][c] = sigmoid(W2[c] . act(hid[b]) + b2[c]) - one lane per output, blockIdx = (sample, 256 outputs)
template <typename T>
__global__ __launch_bounds__(256) void se_output_kernel(const float* __restrict__ hid, const float* __restrict__ w2, const float* __restrict__ b2,
                                                      T* __restrict__ a, int C, int R, int swish) {
    __shared__ float hs[128];
    const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if ((int)threadIdx.x < R) {
        const float hv = hid[(size_t)b * R + threadIdx.x];
        hs[threadIdx.x] = swish ? hv / (1.f + __expf(-hv)) : hv;
    }
    __syncthreads();
    if (c >= C) return;
    float acc = b2[c];
    const float* wr = w2 + (size_t)c * R;
#pragma unroll 8
    for (int r = 0; r < R; ++r) acc += wr[r] * hs[r];
    Elem<T>::st(a + (size_t)b * C + c, 1.f / (1.f + __expf(-acc)));
}
// backward, stage 1a: g2 = da a (1 - a) (the second layer's pre-activation gradient, kept in G2) and g1[r] = (sum_c W2[c][r] g2[c]) act'(hidden)
// (kept in G1): one wave per (sample, r); the blocks of hidden-unit group 0 also write G2
template <typename T>
__global__ __launch_bounds__(256) void se_bwd_hidden_kernel(const T* __restrict__ da, const T* __restrict__ a, const float* __restrict__ hid,
                                                          const float* __restrict__ w2, float* __restrict__ G2, float* __restrict__ G1, int C, int R, int swish) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, r = blockIdx.y * 4 + (threadIdx.x >> 6);
    const T* das = da + (size_t)b * C;
    const T* as = a + (size_t)b * C;
    if (blockIdx.y == 0) {
        for (int c = threadIdx.x; c < C; c += 256) {
            const float av = Elem<T>::ld(as + c);
            G2[(size_t)b * C + c] = Elem<T>::ld(das + c) * av * (1.f - av);
        }
    }
    if (r >= R) return;
    float acc = 0.f;
#pragma unroll 4
    for (int c = lane; c < C; c += 64) {
        const float av = Elem<T>::ld(as + c);
        acc += w2[(size_t)c * R + r] * (Elem<T>::ld(das + c) * av * (1.f - av));
    }
    acc = se_wave_sum(acc);
    if (lane == 0) {
        const float hv = hid[(size_t)b * R + r];
        float d;
        if (swish) { const float sg = 1.f / (1.f + __expf(-hv)); d = sg * (1.f + hv * (1.f - sg)); }
        else d = hv > 0.f ? 1.f : 0.f;
        G1[(size_t)b * R + r] = acc * d;
    }
}
// stage 1b: dp[b][c] = sum_r W1[r][c] g1[b][r] - one lane per output
template <typename T>
__global__ __launch_bounds__(256) void se_bwd_input_kernel(const float* __restrict__ G1, const float* __restrict__ w1, T* __restrict__ dp, int C, int R) {
    __shared__ float g1[128];
    const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if ((int)threadIdx.x < R) g1[threadIdx.x] = G1[(size_t)b * R + threadIdx.x];
    __syncthreads();
    if (c >= C) return;
    float acc = 0.f;
#pragma unroll 8
    for (int r = 0; r < R; ++r) acc += w1[(size_t)r * C + c] * g1[r];
    Elem<T>::st(dp + (size_t)b * C + c, acc);
}
// stage 2: dW1[r][c] = sum_b G1[b][r] p[b][c], dW2[c][r] = sum_b G2[b][c] act(hid[b][r]), db1 = sum_b G1, db2 = sum_b G2 - one lane per
// output, samples in order
template <typename T>
__global__ void se_gate_bwd_params_kernel(const T* __restrict__ p, const float* __restrict__ hid, const float* __restrict__ G2, const float* __restrict__ G1,
                                          float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2, int n, int C,
                                          int R, int swish) {
    const int64_t rc = (int64_t)R * C, i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rc) {
        const int r = (int)(i / C), c = (int)(i % C);
        float acc = 0.f;
#pragma unroll 8
        for (int b = 0; b < n; ++b) acc += G1[(size_t)b * R + r] * Elem<T>::ld(p + (size_t)b * C + c);
        dw1[i] = acc;
    } else if (i < 2 * rc) {
        const int64_t j = i - rc;
        const int c = (int)(j / R), r = (int)(j % R);
        float acc = 0.f;
        for (int b = 0; b < n; ++b) {
            const float hv = hid[(size_t)b * R + r];
            acc += G2[(size_t)b * C + c] * (swish ? hv / (1.f + __expf(-hv)) : hv);
        }
        dw2[j] = acc;
    } else if (i < 2 * rc + R) {
        const int r = (int)(i - 2 * rc);
        float acc = 0.f;
        for (int b = 0; b < n; ++b) acc += G1[(size_t)b * R + r];
        db1[r] = acc;
    } else if (i < 2 * rc + R + C) {
        const int c = (int)(i - 2 * rc - R);
        float acc = 0.f;
        for (int b = 0; b < n; ++b) acc += G2[(size_t)b * C + c];
        db2[c] = acc;
    }
}

// y[n][hw][c] = x[n][hw][c] * g[n][c]  (g in T);  backward: dx = dy * g,  dg[n][c] = sum_p dy * x
template <typename T>
__global__ void channel_gate_kernel(const T* __restrict__ x, const T* __restrict__ g, T* __restrict__ y, int n, int64_t hw, int c) {
    const int cv = c / kVec;
    const int64_t total = (int64_t)n * hw * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % cv);
        const int b = (int)(i / cv / hw);
        float v[kVec], m[kVec];
        ld8(x + i * kVec, v);
        ld8(g + (size_t)b * c + cg * kVec, m);
#pragma unroll
        for (int k = 0; k < kVec; ++k) v[k] *= m[k];
        st8(y + i * kVec, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void channel_dot_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dg, int64_t hw, int c) {
    __shared__ float red[256][kVec];
    const int cs = c < 256 ? c : 256, cv = cs / kVec, rl = 256 / cv, c0 = blockIdx.y * 256;
    const int cg = threadIdx.x % cv, r0 = threadIdx.x / cv;
    const size_t base = (size_t)blockIdx.x * hw * c + c0 + cg * kVec;
    float s[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) s[k] = 0.f;
    for (int64_t r = r0; r0 < rl && r < hw; r += rl) {
        float a[kVec], d[kVec];
        ld8(x + base + (size_t)r * c, a);
        ld8(dy + base + (size_t)r * c, d);
#pragma unroll
        for (int k = 0; k < kVec; ++k) s[k] += a[k] * d[k];
    }
#pragma unroll
    for (int k = 0; k < kVec; ++k) red[threadIdx.x][k] = s[k];
    __syncthreads();
    if ((int)threadIdx.x < cs) {
        const int g = threadIdx.x / kVec, k = threadIdx.x % kVec;
        float t = 0.f;
        for (int r = 0; r < rl; ++r) t += red[r * cv + g][k];
        Elem<T>::st(dg + (size_t)blockIdx.x * c + c0 + threadIdx.x, t);
    }
}

}  // namespace

#define VS_LAUNCH_T(kernel, grid, s, ...)                                                                                \
    do {                                                                                                                  \
        VS_FOR_T(dtype, { hipLaunchKernelGGL((kernel<T>), grid, dim3(256), 0, s, __VA_ARGS__); });                     \
        VS_LAUNCH_CHECK();                                                                                                \
    } while (0)

// PAB attention.  top, center: [n][hw][K]; bottom, x: [n][hw][C].  y = x + reinterpret(softmax_all(center top^T) bottom).
// sp: fp32 [n][hw][hw] (kept for the backward pass); scratch: fp32 [n][hw][C].
extern "C" int vs_pab_attention_fwd(int dtype, const void* top, const void* center, const void* bottom, const void* x, void* y, float* sp,
                                    float* scratch, int n, int hw, int K, int C, void* stream) {
    VS_REQUIRE(top && center && bottom && x && y && sp && scratch && K % kVec == 0 && C % kVec == 0 && hw >= 1, "pab_attention_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VS_LAUNCH_T(bmm_nt_kernel, dim3(grid_for((int64_t)hw * hw), n), s, (const T*)center, (const T*)top, sp, hw, hw, K);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(n), dim3(256), 0, s, sp, (int64_t)hw * hw);
    VS_LAUNCH_CHECK();
    VS_LAUNCH_T(bmm_sv_kernel, dim3(grid_for((int64_t)hw * (C / kVec)), n), s, sp, (const T*)bottom, scratch, hw, hw, C);
    VS_LAUNCH_T(pab_scramble_add_kernel, dim3(grid_for((int64_t)n * hw * C)), s, (const T*)x, scratch, (T*)y, hw, C, (int64_t)n * hw * C);
    return VS_OK;
}
// gradients of the attention term only (the identity path x -> y is the caller's): dy -> dtop, dcenter, dbottom.
// scratch: fp32 [n][hw][C] + [n][hw][hw].
extern "C" int vs_pab_attention_bwd(int dtype, const void* dy, const void* top, const void* center, const void* bottom, const float* sp, void* dtop,
                                    void* dcenter, void* dbottom, float* scratch, int n, int hw, int K, int C, void* stream) {
    VS_REQUIRE(dy && top && center && bottom && sp && dtop && dcenter && dbottom && scratch, "pab_attention_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    float* dO = scratch;
    float* dS = scratch + (size_t)n * hw * C;
    VS_LAUNCH_T(pab_unscramble_kernel, dim3(grid_for((int64_t)n * hw * C)), s, (const T*)dy, dO, hw, C, (int64_t)n * hw * C);
    VS_LAUNCH_T(bmm_dov_kernel, dim3(grid_for((int64_t)hw * hw), n), s, dO, (const T*)bottom, dS, hw, hw, C);
    VS_LAUNCH_T(bmm_stdo_kernel, dim3(grid_for((int64_t)hw * (C / kVec)), n), s, sp, dO, (T*)dbottom, hw, hw, C);
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(n), dim3(256), 0, s, sp, dS, (int64_t)hw * hw);
    VS_LAUNCH_CHECK();
    VS_LAUNCH_T(bmm_dp_kernel, dim3(grid_for((int64_t)hw * (K / kVec)), n), s, dS, (const T*)top, (T*)dcenter, hw, hw, K, 0);
    VS_LAUNCH_T(bmm_dp_kernel, dim3(grid_for((int64_t)hw * (K / kVec)), n), s, dS, (const T*)center, (T*)dtop, hw, hw, K, 1);
    return VS_OK;
}
extern "C" size_t vs_pab_scratch_bytes(int n, int hw, int C) { return ((size_t)n * hw * C + (size_t)n * hw * hw) * sizeof(float); }

// squeeze-excitation gate on pooled features (MFAB's SE_ll / SE_hl after the average pool): a = sigmoid(W2 relu(W1 p + b1) + b2);
// hid [n][R] fp32 keeps the hidden activations for the backward pass.  C <= 4096, R <= 128.  swish = 1: the hidden activation is
// x * sigmoid(x) instead of ReLU (efficientnet-pytorch's MBConvBlock: _se_reduce, swish, _se_expand, sigmoid).
extern "C" int vs_se_gate_fwd(int dtype, const void* p, const float* w1, const float* b1, const float* w2, const float* b2, void* a, float* hid,
                              int n, int C, int R, int swish, void* stream) {
    VS_REQUIRE(p && w1 && b1 && w2 && b2 && a && hid && C >= 1 && C <= 4096 && R >= 1 && R <= 128, "se_gate_fwd: C <= 4096, R <= 128");
    VS_LAUNCH_T(se_hidden_kernel, dim3(n, (R + 3) / 4), (hipStream_t)stream, (const T*)p, w1, b1, hid, C, R, swish);
    VS_LAUNCH_T(se_output_kernel, dim3(n, (C + 255) / 256), (hipStream_t)stream, hid, w2, b2, (T*)a, C, R, swish);
    return VS_OK;
}
// scratch: vs_se_gate_scratch_floats(n, C, R) floats (the two layers' pre-activation gradients, per sample)
extern "C" size_t vs_se_gate_scratch_floats(int n, int C, int R) { return (size_t)n * ((size_t)C + R); }
extern "C" int vs_se_gate_bwd(int dtype, const void* da, const void* a, const void* p, const float* hid, const float* w1, const float* w2, void* dp,
                              float* dw1, float* db1, float* dw2, float* db2, float* scratch, int n, int C, int R, int swish, void* stream) {
    VS_REQUIRE(da && a && p && hid && w1 && w2 && dp && dw1 && db1 && dw2 && db2 && scratch && n > 0 && C >= 1 && C <= 4096 && R >= 1 && R <= 128,
               "se_gate_bwd: bad arguments (C <= 4096, R <= 128)");
    float* G2 = scratch;
    float* G1 = scratch + (size_t)n * C;
    VS_LAUNCH_T(se_bwd_hidden_kernel, dim3(n, (R + 3) / 4), (hipStream_t)stream, (const T*)da, (const T*)a, hid, w2, G2, G1, C, R, swish);
    VS_LAUNCH_T(se_bwd_input_kernel, dim3(n, (C + 255) / 256), (hipStream_t)stream, G1, w1, (T*)dp, C, R);
    const int64_t total = 2 * (int64_t)R * C + R + C;
    VS_LAUNCH_T(se_gate_bwd_params_kernel, dim3((unsigned)((total + 255) / 256)), (hipStream_t)stream, (const T*)p, hid, G2, G1, dw1, db1, dw2, db2, n, C, R, swish);
    return VS_OK;
}
// y = x * g[n][c] broadcast over the hw positions; vs_channel_dot: dg[n][c] = sum over positions of x * dy
extern "C" int vs_channel_gate(int dtype, const void* x, const void* g, void* y, int n, int64_t hw, int c, void* stream) {
    VS_REQUIRE(x && g && y && c > 0 && c % kVec == 0, "channel_gate: channels must be a multiple of 8");
    VS_LAUNCH_T(channel_gate_kernel, dim3(grid_for((int64_t)n * hw * (c / kVec))), (hipStream_t)stream, (const T*)x, (const T*)g, (T*)y, n, hw, c);
    return VS_OK;
}
extern "C" int vs_channel_dot(int dtype, const void* x, const void* dy, void* dg, int n, int64_t hw, int c, void* stream) {
    const int cs = c < 256 ? c : 256;
    VS_REQUIRE(x && dy && dg && c > 0 && c % kVec == 0, "channel_dot: unsupported channel count %d", c);
    if (c % cs || 256 % (cs / kVec)) return vs_sample_rowsum(dtype, x, dy, dg, n, hw, c, 1.f, stream);   // any other multiple of 8 (csrc/effnet.hip)
    VS_LAUNCH_T(channel_dot_kernel, dim3(n, c / cs), (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dg, hw, c);
    return VS_OK;
}
