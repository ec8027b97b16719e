// Pieces shared by the convolution kernels of conv_igemm.hip (tile kernel, direct kernel, head kernel).
#pragma once
#include <type_traits>

#include "common.h"

template <typename T> struct CT;
template <> struct CT<bf16_t> { static constexpr int CK = 32, EPS = 8; };
template <> struct CT<f16_t> { static constexpr int CK = 32, EPS = 8; };
template <> struct CT<float> { static constexpr int CK = 16, EPS = 4; };

// value as it reads back after being stored in T
template <typename T>
__device__ __forceinline__ float4 ld4_roundtrip(const float (&v)[4]) {
    if constexpr (sizeof(T) == 2) {
        const float a[4] = {v[0], v[1], v[2], v[3]};
        T t[4];
        for (int i = 0; i < 4; ++i) Elem<T>::st(t + i, a[i]);
        return make_float4(Elem<T>::ld(t), Elem<T>::ld(t + 1), Elem<T>::ld(t + 2), Elem<T>::ld(t + 3));
    } else
        return make_float4(v[0], v[1], v[2], v[3]);
}

template <typename T>
__device__ __forceinline__ void mma16(f32x4& acc, const uint4& a, const uint4& b) {
    if constexpr (std::is_same<T, f16_t>::value) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    } else if constexpr (sizeof(T) == 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                      acc, 0, 0, 0);
    } else {
        // lane (l>>4)=q holds channels 4q..4q+3 of the chunk for both operands: step s pairs element s
        const float4 fa = __builtin_bit_cast(float4, a), fb = __builtin_bit_cast(float4, b);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, fb.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, fb.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.z, fb.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.w, fb.w, acc, 0, 0, 0);
    }
}


// Epilogue of an implicit-GEMM tile: lane (lq, lr) of wave `wave` holds, for pixel tile i and cout tile j, the 4 output
// channels n0 + 16j + 4lq .. +3 of pixel  wave*PT*16 + 16i + lr  of the TH x TW patch at (n, h0, w0).
// Optional: per-channel sum / sum-of-squares partials of the raw accumulators (row `tile` of p.stats_partial), fused
// affine + residual + ReLU, split output (dgrad through a concat), 2x2 sum-pool (dgrad through nearest x2 upsampling),
// fp32 / NCHW stores (segmentation head).  `smem` must provide 8*BN floats that no wave is reading any more.
template <typename T, int BN, int PT, int NW = 4>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, int tw_shift, int out_nchw, int n, int h0, int w0, int n0,
                                              int tile, f32x4 (&acc)[PT][BN / 16], char* smem) {
    constexpr int NJ = BN / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int TW = 1 << tw_shift;
    struct { int tw_shift, out_nchw; } g{tw_shift, out_nchw};
    // ---- epilogue: lane holds pixel (lr) x couts 4*lq..4*lq+3 of each 16x16 tile ----
    const bool ragged = (p.Cout & 3) != 0 || g.out_nchw;  // segmentation head only
    // (1) optional per-channel statistics of the raw accumulators (train-mode BN of the bf16 path)
    if (p.stats_partial || p.stats_bins) {
        float* red = reinterpret_cast<float*>(smem);  // [NW waves][2][BN]
        __syncthreads();                               // staged tiles are dead
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int pl = wave * (PT * 16) + i * 16 + lr;
                const bool ok = h0 + (pl >> g.tw_shift) < p.Hout && w0 + (pl & (TW - 1)) < p.Wout;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = ok ? acc[i][j][r] : 0.f;
                    s1[r] += x; s2[r] += x * x;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s1[r] += __shfl_xor(s1[r], o, 64); s2[r] += __shfl_xor(s2[r], o, 64); }
            }
            if (lr == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    red[(wave * 2 + 0) * BN + j * 16 + lq * 4 + r] = s1[r];
                    red[(wave * 2 + 1) * BN + j * 16 + lq * 4 + r] = s2[r];
                }
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int k = tid / BN, cc = tid % BN;
            if (n0 + cc < p.Cout) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) a += red[(w * 2 + k) * BN + cc];
                if (p.stats_bins)      // fixed point: the order the workgroups arrive in cannot change the total
                    atomicAdd(p.stats_bins + ((size_t)(tile & (p.stats_nb - 1)) * 2 + k) * p.Cout + n0 + cc,
                              (unsigned long long)__double2ll_rn((double)a * (k ? kStatScale2 : kStatScale1)));
                else
                    p.stats_partial[((size_t)tile * 2 + k) * p.Cout + n0 + cc] = a;
            }
        }
    }
    // (1b) dgrad that completes the gradient of a conv+BN(+ReLU) unit's activation: mask, store g, BN-backward partials
    if (p.bz) {
        float* red = reinterpret_cast<float*>(smem);  // [NW waves][2][BN]
        __syncthreads();                               // staged tiles are dead
        // all global loads of the epilogue first (residual = earlier contributions to this gradient, z, y): issued back to
        // back they cost one memory round trip; interleaved with the stores below each would wait for the previous store
        typename Raw4<T>::type rv[PT][NJ], zq[PT][NJ], yq[PT][NJ];
        size_t off[PT][NJ];
        bool live[PT][NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = n0 + j * 16 + lq * 4;
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int pl = wave * (PT * 16) + i * 16 + lr;
                const int ho = h0 + (pl >> g.tw_shift), wo = w0 + (pl & (TW - 1));
                live[i][j] = ho < p.Hout && wo < p.Wout && c < p.Cout;
                off[i][j] = (((size_t)n * p.Hout + ho) * p.Wout + wo) * p.Cout + c;
                rv[i][j] = zq[i][j] = yq[i][j] = typename Raw4<T>::type{};
                if (live[i][j]) {
                    if (p.residual) rv[i][j] = ld4raw((const T*)p.residual + off[i][j]);
                    zq[i][j] = ld4raw((const T*)p.bz + off[i][j]);
                    if (p.brelu && p.by) yq[i][j] = ld4raw((const T*)p.by + off[i][j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = n0 + j * 16 + lq * 4;
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
            float mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {0.f, 0.f, 0.f, 0.f}, ga[4] = {0.f, 0.f, 0.f, 0.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
            if (c < p.Cout) {
                const float4 m4 = *reinterpret_cast<const float4*>(p.bmean + c), i4 = *reinterpret_cast<const float4*>(p.binvstd + c);
                mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w;
                is[0] = i4.x; is[1] = i4.y; is[2] = i4.z; is[3] = i4.w;
                if (p.brelu && !p.by) {
                    const float4 g4 = *reinterpret_cast<const float4*>(p.bgamma + c), b4 = *reinterpret_cast<const float4*>(p.bbeta + c);
                    ga[0] = is[0] * g4.x; ga[1] = is[1] * g4.y; ga[2] = is[2] * g4.z; ga[3] = is[3] * g4.w;
                    be[0] = b4.x; be[1] = b4.y; be[2] = b4.z; be[3] = b4.w;
                }
            }
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                if (!live[i][j]) continue;
                const float4 r4 = unpack4(rv[i][j]), z4 = unpack4(zq[i][j]), y4 = unpack4(yq[i][j]);
                float v[4] = {acc[i][j][0] + r4.x, acc[i][j][1] + r4.y, acc[i][j][2] + r4.z, acc[i][j][3] + r4.w};
                const float zv[4] = {z4.x, z4.y, z4.z, z4.w};
                if (p.brelu) {
                    if (p.by) {
                        v[0] = y4.x > 0.f ? v[0] : 0.f; v[1] = y4.y > 0.f ? v[1] : 0.f;
                        v[2] = y4.z > 0.f ? v[2] : 0.f; v[3] = y4.w > 0.f ? v[3] : 0.f;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = ((zv[r] - mu[r]) * ga[r] + be[r]) > 0.f ? v[r] : 0.f;
                    }
                }
                st4((T*)p.out + off[i][j], make_float4(v[0], v[1], v[2], v[3]));
                const float4 gq = ld4_roundtrip<T>(v);   // the statistics see exactly what was stored
                const float gv[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) { s1[r] += gv[r]; s2[r] += gv[r] * (zv[r] - mu[r]) * is[r]; }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s1[r] += __shfl_xor(s1[r], o, 64); s2[r] += __shfl_xor(s2[r], o, 64); }
            }
            if (lr == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    red[(wave * 2 + 0) * BN + j * 16 + lq * 4 + r] = s1[r];
                    red[(wave * 2 + 1) * BN + j * 16 + lq * 4 + r] = s2[r];
                }
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int k = tid / BN, cc = tid % BN;
            if (n0 + cc < p.Cout) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) a += red[(w * 2 + k) * BN + cc];
                p.bstats_partial[((size_t)tile * 2 + k) * p.Cout + n0 + cc] = a;
            }
        }
        return;
    }
    // (2) outputs
    const int pool_c = p.pool0 ? (p.out1 ? p.split_c : p.Cout) : 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = n0 + j * 16 + lq * 4;
        if (PT >= 2 && c < pool_c) {
            // dgrad through nearest-x2 upsampling: sum the 2x2 block (rows i, i+1 of this wave; lanes lr, lr^1)
            if constexpr (PT >= 2) {
#pragma unroll
                for (int ip = 0; ip < PT / 2; ++ip) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[2 * ip][j][r] + acc[2 * ip + 1][j][r];
                        v[r] += __shfl_xor(v[r], 1, 64);
                    }
                    const int ho = h0 + wave * PT + 2 * ip, wo = w0 + lr;
                    if ((lr & 1) == 0 && ho < p.Hout && wo < p.Wout && c < p.Cout) {
                        const size_t o = (((size_t)n * (p.Hout >> 1) + (ho >> 1)) * (p.Wout >> 1) + (wo >> 1)) * pool_c + c;
                        st4((T*)p.out + o, make_float4(v[0], v[1], v[2], v[3]));
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int pl = wave * (PT * 16) + i * 16 + lr;
            const int ho = h0 + (pl >> g.tw_shift), wo = w0 + (pl & (TW - 1));
            if (ho >= p.Hout || wo >= p.Wout || c >= p.Cout) continue;
            const size_t pix = ((size_t)n * p.Hout + ho) * p.Wout + wo;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (!ragged) {
                if (p.scale) {
                    const float4 sc = *reinterpret_cast<const float4*>(p.scale + c);
                    const float4 sh = *reinterpret_cast<const float4*>(p.shift + c);
                    v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
                } else if (p.shift) {
                    const float4 sh = *reinterpret_cast<const float4*>(p.shift + c);
                    v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w;
                }
                // destination (possibly split across two tensors: dgrad through a channel concat)
                char* dst = (char*)p.out;
                int cd = c, cstride = p.Cout;
                if (p.out1) {
                    if (c >= p.split_c) { dst = (char*)p.out1; cd = c - p.split_c; cstride = p.Cout - p.split_c; }
                    else cstride = p.split_c;
                }
                const size_t o = pix * cstride + cd;
                if (p.residual && dst == (char*)p.out) {
                    const float4 rv = ld4((const T*)p.residual + o);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                }
                if (p.relu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                else if (p.relu == 2) {      // swish (evaluation-mode EfficientNet: BatchNorm folded into scale / shift, then x * sigmoid(x))
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.f + __expf(-v[r]));
                }
                if (p.out_f32) st4((float*)dst + o, make_float4(v[0], v[1], v[2], v[3]));
                else st4((T*)dst + o, make_float4(v[0], v[1], v[2], v[3]));
            } else {
                const int nv = min(4, p.Cout - c);
                for (int r = 0; r < nv; ++r) {
                    float x = v[r];
                    if (p.scale) x = x * p.scale[c + r] + p.shift[c + r];
                    else if (p.shift) x += p.shift[c + r];
                    if (p.residual) x += Elem<T>::ld((const T*)p.residual + pix * p.Cout + c + r);
                    if (p.relu == 1) x = fmaxf(x, 0.f);
                    else if (p.relu == 2) x = x / (1.f + __expf(-x));
                    if (g.out_nchw) ((float*)p.out)[(((size_t)n * p.Cout + c + r) * p.Hout + ho) * p.Wout + wo] = x;
                    else if (p.out_f32) ((float*)p.out)[pix * p.Cout + c + r] = x;
                    else Elem<T>::st((T*)p.out + pix * p.Cout + c + r, x);
                }
            }
        }
    }
}
