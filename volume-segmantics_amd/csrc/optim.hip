// Parameter-side kernels (gfx950): fused AdamW over the flat fp32 parameter buffer, derivation of
// the low-precision / flipped-transposed weight copies the conv kernels read, and the small layout
// transforms around the segmentation head's gradient.
#include <algorithm>

#include "common.h"

namespace {

// torch.optim.AdamW (single tensor path): p *= 1 - lr*wd; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
struct AdamwScalars { float lr, b1, b2, eps, wd, bc1, bc2_sqrt; };
__device__ __forceinline__ AdamwScalars adamw_scalars(float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                      const float* __restrict__ hyper) {
    // hyper (optional): the step's scalars in device memory, written by vs_train_hyper_set - a captured graph replays the launch with a
    // new learning rate / beta1 / bias correction every step
    if (hyper) return AdamwScalars{hyper[0], hyper[1], hyper[2], hyper[3], hyper[4], hyper[5], hyper[6]};
    return AdamwScalars{lr, b1, b2, eps, wd, bc1, bc2_sqrt};
}
// ONE definition of the element update for every kernel that applies it (flat, ranged, fused with the weight copies): the same
// expression tree, hence the same fused multiply-adds, hence bit-identical parameters whichever launch carried the step
__device__ __forceinline__ float adamw_element(float pi, float gi, float& m, float& v, const AdamwScalars& a) {
    pi = pi * (1.f - a.lr * a.wd);
    const float mi = a.b1 * m + (1.f - a.b1) * gi;
    const float vi = a.b2 * v + (1.f - a.b2) * gi * gi;
    m = mi;
    v = vi;
    const float denom = sqrtf(vi) / a.bc2_sqrt + a.eps;
    pi -= (a.lr / a.bc1) * (mi / denom);
    return pi;
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, const uint8_t* __restrict__ mask, int64_t n, float lr, float b1,
                             float b2, float eps, float wd, float bc1, float bc2_sqrt, const float* __restrict__ hyper) {
    const AdamwScalars a = adamw_scalars(lr, b1, b2, eps, wd, bc1, bc2_sqrt, hyper);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (mask && !mask[i]) continue;
        float mi = m[i], vi = v[i];
        p[i] = adamw_element(p[i], g[i], mi, vi, a);
        m[i] = mi;
        v[i] = vi;
    }
}

__global__ void adamw_ranges_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                    float* __restrict__ v, AdamwRanges r, float lr, float b1, float b2, float eps, float wd,
                                    float bc1, float bc2_sqrt, const float* __restrict__ hyper) {
    const AdamwScalars a = adamw_scalars(lr, b1, b2, eps, wd, bc1, bc2_sqrt, hyper);
    const long off = r.off[blockIdx.y];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < r.len[blockIdx.y]; i += (long)gridDim.x * blockDim.x) {
        const long k = off + i;
        float mi = m[k], vi = v[k];
        p[k] = adamw_element(p[k], g[k], mi, vi, a);
        m[k] = mi;
        v[k] = vi;
    }
}

// What the fused optimiser launch does to ONE weight on its way into the copies (weight_prepare_tile's `upd`): the AdamW element
// update with the gradient from the flat buffer.  widx = element offset inside the layer's weight tensor; pointers offset to the layer.
struct NoUpdate {
    __device__ __forceinline__ float operator()(long, float v) const { return v; }
};
struct AdamwUpdate {
    float* p; const float* g; float* m; float* v;    // the layer's slices of the flat parameter / gradient / moment buffers
    AdamwScalars a;
    __device__ __forceinline__ float operator()(long widx, float w) const {
        float mi = m[widx], vi = v[widx];
        const float pn = adamw_element(w, g[widx], mi, vi, a);
        m[widx] = mi;
        v[widx] = vi;
        p[widx] = pn;
        return pn;
    }
};

// One 32 x 32 (cout x cin) tile of one tap: w fp32 [cout][taps][cin] -> wc (T, same layout, optional) and wt (T, [cin][taps
// flipped][cout_pad], optional).  cg > 0: grouped convolution with cg channels per group (cin == cout) - w is
// [cout][taps][cg]; the tile is the super-group bx's diagonal 32 x 32 block, wc / wt rows are 32 long, and a weight lands in
// its group's cg x cg sub-block (zeros elsewhere).  Threads: tx 0..31, ty 0..7.
template <typename T, typename Upd>
__device__ __forceinline__ void weight_prepare_tile(const float* __restrict__ w, T* __restrict__ wc, T* __restrict__ wt, int cout, int taps,
                                                    int cin, int cout_pad, int cg, int bx, int by, int tap, int tx, int ty,
                                                    float (&tile)[32][33], const Upd& upd) {
    const bool two_groups = cg == 255;      // radix-2 split-attention convolution (ResNeSt): cin -> cout = 2 cin in TWO groups; w is
    if (two_groups) cg = 0;                 // [cout][taps][cin / 2], the copies are dense [cout][taps][cin] with the other group's half zero
    const int ci0 = bx * 32, co0 = (cg ? bx : by) * 32;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        float v = 0.f;
        if (two_groups) {
            if (co < cout && ci < cin) {
                const int half = cin / 2, cl = ci - (co / (cout / 2)) * half;
                if (cl >= 0 && cl < half) { const long widx = ((long)co * taps + tap) * half + cl; v = upd(widx, w[widx]); }
                if (wc) Elem<T>::st(wc + ((size_t)co * taps + tap) * cin + ci, v);
            }
        } else if (cg) {
            if (co / cg == ci / cg) { const long widx = ((long)co * taps + tap) * cg + ci % cg; v = upd(widx, w[widx]); }
            if (wc) Elem<T>::st(wc + ((size_t)co * taps + tap) * 32 + tx, v);
        } else if (co < cout && ci < cin) {
            const long widx = ((long)co * taps + tap) * cin + ci;
            v = upd(widx, w[widx]);
            if (wc) Elem<T>::st(wc + ((size_t)co * taps + tap) * cin + ci, v);
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    if (wt) {
        for (int r = ty; r < 32; r += 8) {
            const int ci = ci0 + r, co = co0 + tx;
            if (cg) Elem<T>::st(wt + ((size_t)ci * taps + (taps - 1 - tap)) * 32 + tx, tile[tx][r]);
            else if (ci < cin && co < cout_pad)
                Elem<T>::st(wt + ((size_t)ci * taps + (taps - 1 - tap)) * cout_pad + co, tile[tx][r]);
        }
    }
}

template <typename T>
__global__ void weight_prepare_kernel(const float* __restrict__ w, T* __restrict__ wc, T* __restrict__ wt, int cout,
                                      int taps, int cin, int cout_pad, int cg) {
    __shared__ float tile[32][33];
    weight_prepare_tile<T>(w, wc, wt, cout, taps, cin, cout_pad, cg, blockIdx.x, blockIdx.y, blockIdx.z, threadIdx.x, threadIdx.y, tile, NoUpdate{});
}

// ConvTranspose2d(kernel 4, stride 2, padding 1) as a 3x3 convolution onto 4 * cout channels + pixel shuffle: output parity
// (a, b) of the transposed convolution reads the 2 x 2 taps kh = a + 3 - 2r, kw = b + 3 - 2q (r, q = 3x3 tap row / column; the
// other five taps are zero).  w fp32 [cin][cout][4][4] (torch's layout) -> wc [(2a+b) * cout + co][r * 3 + q][ci] and its
// flipped / transposed twin wt [ci][8 - (r * 3 + q)][(2a+b) * cout + co] for the data gradient.
template <typename T>
__global__ void convt_weight_prepare_kernel(const float* __restrict__ w, T* __restrict__ wc, T* __restrict__ wt, int cin, int cout) {
    const int total = 4 * cout * 9 * cin;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ci = i % cin, tap = (i / cin) % 9, row = i / (9 * cin);
        const int sub = row / cout, co = row - sub * cout;
        const int kh = (sub >> 1) + 3 - 2 * (tap / 3), kw = (sub & 1) + 3 - 2 * (tap % 3);
        const float v = (kh >= 0 && kh < 4 && kw >= 0 && kw < 4) ? w[(((size_t)ci * cout + co) * 4 + kh) * 4 + kw] : 0.f;
        if (wc) Elem<T>::st(wc + i, v);
        if (wt) Elem<T>::st(wt + ((size_t)ci * 9 + (8 - tap)) * (4 * cout) + row, v);
    }
}

// the weight gradient of that 3x3 convolution, dense[4 * cout][9][cin] fp32 -> dw[cin][cout][4][4]: every real tap once
__global__ void convt_wgrad_gather_kernel(const float* __restrict__ dense, float* __restrict__ dw, int cin, int cout) {
    const int total = cin * cout * 16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int kw = i & 3, kh = (i >> 2) & 3, co = (i >> 4) % cout, ci = (i >> 4) / cout;
        const int a = (kh + 1) & 1, b = (kw + 1) & 1;
        const int r = (a + 3 - kh) >> 1, q = (b + 3 - kw) >> 1;
        dw[i] = dense[((size_t)((a * 2 + b) * cout + co) * 9 + r * 3 + q) * cin + ci];
    }
}

// weight gradient of a two-group convolution: dense[cout][taps][cin] -> dw[cout][taps][cin / 2], every output channel's own group half
__global__ void two_group_wgrad_extract_kernel(const float* __restrict__ dense, float* __restrict__ dw, int cout, int taps, int cin) {
    const int half = cin / 2, total = cout * taps * half;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int cl = i % half, row = i / half, co = row / taps;
        dw[i] = dense[(size_t)row * cin + (co / (cout / 2)) * half + cl];
    }
}

// all layers in ONE launch: the descriptor table travels in the kernel arguments; a block finds its layer by a linear
// scan over the (<= 64) cumulative block counts
struct PrepTable {
    int n;
    int first_block[65];
    long w_off[64], wc_off[64], wt_off[64];  // element offset into params; BYTE offsets into the workspace (-1 = none)
    short cout[64], cin[64], cout_pad[64];
    unsigned char taps[64], cg[64];          // cg: channels per group of a grouped convolution (0 = dense)
    unsigned char update[64];                // fused optimiser step (weight_prepare_all_kernel<T, true>): AdamW runs on this layer (0 = frozen: copies only)
};

template <typename T, bool FUSED>
__global__ void weight_prepare_all_kernel(float* __restrict__ params, char* __restrict__ ws, PrepTable t, const float* __restrict__ grads,
                                          float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq, float lr, float b1, float b2, float eps,
                                          float wd, float bc1, float bc2_sqrt, const float* __restrict__ hyper) {
    __shared__ float tile[32][33];
    // (the grid may be capped below the tile count - launch_adamw_prepare_all: a training step's optimiser launches share the chip with
    // the backward pass and need not finish fast - so a block walks tiles blockIdx.x, + gridDim.x, ..)
    int l = 0;
    for (int tb = blockIdx.x; tb < t.first_block[t.n]; tb += gridDim.x) {
        while (l + 1 < t.n && tb >= t.first_block[l + 1]) ++l;
        const int cout = t.cout[l], cin = t.cin[l], cout_pad = t.cout_pad[l], taps = t.taps[l], cg = t.cg[l];
        const int cib = (cin + 31) / 32, cob = (cg && cg != 255) ? 1 : ((cout_pad > cout ? cout_pad : cout) + 31) / 32;
        int b = tb - t.first_block[l];
        const int bx = b % cib; b /= cib;
        const int by = b % cob;
        const int tap = b / cob;
        float* w = params + t.w_off[l];
        T* wc = t.wc_off[l] >= 0 ? reinterpret_cast<T*>(ws + t.wc_off[l]) : nullptr;
        T* wt = t.wt_off[l] >= 0 ? reinterpret_cast<T*>(ws + t.wt_off[l]) : nullptr;
        if (tb != (int)blockIdx.x) __syncthreads();      // the previous tile's transposed reads are done
        bool done = false;
        if constexpr (FUSED) {
            if (t.update[l]) {
                AdamwUpdate u{w, grads + t.w_off[l], exp_avg + t.w_off[l], exp_avg_sq + t.w_off[l], adamw_scalars(lr, b1, b2, eps, wd, bc1, bc2_sqrt, hyper)};
                weight_prepare_tile<T>(w, wc, wt, cout, taps, cin, cout_pad, cg, bx, by, tap, threadIdx.x & 31, threadIdx.x >> 5, tile, u);
                done = true;
            }
        }
        if (!done) weight_prepare_tile<T>(w, wc, wt, cout, taps, cin, cout_pad, cg, bx, by, tap, threadIdx.x & 31, threadIdx.x >> 5, tile, NoUpdate{});
    }
}

// dlogits (n, K, h, w) fp32 NCHW -> [n*h*w][16] T (zero padded channels); the same sweep leaves per-class partial sums
// (bias gradient): partial[block][k], finished by bias_grad_final in block order
template <typename T>
__global__ __launch_bounds__(256) void dlogits_to_nhwc16_kernel(const float* __restrict__ d, T* __restrict__ o, int n, int k, int64_t hw,
                                                              float* __restrict__ partial) {
    __shared__ float red[4][16];
    const int64_t total = (int64_t)n * hw;
    float sum[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) sum[c] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / hw, px = i % hw;
        float v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) { v[c] = c < k ? d[((size_t)b * k + c) * hw + px] : 0.f; sum[c] += v[c]; }
        if (o) {                                           // (null: the bias gradient only)
            st8(o + (size_t)i * 16, v);
            st8(o + (size_t)i * 16 + 8, v + 8);
        }
    }
    if (!partial) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float s = wave_sum(sum[c]);
        if (lane == 0) red[wave][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < k) partial[(size_t)blockIdx.x * k + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// per-class sum of fp32 NCHW planes d (n, k, hw) without the NHWC copy: block (b, s, c) sums segment s of plane (b, c), 16 bytes per lane;
// partial[(b * S + s) * k + c], finished by bias_grad_final in block order
constexpr int kPlaneSegs = 4;
__global__ __launch_bounds__(256) void planes_sum_kernel(const float* __restrict__ d, int k, int64_t hw, float* __restrict__ partial) {
    __shared__ float red[4];
    const int c = blockIdx.x % k, bs = blockIdx.x / k, s = bs % kPlaneSegs, b = bs / kPlaneSegs;
    const float* p = d + ((size_t)b * k + c) * hw;
    const int64_t per = ((hw + kPlaneSegs - 1) / kPlaneSegs + 3) & ~(int64_t)3, i0 = s * per, i1 = i0 + per < hw ? i0 + per : hw;
    float sum = 0.f;
    if ((((uintptr_t)p) & 15) == 0) {
        int64_t i = i0 + threadIdx.x * 4;
        for (; i + 3 < i1; i += 1024) { const float4 v = *reinterpret_cast<const float4*>(p + i); sum += (v.x + v.y) + (v.z + v.w); }
        for (; i < i1; ++i) sum += p[i];                       // (a ragged tail: at most 3 elements of one thread)
    } else {
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) sum += p[i];
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)bs * k + c] = (red[0] + red[1]) + (red[2] + red[3]);
}

// per-class sum of dlogits: stage 2 = one wave per class over the sweep's block partials (fixed order)
__global__ __launch_bounds__(64) void bias_grad_final(const float* __restrict__ partial, float* __restrict__ db, int nblk, int k) {
    const int c = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 64) s += partial[(size_t)b * k + c];
    s = wave_sum(s);
    if (threadIdx.x == 0) db[c] = s;
}

template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ a, T* __restrict__ b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        Elem<T>::st(b + i, a[i]);
}

inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int vs_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* mask,
                             int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                             void* stream) {
    VS_REQUIRE(params && grads && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "adamw_step: bad arguments");
    if (n == 0) return VS_OK;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, mask, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, (const float*)nullptr);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

// one slice [off, off + n) of the flat buffers
int launch_adamw_slice(const vs_adamw_args& a, const float* grads, int64_t off, int64_t n, hipStream_t s) {
    if (n <= 0) return VS_OK;
    const float bc1 = 1.f - powf(a.beta1, (float)a.step);
    const float bc2s = sqrtf(1.f - powf(a.beta2, (float)a.step));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, s, a.params + off, grads + off, a.exp_avg + off,
                       a.exp_avg_sq + off, (const uint8_t*)nullptr, n, a.lr, a.beta1, a.beta2, a.eps, a.weight_decay, bc1, bc2s, a.hyper);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

// several slices in one launch: grid.y = slice, grid.x blocks stride over it
int launch_adamw_ranges(const vs_adamw_args& a, const float* grads, const AdamwRanges& r, hipStream_t s) {
    if (r.n <= 0) return VS_OK;
    long longest = 0;
    for (int i = 0; i < r.n; ++i) longest = std::max(longest, r.len[i]);
    const float bc1 = 1.f - powf(a.beta1, (float)a.step);
    const float bc2s = sqrtf(1.f - powf(a.beta2, (float)a.step));
    hipLaunchKernelGGL(adamw_ranges_kernel, dim3((unsigned)std::min<long>(1024, (longest + 1023) / 1024), (unsigned)r.n), dim3(256), 0, s, a.params, grads, a.exp_avg, a.exp_avg_sq, r, a.lr, a.beta1,
                       a.beta2, a.eps, a.weight_decay, bc1, bc2s, a.hyper);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

// ---- per-step scalars of a captured training step ----------------------------------------------------------------------
__global__ void train_hyper_kernel(float* hyper, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2s,
                                   float step, int64_t* nbt, int n_nbt) {
    if (threadIdx.x == 0) {
        hyper[0] = lr; hyper[1] = b1; hyper[2] = b2; hyper[3] = eps; hyper[4] = wd; hyper[5] = bc1; hyper[6] = bc2s; hyper[7] = step;
    }
    for (int i = threadIdx.x; i < n_nbt; i += blockDim.x) nbt[i] += 1;   // BatchNorm num_batches_tracked
}

extern "C" int vs_train_hyper_set(float* hyper, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                  int64_t* num_batches_tracked, int n_bn, void* stream) {
    VS_REQUIRE(hyper && step >= 1 && (n_bn == 0 || num_batches_tracked), "train_hyper_set: bad arguments");
    const float bc1 = 1.f - powf(beta1, (float)step);          // the same host arithmetic as the scalar-argument path:
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));  // a replayed step updates the parameters bit-identically
    hipLaunchKernelGGL(train_hyper_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hyper, lr, beta1, beta2, eps, weight_decay,
                       bc1, bc2s, (float)step, num_batches_tracked, n_bn);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

int launch_weight_prepare(int dtype, const float* w, void* wc, void* wt, int cout, int taps, int cin, int cout_pad,
                          hipStream_t s) {
    dim3 grid(cdiv(cin, 32), cdiv(cout_pad > cout ? cout_pad : cout, 32), taps);
    VS_FOR_T(dtype, hipLaunchKernelGGL(weight_prepare_kernel<T>, grid, dim3(32, 8), 0, s, w, (T*)wc, (T*)wt, cout, taps,
                           cin, cout_pad, 0));
    VS_LAUNCH_CHECK();
    return VS_OK;
}

int launch_convt_weight_prepare(int dtype, const float* w, void* wc, void* wt, int cin, int cout, hipStream_t s) {
    const int blocks = grid_for((int64_t)4 * cout * 9 * cin);
    VS_FOR_T(dtype, hipLaunchKernelGGL(convt_weight_prepare_kernel<T>, dim3(blocks), dim3(256), 0, s, w, (T*)wc, (T*)wt, cin, cout));
    VS_LAUNCH_CHECK();
    return VS_OK;
}
int launch_two_group_wgrad_extract(const float* dense, float* dw, int cout, int taps, int cin, hipStream_t s) {
    hipLaunchKernelGGL(two_group_wgrad_extract_kernel, dim3(grid_for((int64_t)cout * taps * (cin / 2))), dim3(256), 0, s, dense, dw, cout, taps, cin);
    VS_LAUNCH_CHECK();
    return VS_OK;
}
int launch_convt_wgrad_gather(const float* dense, float* dw, int cin, int cout, hipStream_t s) {
    hipLaunchKernelGGL(convt_wgrad_gather_kernel, dim3(grid_for((int64_t)cin * cout * 16)), dim3(256), 0, s, dense, dw, cin, cout);
    VS_LAUNCH_CHECK();
    return VS_OK;
}

int launch_weight_prepare_grouped(int dtype, const float* w, void* wc, void* wt, int cout, int taps, int cg, hipStream_t s) {
    VS_REQUIRE(cout % 32 == 0 && cg >= 4 && cg <= 32 && 32 % cg == 0, "weight_prepare_grouped: %d channels in groups of %d", cout, cg);
    dim3 grid(cout / 32, 1, taps);
    VS_FOR_T(dtype, hipLaunchKernelGGL(weight_prepare_kernel<T>, grid, dim3(32, 8), 0, s, w, (T*)wc, (T*)wt, cout, taps, cout, cout, cg));
    VS_LAUNCH_CHECK();
    return VS_OK;
}

static int fill_prep_table(PrepTable& t, int n, const long* w_off, const long* wc_off, const long* wt_off, const int* cout, const int* taps,
                           const int* cin, const int* cout_pad, const int* cg) {
    VS_REQUIRE(n <= 64, "weight_prepare_all: too many layers (%d)", n);
    t.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        t.first_block[i] = blocks;
        t.w_off[i] = w_off[i]; t.wc_off[i] = wc_off[i]; t.wt_off[i] = wt_off[i];
        t.cout[i] = (short)cout[i]; t.cin[i] = (short)cin[i]; t.cout_pad[i] = (short)cout_pad[i]; t.taps[i] = (unsigned char)taps[i];
        t.cg[i] = (unsigned char)(cg ? cg[i] : 0);
        blocks += cdiv(cin[i], 32) * ((t.cg[i] && t.cg[i] != 255) ? 1 : cdiv(cout_pad[i] > cout[i] ? cout_pad[i] : cout[i], 32)) * taps[i];
    }
    t.first_block[n] = blocks;
    return VS_OK;
}

int launch_weight_prepare_all(int dtype, const float* params, void* ws, int n, const long* w_off, const long* wc_off,
                              const long* wt_off, const int* cout, const int* taps, const int* cin, const int* cout_pad,
                              const int* cg, hipStream_t s) {
    PrepTable t{};
    int rc = fill_prep_table(t, n, w_off, wc_off, wt_off, cout, taps, cin, cout_pad, cg);
    if (rc) return rc;
    VS_FOR_T(dtype, hipLaunchKernelGGL((weight_prepare_all_kernel<T, false>), dim3(t.first_block[n]), dim3(256), 0, s, const_cast<float*>(params), (char*)ws, t,
                                       (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 1.f, (const float*)nullptr));
    VS_LAUNCH_CHECK();
    return VS_OK;
}

// The optimiser step of a parameter group's convolution weights and the derivation of their copies for the next forward in ONE
// launch: per weight  AdamW element update (gradient from the flat buffer)  ->  fp32 master, low-precision copy, flipped /
// transposed copy.  update[i] == 0 (a frozen layer): copies only.  Replaces the AdamW launch over the group's slice + the copy launch.
int launch_adamw_prepare_all(int dtype, const vs_adamw_args& a, const float* grads, void* ws, int n, const long* w_off, const long* wc_off,
                             const long* wt_off, const int* cout, const int* taps, const int* cin, const int* cout_pad, const int* cg,
                             const int* update, hipStream_t s) {
    PrepTable t{};
    int rc = fill_prep_table(t, n, w_off, wc_off, wt_off, cout, taps, cin, cout_pad, cg);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) t.update[i] = (unsigned char)update[i];
    const float bc1 = 1.f - powf(a.beta1, (float)a.step);
    const float bc2s = sqrtf(1.f - powf(a.beta2, (float)a.step));
    const int blocks = t.first_block[n];      // (a capped grid - a slower optimiser that leaves the chip to the backward pass - was measured: 192
                                              // blocks 4.87 vs 4.49 ms per step: the side stream's length IS the step's)
    VS_FOR_T(dtype, hipLaunchKernelGGL((weight_prepare_all_kernel<T, true>), dim3(blocks), dim3(256), 0, s, a.params, (char*)ws, t, grads,
                                       a.exp_avg, a.exp_avg_sq, a.lr, a.beta1, a.beta2, a.eps, a.weight_decay, bc1, bc2s, a.hyper));
    VS_LAUNCH_CHECK();
    return VS_OK;
}

// db (optional): the bias gradient comes out of the same sweep; partial must hold kHeadBlocks * k floats
constexpr int kHeadBlocks = 1024;
int launch_dlogits_to_nhwc16(int dtype, const float* d, void* o, int n, int k, int64_t hw, float* db, float* partial, hipStream_t s) {
    VS_REQUIRE(k >= 1 && k <= 16, "segmentation head: classes must be <= 16 (got %d)", k);
    VS_REQUIRE(!db || partial, "segmentation head: the bias gradient needs a partial buffer");
    if (!o) {   // the bias gradient only (the head's backward reads the planes itself)
        VS_REQUIRE(db && (size_t)n * kPlaneSegs <= (size_t)kHeadBlocks, "segmentation head: bias-only sweep needs db and at most %d images", kHeadBlocks / kPlaneSegs);
        hipLaunchKernelGGL(planes_sum_kernel, dim3(n * kPlaneSegs * k), dim3(256), 0, s, d, k, hw, partial);
        VS_LAUNCH_CHECK();
        hipLaunchKernelGGL(bias_grad_final, dim3(k), dim3(64), 0, s, partial, db, n * kPlaneSegs, k);
        VS_LAUNCH_CHECK();
        return VS_OK;
    }
    const int blocks = db ? kHeadBlocks : grid_for((int64_t)n * hw);
    VS_FOR_T(dtype, hipLaunchKernelGGL(dlogits_to_nhwc16_kernel<T>, dim3(blocks), dim3(256), 0, s, d, (T*)o, n, k, hw, db ? partial : nullptr));
    VS_LAUNCH_CHECK();
    if (db) {
        hipLaunchKernelGGL(bias_grad_final, dim3(k), dim3(64), 0, s, partial, db, kHeadBlocks, k);
        VS_LAUNCH_CHECK();
    }
    return VS_OK;
}

