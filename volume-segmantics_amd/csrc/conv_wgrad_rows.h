// Weight gradient of the FULL-RESOLUTION 16-channel 3x3 layers (the last decoder block's second convolution and the
// segmentation head at 256 x 256 x 32 images: 16 couts x 9 taps x 16 cins = 9 KB of result from 134 MB of operands) as a
// streaming kernel.  The tiled kernels (conv_wgrad_bf16_kernel) pad these layers to a 32-channel cin chunk - half of a
// workgroup's waves multiply zeros - and re-stage a halo per 128-pixel tile; here a workgroup walks a contiguous range of
// output ROWS of the batch, keeps the three input rows a 3x3 window needs in a four-row LDS ring (every input row is staged
// once per workgroup and used by the three output rows around it) and the gradient row in a double buffer, and its four waves
// split a row's 32-pixel k-steps.  Operand fragments: ds_read_b64_tr_b16 on [pixel][16 channels] rows of 32 bytes, 128-byte
// groups of four pixels swapped by bit 3 of the pixel index, so the two 4-pixel blocks of a 32-lane half (8 pixels apart) sit on
// different bank halves at every tap shift.  HBM-bound by design: algorithmic bytes = x + dy, each read once (+ 2 halo rows per
// workgroup).  Each workgroup leaves one fp32 slab; launch_slab_reduce sums the slabs in a fixed order (bit-reproducible).
//
// Replaces: the weight gradients of decoder.blocks.4.conv2 / segmentation_head of loss.backward() (vol_seg_2d_trainer.py:429).
#pragma once
#include "common.h"

namespace rows {

struct RGeom {
    int rows_total;   // N * H output rows
    int per;          // rows per workgroup
    int nsplit;       // workgroups = slabs
    int H, W;
    int cout_live;    // leading couts that are real (the head's class count; the other gradient channels are zero padding): slab = [cout_live][9][cin]
};

__device__ __forceinline__ uint2 tr16(const char* p) {
    typedef __attribute__((ext_vector_type(4))) short short4v;
    const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(p));
    return __builtin_bit_cast(uint2, v);
}

__device__ __host__ __forceinline__ constexpr int swz(int pos) { return pos ^ (((pos >> 3) & 1) << 2); }

constexpr int kRing = 4;                                    // input rows in LDS
__host__ __device__ constexpr int row_bytes(int W) { return ((W + 2 + 7) / 8 * 8) * 32; }
__host__ __device__ constexpr size_t lds_bytes(int W) {
    const size_t stage = (size_t)kRing * row_bytes(W) + 2 * (size_t)W * 32;
    const size_t red = 4 * 9 * 64 * 16;                     // the four waves' accumulators meet here at the end
    return stage > red ? stage : red;
}

// W = 128 * WQ pixels per row: wave w takes the k-steps w, w + 4, .. (WQ of them)
// PLANES: the gradient operand is NOT a 16-channel NHWC tensor but fp32 NCHW planes of g.cout_live (<= 7) channels - dLoss / dlogits as
// autograd hands it to the segmentation head: the even threads gather their pixels' values from the planes (the loads stay in flight like the
// NHWC pieces) and round them to bf16 on the way into the LDS row, the odd threads write the zero halves.  No conversion launch, 17 instead of
// 67 MB read.
template <int WQ, int PF, bool PLANES = false>
__global__ __launch_bounds__(256) void conv_wgrad_rows16_kernel(WgradParams p, RGeom g) {
    constexpr int W = 128 * WQ, ROWB = row_bytes(W), DZB = W * 32;
    constexpr int kMaxPlanes = 7;
    constexpr int SEGS = W * 2 / 256;                       // 16-byte pieces of one row per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xring = smem;
    char* dzb = smem + kRing * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int wg = xcd_block(1);
    const int g0 = wg * g.per, g1 = min(g.rows_total, g0 + g.per);
    if (g0 >= g1) return;                                   // (uniform: whole workgroups only)
    const int total_bytes = g.rows_total * W * 32;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.src0, total_bytes);
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(p.dy, PLANES ? 0 : total_bytes);

    // zero halo pixels (positions 0 and W + 1) of the four ring rows: the staging below never writes them
    if (tid < kRing * 4) {
        const int r = tid >> 2, which = (tid >> 1) & 1, seg = tid & 1;
        *reinterpret_cast<uint4*>(xring + r * ROWB + swz(which ? W + 1 : 0) * 32 + seg * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    // staging: piece i of a thread = 16 bytes at byte offset (i * 256 + tid) * 16 of the row (pixel idx >> 1, channel half idx & 1)
    int xdst[SEGS], ddst[SEGS];
#pragma unroll
    for (int i = 0; i < SEGS; ++i) {
        const int idx = i * 256 + tid, px = idx >> 1, seg = idx & 1;
        xdst[i] = swz(px + 1) * 32 + seg * 16;
        ddst[i] = swz(px) * 32 + seg * 16;
    }
    auto load_x = [&](int grow, uint4 (&v)[SEGS]) {       // input row `grow` of the batch (row of ANY image; outside the batch: zeros)
        const bool ok = grow >= 0 && grow < g.rows_total;
#pragma unroll
        for (int i = 0; i < SEGS; ++i) v[i] = bload(rx, ok ? grow * (W * 32) + (i * 256 + tid) * 16 : -1, 0);
    };
    auto load_d = [&](int grow, uint4 (&v)[SEGS]) {
        const bool ok = grow < g1;
#pragma unroll
        for (int i = 0; i < SEGS; ++i) v[i] = bload(rd, ok ? grow * (W * 32) + (i * 256 + tid) * 16 : -1, 0);
    };
    auto put_x = [&](int grow, const uint4 (&v)[SEGS]) {
        char* row = xring + ((grow + kRing) % kRing) * ROWB;
#pragma unroll
        for (int i = 0; i < SEGS; ++i) *reinterpret_cast<uint4*>(row + xdst[i]) = v[i];
    };
    auto put_d = [&](int grow, const uint4 (&v)[SEGS]) {
        char* row = dzb + (grow & 1) * DZB;
#pragma unroll
        for (int i = 0; i < SEGS; ++i) *reinterpret_cast<uint4*>(row + ddst[i]) = v[i];
    };
    // PLANES: piece i of a thread is the channel half tid & 1 of pixel (i * 256 + tid) >> 1: even threads own the live halves
    const float* planes = reinterpret_cast<const float*>(p.dy);
    const int ncls = g.cout_live;
    auto load_dp = [&](int grow, float (&v)[SEGS][kMaxPlanes]) {
        const bool ok = grow < g1 && !(tid & 1);
        const int n = grow / g.H, y = grow - n * g.H;
        const float* base = planes + ((size_t)n * ncls * g.H + y) * W;
#pragma unroll
        for (int i = 0; i < SEGS; ++i)
#pragma unroll
            for (int k = 0; k < kMaxPlanes; ++k) v[i][k] = (ok && k < ncls) ? base[(size_t)k * g.H * W + ((i * 256 + tid) >> 1)] : 0.f;
    };
    auto put_dp = [&](int grow, const float (&v)[SEGS][kMaxPlanes]) {
        char* row = dzb + (grow & 1) * DZB;
#pragma unroll
        for (int i = 0; i < SEGS; ++i) {
            uint4 o;
            o.x = pack2<bf16_t>(v[i][0], v[i][1]); o.y = pack2<bf16_t>(v[i][2], v[i][3]);
            o.z = pack2<bf16_t>(v[i][4], v[i][5]); o.w = pack2<bf16_t>(v[i][6], 0.f);
            *reinterpret_cast<uint4*>(row + ddst[i]) = (tid & 1) ? make_uint4(0u, 0u, 0u, 0u) : o;
        }
    };

    // fragment addresses of k-step `wave`: pixel 32 * wave + 8 lq + (lr >> 2) + 4 h, 8 bytes at channel 4 (lr & 3)
    int da[2], xa[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int px = 32 * wave + 8 * lq + (lr >> 2) + 4 * h;
        da[h] = swz(px) * 32 + (lr & 3) * 8;
#pragma unroll
        for (int s = 0; s < 3; ++s) xa[h][s] = swz(px + s) * 32 + (lr & 3) * 8;   // position = pixel + 1, pixel = px + s - 1
    }

    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 xs[PF][SEGS], ds[PF][SEGS];                       // rows in flight: PF steps ahead of the MFMAs
    float dp[PLANES ? PF : 1][SEGS][kMaxPlanes];
    load_x(g0 - 1, xs[0]);
    load_x(g0, ds[0]);
    put_x(g0 - 1, xs[0]);
    put_x(g0, ds[0]);
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        load_x(g0 + 1 + j, xs[j]);
        if constexpr (PLANES) load_dp(g0 + j, dp[j]); else load_d(g0 + j, ds[j]);
    }
    int y = g0 % g.H;
    for (int gbase = g0; gbase < g1; gbase += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int grow = gbase + j;
            if (grow >= g1) break;                          // (uniform)
            put_x(grow + 1, xs[j]);
            if constexpr (PLANES) put_dp(grow, dp[j]); else put_d(grow, ds[j]);
            __syncthreads();
            load_x(grow + 1 + PF, xs[j]);                   // in flight while this and the next PF - 1 rows' MFMAs run
            if constexpr (PLANES) load_dp(grow + PF, dp[j]); else load_d(grow + PF, ds[j]);
            const char* drow = dzb + (grow & 1) * DZB;
#pragma unroll
            for (int q = 0; q < WQ; ++q) {
                const int koff = q * 4 * 32 * 32;           // k-step wave + 4 q: 128 pixels further
                const uint2 alo = tr16(drow + da[0] + koff), ahi = tr16(drow + da[1] + koff);
                const bf16x8 af = __builtin_bit_cast(bf16x8, make_uint4(alo.x, alo.y, ahi.x, ahi.y));
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    if ((r == 0 && y == 0) || (r == 2 && y == g.H - 1)) continue;   // the window's row lies outside the image (uniform)
                    const char* xrow = xring + ((grow + r - 1 + kRing) % kRing) * ROWB + koff;
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const uint2 blo = tr16(xrow + xa[0][s]), bhi = tr16(xrow + xa[1][s]);
                        acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, make_uint4(blo.x, blo.y, bhi.x, bhi.y)),
                                                                                acc[r * 3 + s], 0, 0, 0);
                    }
                }
            }
            y = y + 1 == g.H ? 0 : y + 1;
        }
    }

    // the four waves' partial sums (disjoint pixel ranges) meet in LDS, summed in wave order; slab layout [cout][tap][cin]
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int t = 0; t < 9; ++t) red[(wave * 9 + t) * 64 + lane] = acc[t];
    __syncthreads();
    float* out = p.partials + (size_t)wg * g.cout_live * 9 * 16;
    for (int item = tid; item < 9 * 64; item += 256) {
        const int t = item >> 6, l = item & 63;
        f32x4 v = red[(0 * 9 + t) * 64 + l];
#pragma unroll
        for (int w = 1; w < 4; ++w) v += red[(w * 9 + t) * 64 + l];
        const int ci = l & 15, cq = l >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (cq * 4 + r < g.cout_live) out[((cq * 4 + r) * 9 + t) * 16 + ci] = v[r];
    }
}

// The same walk for the layer in front: 32 channels at HALF resolution, nearest-upsampled by the forward convolution's loader
// (the last decoder block's first convolution: up(32 @ 128 x 128) -> 16 @ 256 x 256).  A step is one SOURCE row = two output
// rows; the ring holds source rows, un-duplicated (positions of 64 bytes, the two 16-channel slices swapped by bit 2 of the
// position), and a lane's transposed read simply names the source pixel of each of its four upsampled pixels - two of them share
// an address.  The window rows of the two output rows fall on three source rows (j - 1: tap row 0 of the even output row; j: tap
// rows 1, 2 of the even and 0, 1 of the odd one; j + 1: tap row 2 of the odd one), so a source fragment feeds up to four MFMAs.
__host__ __device__ constexpr int up_row_bytes(int W) { return ((W / 2 + 2 + 7) / 8 * 8) * 64; }
__host__ __device__ constexpr size_t up_lds_bytes(int W) {
    const size_t stage = (size_t)kRing * up_row_bytes(W) + 4 * (size_t)W * 32;
    const size_t red = 4 * 18 * 64 * 16;
    return stage > red ? stage : red;
}
__device__ __host__ __forceinline__ constexpr int up_off(int pos, int nb) { return pos * 64 + (((nb ^ (pos >> 2)) & 1) << 5); }

template <int WQ, int PF>
__global__ __launch_bounds__(256) void conv_wgrad_rows_up32_kernel(WgradParams p, RGeom g) {
    constexpr int W = 128 * WQ, WS = W / 2, ROWB = up_row_bytes(W), DZB = W * 32;
    constexpr int XSEGS = WS * 4 / 256, DSEGS = W * 2 / 256;   // 16-byte pieces per thread: one source row, ONE gradient row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xring = smem;
    char* dzb = smem + kRing * ROWB;                        // [step parity][output row parity][W][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int wg = xcd_block(1);
    const int HS = g.H >> 1, steps_total = g.rows_total >> 1;   // source rows of the batch
    const int g0 = wg * g.per, g1 = min(steps_total, g0 + g.per);
    if (g0 >= g1) return;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.src0, steps_total * WS * 64);
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(p.dy, g.rows_total * W * 32);

    if (tid < kRing * 8) {                                  // zero halo positions 0 and WS + 1 of the ring rows
        const int r = tid >> 3, which = (tid >> 2) & 1, seg = tid & 3;
        *reinterpret_cast<uint4*>(xring + r * ROWB + up_off(which ? WS + 1 : 0, seg >> 1) + (seg & 1) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    int xdst[XSEGS], ddst[DSEGS];
#pragma unroll
    for (int i = 0; i < XSEGS; ++i) {
        const int idx = i * 256 + tid, px = idx >> 2, seg = idx & 3;
        xdst[i] = up_off(px + 1, seg >> 1) + (seg & 1) * 16;
    }
#pragma unroll
    for (int i = 0; i < DSEGS; ++i) {
        const int idx = i * 256 + tid, px = idx >> 1, seg = idx & 1;
        ddst[i] = swz(px) * 32 + seg * 16;
    }
    auto load_x = [&](int srow, uint4 (&v)[XSEGS]) {
        const bool ok = srow >= 0 && srow < steps_total;
#pragma unroll
        for (int i = 0; i < XSEGS; ++i) v[i] = bload(rx, ok ? srow * (WS * 64) + (i * 256 + tid) * 16 : -1, 0);
    };
    auto load_d = [&](int srow, uint4 (&v)[2][DSEGS]) {    // the two gradient rows of source row `srow`: contiguous in the batch
        const bool ok = srow < g1;
#pragma unroll
        for (int yy = 0; yy < 2; ++yy)
#pragma unroll
            for (int i = 0; i < DSEGS; ++i) v[yy][i] = bload(rd, ok ? (2 * srow + yy) * (W * 32) + (i * 256 + tid) * 16 : -1, 0);
    };
    auto put_x = [&](int srow, const uint4 (&v)[XSEGS]) {
        char* row = xring + ((srow + kRing) % kRing) * ROWB;
#pragma unroll
        for (int i = 0; i < XSEGS; ++i) *reinterpret_cast<uint4*>(row + xdst[i]) = v[i];
    };
    auto put_d = [&](int srow, const uint4 (&v)[2][DSEGS]) {
        char* base = dzb + (srow & 1) * 2 * DZB;
#pragma unroll
        for (int yy = 0; yy < 2; ++yy)
#pragma unroll
            for (int i = 0; i < DSEGS; ++i) *reinterpret_cast<uint4*>(base + yy * DZB + ddst[i]) = v[yy][i];
    };

    // fragment addresses of k-step `wave` (32 OUTPUT pixels): output pixel px = 32 wave + 8 lq + (lr >> 2) + 4 h; its window column
    // s reads upsampled pixel px + s - 1 = source pixel (px + s - 1) >> 1 = position ((px + s + 1) >> 1)
    int da[2], xa[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int px = 32 * wave + 8 * lq + (lr >> 2) + 4 * h;
        da[h] = swz(px) * 32 + (lr & 3) * 8;
#pragma unroll
        for (int s = 0; s < 3; ++s) xa[h][s] = up_off((px + s + 1) >> 1, 0) + (lr & 3) * 8;   // slice nb: ^ (nb << 5)
    }

    f32x4 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 xs[PF][XSEGS], ds[PF][2][DSEGS];
    {
        uint4 t0[XSEGS], t1[XSEGS];
        load_x(g0 - 1, t0);
        load_x(g0, t1);
        put_x(g0 - 1, t0);
        put_x(g0, t1);
    }
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        load_x(g0 + 1 + j, xs[j]);
        load_d(g0 + j, ds[j]);
    }
    int y = g0 % HS;
    for (int gbase = g0; gbase < g1; gbase += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int srow = gbase + j;
            if (srow >= g1) break;                          // (uniform)
            put_x(srow + 1, xs[j]);
            put_d(srow, ds[j]);
            __syncthreads();
            load_x(srow + 1 + PF, xs[j]);
            load_d(srow + PF, ds[j]);
            const char* dbase = dzb + (srow & 1) * 2 * DZB;
            const bool top = y == 0, bot = y == HS - 1;     // the image's first / last source row (uniform)
#pragma unroll
            for (int q = 0; q < WQ; ++q) {
                const int koff = q * 4 * 32 * 32;           // gradient rows: 128 output pixels further
                const int xoff = q * 64 * 64;               // source rows: 64 positions further (the swizzle keys are unchanged)
                bf16x8 af[2];
#pragma unroll
                for (int yy = 0; yy < 2; ++yy) {
                    const uint2 lo = tr16(dbase + yy * DZB + da[0] + koff), hi = tr16(dbase + yy * DZB + da[1] + koff);
                    af[yy] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                }
#pragma unroll
                for (int sr = 0; sr < 3; ++sr) {            // source row srow - 1 + sr
                    if ((sr == 0 && top) || (sr == 2 && bot)) continue;
                    const char* xrow = xring + ((srow + sr - 1 + kRing) % kRing) * ROWB + xoff;
#pragma unroll
                    for (int s = 0; s < 3; ++s)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            const uint2 lo = tr16(xrow + (xa[0][s] ^ (nb << 5))), hi = tr16(xrow + (xa[1][s] ^ (nb << 5)));
                            const bf16x8 bf = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                            // even output row 2 srow: window rows 0 / 1 / 2 = source rows srow - 1 / srow / srow; odd one: srow / srow / srow + 1
                            if (sr == 0) acc[0 * 3 + s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf, acc[0 * 3 + s][nb], 0, 0, 0);
                            if (sr == 1) {
                                acc[1 * 3 + s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf, acc[1 * 3 + s][nb], 0, 0, 0);
                                acc[2 * 3 + s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf, acc[2 * 3 + s][nb], 0, 0, 0);
                                acc[0 * 3 + s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf, acc[0 * 3 + s][nb], 0, 0, 0);
                                acc[1 * 3 + s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf, acc[1 * 3 + s][nb], 0, 0, 0);
                            }
                            if (sr == 2) acc[2 * 3 + s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf, acc[2 * 3 + s][nb], 0, 0, 0);
                        }
                }
            }
            y = y + 1 == HS ? 0 : y + 1;
        }
    }

    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) red[((wave * 9 + t) * 2 + nb) * 64 + lane] = acc[t][nb];
    __syncthreads();
    float* out = p.partials + (size_t)wg * 16 * 9 * 32;
    for (int item = tid; item < 18 * 64; item += 256) {
        const int tn = item >> 6, l = item & 63, t = tn >> 1, nb = tn & 1;
        f32x4 v = red[(0 * 18 + tn) * 64 + l];
#pragma unroll
        for (int w = 1; w < 4; ++w) v += red[(w * 18 + tn) * 64 + l];
        const int ci = nb * 16 + (l & 15), cq = l >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[((cq * 4 + r) * 9 + t) * 32 + ci] = v[r];
    }
}

}  // namespace rows
