// Training augmentations on the device (gfx950; HBM-bound byte work): the reference's albumentations pipeline
// (volume_segmantics/data/augmentations.py:68-101) applied to a whole batch of uint8 slices + masks that are already resident,
// ending in the normalised fp32 network input - the 4-worker CPU feed of the reference (utilities/config.py:33) cannot keep up
// with a step of a few milliseconds.  The HOST draws every transform's coin and parameters (data/augmentations.py:
// sample_params) and hands them over as one small table; the stages below then run for all samples at once, each sample taking
// its own branch:
//   1 RandomSizedCrop        crop window -> bilinear resize to (size, size); masks nearest
//   2 VerticalFlip / RandomRotate90 / Transpose        one exact index permutation
//   3 ElasticTransform (small affine map, then a displacement field: Gaussian-smoothed uniform noise x alpha, generated here)
//     | GridDistortion (per-axis coordinate tables from the host) | OpticalDistortion (closed-form radial map)
//   4 CLAHE                  per-tile clipped histogram -> LUT (one workgroup per tile), bilinear blend of 4 tile LUTs
//   5 RandomBrightnessContrast | RandomGamma as a 256-entry LUT per sample (built on the host: exact), then
//     (v / 255 - 0.449) / 0.226 in NumPy's fp32 order (data/datasets.py:63-69)
// Sampling = cv2.remap semantics: bilinear in fp32, BORDER_REFLECT_101, result rounded to uint8 after every stage (the
// library resamples stage by stage too).  data/augmentations.py is the NumPy form of the same arithmetic (host path + tests).
#include "common.h"

namespace {

__device__ __forceinline__ int refl101(int i, int n) {
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    i %= period;
    if (i < 0) i += period;
    return i >= n ? period - i : i;
}

// bilinear sample of an (h x w) uint8 image at (sx, sy), reflect-101 outside, rounded half-to-even like np.rint
__device__ __forceinline__ uint8_t sample_bilinear(const uint8_t* img, int pitch, int h, int w, float sx, float sy) {
    const float x0f = floorf(sx), y0f = floorf(sy);
    const float fx = sx - x0f, fy = sy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int xa = refl101(x0, w), xb = refl101(x0 + 1, w), ya = refl101(y0, h), yb = refl101(y0 + 1, h);
    const float a = img[ya * pitch + xa], b = img[ya * pitch + xb], c = img[yb * pitch + xa], d = img[yb * pitch + xb];
    const float top = __fadd_rn(__fmul_rn(a, 1.f - fx), __fmul_rn(b, fx));
    const float bot = __fadd_rn(__fmul_rn(c, 1.f - fx), __fmul_rn(d, fx));
    const float v = __fadd_rn(__fmul_rn(top, 1.f - fy), __fmul_rn(bot, fy));
    return (uint8_t)fminf(fmaxf(rintf(v), 0.f), 255.f);
}
__device__ __forceinline__ uint8_t sample_nearest(const uint8_t* img, int pitch, int h, int w, float sx, float sy) {
    return img[refl101((int)rintf(sy), h) * pitch + refl101((int)rintf(sx), w)];
}

enum { ST_CROP = 0, ST_DIHEDRAL = 1, ST_AFFINE = 2, ST_DISTORT = 3 };

// one stage for every sample: dst = stage(src) where the sample takes part, a copy otherwise
template <int STAGE>
__global__ __launch_bounds__(256) void aug_stage_kernel(const uint8_t* __restrict__ src_i, const uint8_t* __restrict__ src_m,
                                                      uint8_t* __restrict__ dst_i, uint8_t* __restrict__ dst_m, int s,
                                                      const vs_aug_params* __restrict__ params, const float* __restrict__ tables,
                                                      const float* __restrict__ fields) {
    const int b = blockIdx.y;
    const vs_aug_params p = params[b];
    const size_t plane = (size_t)s * s;
    const uint8_t* si = src_i + b * plane;
    const uint8_t* sm = src_m + b * plane;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < s * s; i += gridDim.x * 256) {
        const int y = i / s, x = i - y * s;
        uint8_t vi, vm;
        if (STAGE == ST_CROP) {
            if (!p.crop) { vi = si[i]; vm = sm[i]; }
            else {   // cv2.resize of the window: centres (dst + 0.5) * scale - 0.5, clamped to the window; INTER_NEAREST: floor(dst * scale)
                const float scy = (float)p.ch / (float)s, scx = (float)p.cw / (float)s;
                const float sy = fminf(fmaxf(__fsub_rn(__fmul_rn((float)y + 0.5f, scy), 0.5f), 0.f), (float)(p.ch - 1));
                const float sx = fminf(fmaxf(__fsub_rn(__fmul_rn((float)x + 0.5f, scx), 0.5f), 0.f), (float)(p.cw - 1));
                vi = sample_bilinear(si + p.y1 * s + p.x1, s, p.ch, p.cw, sx, sy);
                const int ny = min((int)((float)y * scy), p.ch - 1), nx = min((int)((float)x * scx), p.cw - 1);
                vm = sm[(p.y1 + ny) * s + p.x1 + nx];
            }
        } else if (STAGE == ST_DIHEDRAL) {   // out = transpose?(rot90^k(flip?(in))): walk the output index back through the three
            int yy = y, xx = x;
            if (p.transpose) { const int t = yy; yy = xx; xx = t; }
            for (int k = 0; k < p.rot_k; ++k) { const int t = yy; yy = xx; xx = s - 1 - t; }     // rot90(m)[i][j] = m[j][n-1-i]
            if (p.flip_v) yy = s - 1 - yy;
            vi = si[yy * s + xx]; vm = sm[yy * s + xx];
        } else if (STAGE == ST_AFFINE) {
            if (p.distort != 1) { vi = si[i]; vm = sm[i]; }
            else {
                const float sx = __fadd_rn(__fadd_rn(__fmul_rn(p.inv_affine[0], (float)x), __fmul_rn(p.inv_affine[1], (float)y)), p.inv_affine[2]);
                const float sy = __fadd_rn(__fadd_rn(__fmul_rn(p.inv_affine[3], (float)x), __fmul_rn(p.inv_affine[4], (float)y)), p.inv_affine[5]);
                vi = sample_bilinear(si, s, s, s, sx, sy); vm = sample_nearest(sm, s, s, s, sx, sy);
            }
        } else {
            if (p.distort == 0) { vi = si[i]; vm = sm[i]; }
            else {
                float sx, sy;
                if (p.distort == 1) {          // elastic: displacement fields [b][2][s][s]
                    sx = __fadd_rn((float)x, fields[((size_t)b * 2 + 0) * plane + i]);
                    sy = __fadd_rn((float)y, fields[((size_t)b * 2 + 1) * plane + i]);
                } else if (p.distort == 2) {   // grid: coordinate tables [b][2][s]
                    sx = tables[((size_t)b * 2 + 0) * s + x];
                    sy = tables[((size_t)b * 2 + 1) * s + y];
                } else {                        // optical: x' = (x - cx) / w, f = 1 + k r^2 + k r^4, map = x' f w + cx
                    const float fs = (float)s;
                    const float xn = __fdiv_rn(__fsub_rn((float)x, p.cx), fs), yn = __fdiv_rn(__fsub_rn((float)y, p.cy), fs);
                    const float r2 = __fadd_rn(__fmul_rn(xn, xn), __fmul_rn(yn, yn));
                    const float f = __fadd_rn(__fadd_rn(1.f, __fmul_rn(p.k, r2)), __fmul_rn(__fmul_rn(p.k, r2), r2));
                    sx = __fadd_rn(__fmul_rn(__fmul_rn(xn, f), fs), p.cx);
                    sy = __fadd_rn(__fmul_rn(__fmul_rn(yn, f), fs), p.cy);
                }
                vi = sample_bilinear(si, s, s, s, sx, sy); vm = sample_nearest(sm, s, s, s, sx, sy);
            }
        }
        dst_i[b * plane + i] = vi; dst_m[b * plane + i] = vm;
    }
}

// ---- elastic displacement fields: uniform noise in [-1, 1), separable Gaussian (scipy.ndimage.gaussian_filter: radius
// int(4 sigma + 0.5), boundary 'reflect' = the edge sample repeated), scaled by alpha --------------------------------------------------
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ void aug_noise_kernel(float* __restrict__ f, int s, const vs_aug_params* __restrict__ params) {
    const int b = blockIdx.y;
    if (params[b].distort != 1) return;
    const size_t plane = (size_t)s * s;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * s * s; i += gridDim.x * 256) {
        const uint32_t h = hash32(hash32(params[b].noise_seed + 0x9e3779b9u * (uint32_t)b) ^ (uint32_t)i * 0x85ebca6bu);
        f[(size_t)b * 2 * plane + i] = __fsub_rn(__fmul_rn((float)(h >> 8), 2.f / 16777216.f), 1.f);
    }
}
__device__ __forceinline__ int refl_sym(int i, int n) {   // scipy 'reflect': (d c b a | a b c d | d c b a)
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i >= n ? period - 1 - i : i;
}
template <int AXIS>
__global__ void aug_blur_kernel(const float* __restrict__ src, float* __restrict__ dst, int s, float sigma, float scale,
                                const vs_aug_params* __restrict__ params) {
    const int b = blockIdx.y;
    if (params[b].distort != 1) return;
    const int radius = (int)(4.f * sigma + 0.5f);
    const size_t plane = (size_t)s * s;
    float norm = 0.f;
    for (int k = -radius; k <= radius; ++k) norm += expf(-0.5f * (float)(k * k) / (sigma * sigma));
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * s * s; i += gridDim.x * 256) {
        const int fld = i / (s * s), r = i - fld * s * s, y = r / s, x = r - y * s;
        const float* p = src + ((size_t)b * 2 + fld) * plane;
        float acc = 0.f;
        for (int k = -radius; k <= radius; ++k) {
            const float wgt = expf(-0.5f * (float)(k * k) / (sigma * sigma));
            acc += wgt * (AXIS == 0 ? p[refl_sym(y + k, s) * s + x] : p[y * s + refl_sym(x + k, s)]);
        }
        dst[((size_t)b * 2 + fld) * plane + r] = acc / norm * scale;
    }
}

// ---- CLAHE ---------------------------------------------------------------------------------------------------------------------------
constexpr int kTiles = 8;
__global__ __launch_bounds__(256) void aug_clahe_lut_kernel(const uint8_t* __restrict__ img, int s, const vs_aug_params* __restrict__ params,
                                                          float* __restrict__ luts) {
    const int b = blockIdx.y, tile = blockIdx.x, ty = tile / kTiles, tx = tile - ty * kTiles;
    const float clip = params[b].clahe_clip;
    if (clip == 0.f) return;
    __shared__ int hist[256];
    __shared__ int red[256];
    const int t = threadIdx.x, th = s / kTiles, area = th * th;
    hist[t] = 0;
    __syncthreads();
    const uint8_t* p = img + (size_t)b * s * s + (size_t)ty * th * s + tx * th;
    for (int i = t; i < area; i += 256) atomicAdd(&hist[p[(i / th) * s + (i % th)]], 1);
    __syncthreads();
    const int limit = params[b].clahe_limit;      // max(int(clip * area / 256), 1), computed on the host in double as NumPy does
    const int hv = hist[t];
    red[t] = max(hv - limit, 0);
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    const int excess = red[0];
    __syncthreads();
    int v = min(hv, limit) + excess / 256;
    const int rest = excess % 256;
    if (rest) {
        const int step = max(256 / rest, 1);
        if (t % step == 0 && t / step < rest) v += 1;
    }
    red[t] = v;           // inclusive scan (Hillis-Steele)
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int add = t >= o ? red[t - o] : 0;
        __syncthreads();
        red[t] += add;
        __syncthreads();
    }
    luts[(((size_t)b * kTiles + ty) * kTiles + tx) * 256 + t] = fminf(fmaxf(rintf(__fmul_rn((float)red[t], __fdiv_rn(255.0f, (float)area))), 0.f), 255.f);
}
__global__ void aug_clahe_apply_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int s,
                                       const vs_aug_params* __restrict__ params, const float* __restrict__ luts) {
    const int b = blockIdx.y;
    const bool on = params[b].clahe_clip != 0.f;
    const float th = (float)(s / kTiles);
    const float* L = luts + (size_t)b * kTiles * kTiles * 256;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < s * s; i += gridDim.x * 256) {
        const uint8_t v = src[(size_t)b * s * s + i];
        if (!on) { dst[(size_t)b * s * s + i] = v; continue; }
        const int y = i / s, x = i - y * s;
        const float yy = __fsub_rn(__fdiv_rn((float)y + 0.5f, th), 0.5f), xx = __fsub_rn(__fdiv_rn((float)x + 0.5f, th), 0.5f);
        const float y0f = floorf(yy), x0f = floorf(xx);
        const float wy = yy - y0f, wx = xx - x0f;
        const int y0 = min(max((int)y0f, 0), kTiles - 1), y1 = min(max((int)y0f + 1, 0), kTiles - 1);
        const int x0 = min(max((int)x0f, 0), kTiles - 1), x1 = min(max((int)x0f + 1, 0), kTiles - 1);
        const float a = L[(y0 * kTiles + x0) * 256 + v], bb = L[(y0 * kTiles + x1) * 256 + v];
        const float c = L[(y1 * kTiles + x0) * 256 + v], d = L[(y1 * kTiles + x1) * 256 + v];
        const float top = __fadd_rn(__fmul_rn(a, 1.f - wx), __fmul_rn(bb, wx)), bot = __fadd_rn(__fmul_rn(c, 1.f - wx), __fmul_rn(d, wx));
        const float o = __fadd_rn(__fmul_rn(top, 1.f - wy), __fmul_rn(bot, wy));
        dst[(size_t)b * s * s + i] = (uint8_t)fminf(fmaxf(rintf(o), 0.f), 255.f);
    }
}

// ---- intensity LUT + normalisation ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aug_finish_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ msk, int s,
                                                       const uint8_t* __restrict__ luts, float* __restrict__ x, uint8_t* __restrict__ out_m) {
    __shared__ uint8_t lut[256];
    const int b = blockIdx.y;
    lut[threadIdx.x] = luts[(size_t)b * 256 + threadIdx.x];
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < s * s; i += gridDim.x * 256) {
        const float u = (float)lut[img[(size_t)b * s * s + i]];
        x[(size_t)b * s * s + i] = __fdiv_rn(__fsub_rn(__fdiv_rn(u, 255.0f), 0.449f), 0.226f);
        out_m[(size_t)b * s * s + i] = msk[(size_t)b * s * s + i];
    }
}

}  // namespace

extern "C" size_t vs_augment_workspace(int n, int size) {
    const size_t plane = (size_t)n * size * size;
    return 4 * plane + 2 * 2 * plane * sizeof(float) + (size_t)n * kTiles * kTiles * 256 * sizeof(float) + 1024;
}

extern "C" int vs_augment_batch(const uint8_t* images, const uint8_t* masks, int n, int size, const vs_aug_params* params_dev,
                                const uint8_t* luts_dev, const float* grid_tables_dev, float* out_x, uint8_t* out_masks,
                                void* workspace, size_t workspace_bytes, float* fields_out, void* stream) {
    VS_REQUIRE(images && masks && params_dev && luts_dev && grid_tables_dev && out_x && out_masks && workspace, "augment_batch: null pointer");
    VS_REQUIRE(n >= 1 && size >= 16 && size % kTiles == 0 && size <= 4096, "augment_batch: size must be a multiple of 8 in [16, 4096], got %d", size);
    VS_REQUIRE(workspace_bytes >= vs_augment_workspace(n, size), "augment_batch: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t plane = (size_t)n * size * size;
    uint8_t* ia = (uint8_t*)workspace;
    uint8_t* ib = ia + plane;
    uint8_t* ma = ib + plane;
    uint8_t* mb = ma + plane;
    float* f0 = reinterpret_cast<float*>((char*)workspace + align_up(4 * plane, 256));
    float* f1 = f0 + 2 * plane;
    float* cl = f1 + 2 * plane;
    const dim3 grid((unsigned)std::min(1024, cdiv(size * size, 256)), (unsigned)n), grid2((unsigned)std::min(2048, cdiv(2 * size * size, 256)), (unsigned)n);
    // displacement fields of the elastic samples (alpha 120, sigma 8.4: augmentations.py:86-88)
    hipLaunchKernelGGL(aug_noise_kernel, grid2, dim3(256), 0, s, f0, size, params_dev);
    hipLaunchKernelGGL(aug_blur_kernel<0>, grid2, dim3(256), 0, s, f0, f1, size, 120.f * 0.07f, 1.f, params_dev);
    hipLaunchKernelGGL(aug_blur_kernel<1>, grid2, dim3(256), 0, s, f1, f0, size, 120.f * 0.07f, 120.f, params_dev);
    VS_LAUNCH_CHECK();
    if (fields_out) VS_CHECK_HIP(hipMemcpyAsync(fields_out, f0, 2 * plane * sizeof(float), hipMemcpyDeviceToDevice, s));   // tests: the fields that were used
    hipLaunchKernelGGL(aug_stage_kernel<ST_CROP>, grid, dim3(256), 0, s, images, masks, ia, ma, size, params_dev, grid_tables_dev, f0);
    hipLaunchKernelGGL(aug_stage_kernel<ST_DIHEDRAL>, grid, dim3(256), 0, s, ia, ma, ib, mb, size, params_dev, grid_tables_dev, f0);
    hipLaunchKernelGGL(aug_stage_kernel<ST_AFFINE>, grid, dim3(256), 0, s, ib, mb, ia, ma, size, params_dev, grid_tables_dev, f0);
    hipLaunchKernelGGL(aug_stage_kernel<ST_DISTORT>, grid, dim3(256), 0, s, ia, ma, ib, mb, size, params_dev, grid_tables_dev, f0);
    VS_LAUNCH_CHECK();
    hipLaunchKernelGGL(aug_clahe_lut_kernel, dim3(kTiles * kTiles, n), dim3(256), 0, s, ib, size, params_dev, cl);
    hipLaunchKernelGGL(aug_clahe_apply_kernel, grid, dim3(256), 0, s, ib, ia, size, params_dev, cl);
    hipLaunchKernelGGL(aug_finish_kernel, grid, dim3(256), 0, s, ia, mb, size, luts_dev, out_x, out_masks);
    VS_LAUNCH_CHECK();
    return VS_OK;
}
