// Standalone laboratory for the convolution kernels (needs a GPU): runs conv_ring_kernel variants and the library's
// conv_igemm path on the layer shapes of the batch-32 256x256 training step, checks them against a naive fp32 kernel and
// prints back-to-back launch times, interleaved in one process.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/convlab/convlab.hip -Lvolume-segmantics_amd/lib -lvolseg_hip -o gpurun_out/convlab
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../volume-segmantics_amd/csrc/conv_stream.h"

void vs_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

// naive reference: one thread per output element, fp32 accumulation in the kernel's chunk / tap order is NOT reproduced -
// the comparison is a tolerance one.  Virtual input = cat(up(src0), src1)
__global__ void ref_conv(const uint16_t* s0, const uint16_t* s1, const uint16_t* w, float* y, int N, int H, int W, int C0, int C1, int up0, int Cout) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)N * H * W * Cout;
    if (idx >= total) return;
    const int co = idx % Cout; long r = idx / Cout;
    const int wo = r % W; r /= W;
    const int ho = r % H; const int n = r / H;
    const int Cin = C0 + C1, ush = up0 ? 1 : 0, H0 = H >> ush, W0 = W >> ush;
    float a = 0.f;
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
            const int hi = ho + kh - 1, wi = wo + kw - 1;
            if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
            const uint16_t* wp = w + ((long)co * 9 + kh * 3 + kw) * Cin;
            if (!(up0 == 2 && ((hi | wi) & 1))) {
                const uint16_t* xp = s0 + (((long)n * H0 + (hi >> ush)) * W0 + (wi >> ush)) * C0;
                for (int c = 0; c < C0; ++c) a += __uint_as_float((uint32_t)xp[c] << 16) * __uint_as_float((uint32_t)wp[c] << 16);
            }
            if (C1) {
                const uint16_t* xp = s1 + (((long)n * H + hi) * W + wi) * C1;
                for (int c = 0; c < C1; ++c) a += __uint_as_float((uint32_t)xp[c] << 16) * __uint_as_float((uint32_t)wp[C0 + c] << 16);
            }
        }
    y[idx] = a;
}

// helpers of the chain experiments: a streaming rewrite of a tensor (y = x, 16 B per lane) and a one-block no-op
__global__ void copy16_kernel(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ void tiny_kernel(int* p) { if (threadIdx.x == 999) *p = 1; }

struct Shape { const char* name; int N, H, C0, C1, up0, Cout; };

// mean duration (us) of the four phases between the five 100 MHz stamps every workgroup left in a probe buffer
static void print_probe(const char* what, unsigned long long* dbuf, size_t cap) {
    std::vector<unsigned long long> h(cap * 8);
    CK(hipMemcpy(h.data(), dbuf, cap * 64, hipMemcpyDeviceToHost));
    double ph[4] = {0, 0, 0, 0}, life = 0; size_t n = 0; unsigned long long tmin = ~0ull, tmax = 0, smax = 0;
    {   // conv_stream_kernel's own record: begin, loop start, summed tap-7 waits, summed epilogues, end, tiles
        double lf = 0, pro = 0, wt = 0, ep = 0, nt = 0; size_t m = 0;
        for (size_t i = 0; i < cap; ++i) {
            const unsigned long long* o = &h[i * 8];
            if (!o[0] || o[7] != 0x53) continue;
            ++m; lf += (o[4] - o[0]) * 0.01; pro += (o[1] - o[0]) * 0.01; wt += o[2] * 0.01; ep += o[3] * 0.01; nt += (double)o[5];
        }
        if (m) {
            printf("      [probe] %s: %zu WGs, life %.1f us, prologue %.2f us, %.1f tiles per WG: per tile %.2f us = waits at the chunk barriers %.2f + epilogue %.2f + rest %.2f\n",
                   what, m, lf / m, pro / m, nt / m, (lf - pro) / nt, wt / nt, ep / nt, (lf - pro - wt - ep) / nt);
            return;
        }
    }
    for (size_t i = 0; i < cap; ++i) {
        const unsigned long long* o = &h[i * 8];
        if (!o[0]) continue;
        ++n;
        for (int k = 0; k < 4; ++k) ph[k] += (double)(o[k + 1] - o[k]) * 0.01;
        life += (double)(o[4] - o[0]) * 0.01;
        tmin = std::min(tmin, o[0]); tmax = std::max(tmax, o[4]); smax = std::max(smax, o[0]);
    }
    if (!n) { printf("      [probe] %s: no stamps\n", what); return; }
    printf("      [probe] %s: %zu WGs, span %.1f us, starts spread %.1f us, WG life %.1f us = setup %.2f | first data in LDS %.2f | main loop %.2f | epilogue + store ack %.2f\n",
           what, n, (tmax - tmin) * 0.01, (smax - tmin) * 0.01, life / n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n);
}

static unsigned long long* g_probe = nullptr;   // passed to the ring launches when a timeline is wanted
struct Variant { std::string name; int (*fn)(const ConvParams&, hipStream_t); bool (*ok)(const ConvParams&); };

template <int BN, int PT, int NW, int TWS, int IMGS, int WPS, int PIN = 0>
static int run_ring(const ConvParams& p, hipStream_t s) { return ring::launch_ring<BN, PT, NW, TWS, IMGS, WPS, PIN>(p, 0, g_probe, s); }
template <int PT, int NW, int TWS, int IMGS>
static bool ok_ring(const ConvParams& p) { return ring::ring_geom_ok<PT, NW, TWS, IMGS>(p) && (IMGS > 1 || (p.Hout >= 8 && p.Wout >= (1 << TWS))); }
template <int BN, int PT, int NW, int TWS, int WPS, int PIN = 2>
static int run_stream(const ConvParams& p, hipStream_t s) { return ring::launch_stream<bf16_t, BN, PT, NW, TWS, WPS, PIN>(p, g_probe, s, 256, getenv("STREAM_STAGGER") ? atoi(getenv("STREAM_STAGGER")) : 0); }
static bool ok_stream(const ConvParams& p) { return ring::stream_ok(p, 0); }
static int run_lib(const ConvParams& p, hipStream_t s) {
    vs_conv_desc d{};
    d.dtype = VS_BF16; d.n = p.N; d.hin = p.Hin; d.win = p.Win; d.c0 = p.C0; d.c1 = p.C1; d.up0 = p.up0; d.cout = p.Cout;
    d.kh = d.kw = 3; d.stride = 1; d.pad = 1;
    return vs_conv2d_fwd(&d, p.src0, p.src1, p.w, nullptr, nullptr, nullptr, p.out, nullptr, (void*)s);
}
static bool ok_any(const ConvParams&) { return true; }

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 30;
    const int rounds = argc > 2 ? atoi(argv[2]) : 3;
    const bool do_probe = argc > 3 && atoi(argv[3]) != 0, do_wgrad = argc > 4 && atoi(argv[4]) != 0, do_conv = !(argc > 5 && atoi(argv[5]) == 0), do_chain = argc > 6 && atoi(argv[6]) != 0, do_bn = argc > 7 && atoi(argv[7]) != 0;
    const bool pred_shapes = argc > 8 && atoi(argv[8]) != 0;     // the layers of a 512^2 prediction forward at batch 64 instead
    std::vector<Shape> shapes = {
        {"layer1 64->64 @64", 32, 64, 64, 0, 0, 64},
        {"layer2 128->128 @32", 32, 32, 128, 0, 0, 128},
        {"layer3 256->256 @16", 32, 16, 256, 0, 0, 256},
        {"layer4 512->512 @8", 32, 8, 512, 0, 0, 512},
        {"dec0.c1 up512+256->256 @16", 32, 16, 512, 256, 1, 256},
        {"dec1.c1 up256+128->128 @32", 32, 32, 256, 128, 1, 128},
        {"dec2.c1 up128+64->64 @64", 32, 64, 128, 64, 1, 64},
        {"dec3.c1 up64+64->32 @128", 32, 128, 64, 64, 1, 32},
        {"dec3.c2 32->32 @128", 32, 128, 32, 0, 0, 32},
        {"dgrad-s2 stuffed 128->64 @64", 32, 64, 128, 0, 2, 64},
        {"dec4.c1 up32->16 @256", 32, 256, 32, 0, 1, 16},
        {"ragged 40->48 @24 (n=3)", 3, 24, 40, 0, 0, 48},
    };
    if (pred_shapes)
        shapes = {
            {"P layer1 64->64 @128", 64, 128, 64, 0, 0, 64},
            {"P layer2 128->128 @64", 64, 64, 128, 0, 0, 128},
            {"P layer3 256->256 @32", 64, 32, 256, 0, 0, 256},
            {"P layer4 512->512 @16", 64, 16, 512, 0, 0, 512},
            {"P dec0.c1 up512+256->256 @32", 64, 32, 512, 256, 1, 256},
            {"P dec1.c1 up256+128->128 @64", 64, 64, 256, 128, 1, 128},
            {"P dec2.c1 up128+64->64 @128", 64, 128, 128, 64, 1, 64},
            {"P dec3.c1 up64+64->32 @256", 64, 256, 64, 64, 1, 32},
            {"P dec3.c2 32->32 @256", 64, 256, 32, 0, 0, 32},
        };
    std::vector<Variant> vars = {
        {"lib (conv_igemm)", run_lib, ok_any},
        {"stream BN64 PT2 NW8 16x16 pin2", run_stream<64, 2, 8, 4, 2, 2>, ok_stream},
        {"stream BN64 PT2 NW8 16x16 pin0", run_stream<64, 2, 8, 4, 2, 0>, ok_stream},
        {"stream BN32 PT2 NW8 16x16 pin2", run_stream<32, 2, 8, 4, 2, 2>, ok_stream},
        {"ring BN64 PT2 NW4 16x8  1w pin0", run_ring<64, 2, 4, 4, 1, 1, 0>, ok_ring<2, 4, 4, 1>},
        {"ring BN64 PT2 NW4 16x8  1w pin1", run_ring<64, 2, 4, 4, 1, 1, 1>, ok_ring<2, 4, 4, 1>},
        {"ring BN64 PT2 NW4 16x8  1w pin2", run_ring<64, 2, 4, 4, 1, 1, 2>, ok_ring<2, 4, 4, 1>},
        {"ring BN64 PT2 NW8 16x16 2w pin0", run_ring<64, 2, 8, 4, 1, 2, 0>, ok_ring<2, 8, 4, 1>},
        {"ring BN64 PT2 NW8 16x16 2w pin2", run_ring<64, 2, 8, 4, 1, 2, 2>, ok_ring<2, 8, 4, 1>},
        {"ring BN64 PT4 NW4 16x16 1w pin0", run_ring<64, 4, 4, 4, 1, 1, 0>, ok_ring<4, 4, 4, 1>},
        {"ring BN64 PT4 NW4 16x16 1w pin1", run_ring<64, 4, 4, 4, 1, 1, 1>, ok_ring<4, 4, 4, 1>},
        {"ring BN64 PT4 NW4 16x16 1w pin2", run_ring<64, 4, 4, 4, 1, 1, 2>, ok_ring<4, 4, 4, 1>},
        {"ring BN32 PT2 NW8 16x16 4w pin0", run_ring<32, 2, 8, 4, 1, 4, 0>, ok_ring<2, 8, 4, 1>},
        {"ring BN32 PT2 NW8 16x16 4w pin2", run_ring<32, 2, 8, 4, 1, 4, 2>, ok_ring<2, 8, 4, 1>},
        {"ring BN32 PT2 NW4 16x8  2w pin0", run_ring<32, 2, 4, 4, 1, 2, 0>, ok_ring<2, 4, 4, 1>},
        {"ring BN32 PT4 NW4 16x16 1w pin2", run_ring<32, 4, 4, 4, 1, 1, 2>, ok_ring<4, 4, 4, 1>},
        {"ring BN32 PT4 NW4 16x16 2w pin2", run_ring<32, 4, 4, 4, 1, 2, 2>, ok_ring<4, 4, 4, 1>},
        {"ring BN32 PT2 NW4 8x8 x2img pin0", run_ring<32, 2, 4, 3, 2, 1, 0>, ok_ring<2, 4, 3, 2>},
        {"ring BN32 PT2 NW4 8x8 x2img pin2", run_ring<32, 2, 4, 3, 2, 1, 2>, ok_ring<2, 4, 3, 2>},
        {"ring BN32 PT2 NW8 8x8 x4img pin2", run_ring<32, 2, 8, 3, 4, 2, 2>, ok_ring<2, 8, 3, 4>},
    };
    if (!do_conv) vars.resize(1);
    if (const char* only = getenv("CONVLAB_ONLY")) {      // keep the library path and the variants whose name contains the string
        std::vector<Variant> keep;
        for (auto& v : vars) if (v.fn == run_lib || v.name.find(only) != std::string::npos) keep.push_back(v);
        vars = keep;
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    srand(1234);
    for (const Shape& sh : shapes) {
        const int ush = sh.up0 ? 1 : 0, H0 = sh.H >> ush;
        const size_t n0 = (size_t)sh.N * H0 * H0 * sh.C0, n1 = (size_t)sh.N * sh.H * sh.H * sh.C1;
        const int Cin = sh.C0 + sh.C1;
        const size_t nw = (size_t)sh.Cout * 9 * Cin, ny = (size_t)sh.N * sh.H * sh.H * sh.Cout;
        std::vector<uint16_t> h0(n0), h1(std::max<size_t>(n1, 1)), hw(nw);
        for (auto& v : h0) v = f2bf((rand() / (float)RAND_MAX) * 2.f - 1.f);
        for (auto& v : h1) v = f2bf((rand() / (float)RAND_MAX) * 2.f - 1.f);
        for (auto& v : hw) v = f2bf(((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f);
        uint16_t *d0, *d1, *dw, *dy; float* dref;
        CK(hipMalloc(&d0, n0 * 2)); CK(hipMalloc(&d1, std::max<size_t>(n1, 1) * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dy, ny * 2));
        CK(hipMalloc(&dref, ny * 4));
        CK(hipMemcpy(d0, h0.data(), n0 * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(d1, h1.data(), std::max<size_t>(n1, 1) * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(ref_conv, dim3((unsigned)((ny + 255) / 256)), dim3(256), 0, st, d0, sh.C1 ? d1 : nullptr, dw, dref, sh.N, sh.H, sh.H, sh.C0, sh.C1, sh.up0, sh.Cout);
        CK(hipStreamSynchronize(st));
        std::vector<float> ref(ny);
        CK(hipMemcpy(ref.data(), dref, ny * 4, hipMemcpyDeviceToHost));
        double rms = 0; for (float v : ref) rms += (double)v * v; rms = std::sqrt(rms / ny);
        ConvParams p{};
        p.src0 = d0; p.src1 = sh.C1 ? d1 : nullptr; p.C0 = sh.C0; p.C1 = sh.C1; p.up0 = sh.up0;
        p.N = sh.N; p.Hin = p.Win = p.Hout = p.Wout = sh.H; p.stride = 1; p.pad = 1; p.KH = p.KW = 3; p.w = dw; p.Cout = sh.Cout; p.out = dy;
        const double flops = 2.0 * ny * 9 * Cin * (sh.up0 == 2 ? 0.25 * 1.0 : 1.0);
        printf("== %s: N=%d, %.2f GFLOP (dense count), out rms %.3f\n", sh.name, sh.N, 2.0 * ny * 9 * Cin * 1e-9, rms);
        (void)flops;
        std::vector<std::vector<float>> times(vars.size());
        std::vector<double> errs(vars.size(), -1.0);
        for (size_t v = 0; v < vars.size(); ++v) {
            if (!vars[v].ok(p)) continue;
            CK(hipMemsetAsync(dy, 0xff, ny * 2, st));
            if (vars[v].fn(p, st)) { printf("   %-40s launch refused\n", vars[v].name.c_str()); times[v].clear(); errs[v] = -2; continue; }
            CK(hipStreamSynchronize(st));
            std::vector<uint16_t> out(ny);
            CK(hipMemcpy(out.data(), dy, ny * 2, hipMemcpyDeviceToHost));
            double mx = 0;
            for (size_t i = 0; i < ny; ++i) { const double d = std::fabs((double)bf2f(out[i]) - ref[i]); if (!(d <= mx)) mx = d; }
            errs[v] = mx / rms;
        }
        for (int r = 0; r < rounds; ++r)
            for (size_t v = 0; v < vars.size(); ++v) {
                if (errs[v] < 0) continue;
                for (int i = 0; i < 3; ++i) vars[v].fn(p, st);
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < reps; ++i) vars[v].fn(p, st);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                times[v].push_back(ms * 1e3f / reps);
            }
        for (size_t v = 0; v < vars.size(); ++v) {
            if (errs[v] < 0) continue;
            std::sort(times[v].begin(), times[v].end());
            const float med = times[v][times[v].size() / 2];
            printf("   %-40s %7.2f us (min %7.2f)  %7.0f TFLOP/s   max err / rms %.2e %s\n", vars[v].name.c_str(), med, times[v][0],
                   2.0 * ny * 9 * Cin / med * 1e-6, errs[v], errs[v] > 2e-2 ? "  <-- WRONG" : "");
        }
        if (do_probe) {
            const size_t cap = 1 << 15;
            unsigned long long* pb; CK(hipMalloc(&pb, cap * 64));
            for (size_t v = 0; v < vars.size(); ++v) {
                if (errs[v] < 0) continue;
                CK(hipMemset(pb, 0, cap * 64));
                if (v == 0) vs_debug_probe(pb, cap); else g_probe = pb;
                vars[v].fn(p, st);
                CK(hipStreamSynchronize(st));
                vs_debug_probe(nullptr, 0); g_probe = nullptr;
                print_probe(vars[v].name.c_str(), pb, cap);
            }
            CK(hipFree(pb));
        }
        if (do_chain && sh.C1 == 0 && sh.up0 == 0 && sh.C0 == sh.Cout) {
            // how much of an in-step convolution's time is "cold start"?  (a) the same launch back to back; (b) ping-pong
            // x -> y -> x (every input freshly written by the previous launch, same code); (c) a one-block kernel between
            // launches (another code object in between, data untouched); (d) a streaming rewrite of the input between
            // launches (other code AND a freshly written input - what the training step does)
            uint16_t* dx2; CK(hipMalloc(&dx2, n0 * 2));
            int* dflag; CK(hipMalloc(&dflag, 4));
            for (size_t v = 0; v < vars.size(); ++v) {
                if (errs[v] < 0) continue;
                ConvParams pa = p, pb = p;
                pb.src0 = dy; pb.out = d0;
                auto timeit = [&](auto&& body) {
                    for (int i = 0; i < 3; ++i) body();
                    CK(hipEventRecord(e0, st));
                    for (int i = 0; i < reps; ++i) body();
                    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f / reps;
                };
                const float ta = timeit([&] { vars[v].fn(pa, st); });
                const float tb = timeit([&] { vars[v].fn(pa, st); vars[v].fn(pb, st); }) / 2;
                const float tt = timeit([&] { hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, st, dflag); });
                const float tc = timeit([&] { vars[v].fn(pa, st); hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, st, dflag); }) - tt;
                const float tcp = timeit([&] { hipLaunchKernelGGL(copy16_kernel, dim3(2048), dim3(256), 0, st, (const uint4*)dx2, (uint4*)d0, n0 / 8); });
                const float td = timeit([&] { vars[v].fn(pa, st); hipLaunchKernelGGL(copy16_kernel, dim3(2048), dim3(256), 0, st, (const uint4*)dx2, (uint4*)d0, n0 / 8); }) - tcp;
                // (e) everything cold: a 1 GB streaming rewrite of an unrelated buffer between launches (L2 and the Infinity
                // Cache hold neither the input nor the weights); timed with events around the convolution alone
                float te = 0.f;
                {
                    static uint4* big = nullptr; const size_t bign = (size_t)1 << 26;   // 64 M x 16 B = 1 GiB
                    if (!big) CK(hipMalloc(&big, bign * 16));
                    const int nrep = 8;
                    for (int i = 0; i < nrep + 1; ++i) {
                        hipLaunchKernelGGL(copy16_kernel, dim3(4096), dim3(256), 0, st, (const uint4*)big + bign / 2, big, bign / 2);
                        CK(hipEventRecord(e0, st));
                        vars[v].fn(pa, st);
                        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i) te += ms * 1e3f / nrep;
                    }
                }
                printf("   [chain] %-34s cold (1 GB streamed between) %6.2f |", vars[v].name.c_str(), te);
                printf("   [chain] %-34s back-to-back %6.2f | ping-pong %6.2f | + tiny kernel between %6.2f (tiny alone %.2f) | + input rewritten between %6.2f (copy alone %.2f) us\n",
                       vars[v].name.c_str(), ta, tb, tc, tt, td, tcp);
            }
            CK(hipMemcpy(d0, h0.data(), n0 * 2, hipMemcpyHostToDevice));
            CK(hipFree(dx2)); CK(hipFree(dflag));
        }
        if (do_bn && sh.C1 == 0 && sh.up0 == 0) {   // BatchNorm backward on this layer's output tensor: three launches vs one
            const int c = sh.Cout; const long rows = (long)sh.N * sh.H * sh.H;
            uint16_t *bdy, *bz, *bdx; float *bst, *bws;
            CK(hipMalloc(&bdy, ny * 2)); CK(hipMalloc(&bz, ny * 2)); CK(hipMalloc(&bdx, ny * 2));
            CK(hipMalloc(&bst, 6 * c * 4)); const size_t wsb = vs_bn_workspace(rows, c); CK(hipMalloc(&bws, wsb));
            CK(hipMemcpy(bz, dy, ny * 2, hipMemcpyDeviceToDevice)); CK(hipMemcpy(bdy, dy, ny * 2, hipMemcpyDeviceToDevice));
            std::vector<float> stv(6 * c, 0.f); for (int i = 0; i < c; ++i) { stv[c + i] = 1.f; stv[2 * c + i] = 1.f; stv[3 * c + i] = 0.1f; }
            CK(hipMemcpy(bst, stv.data(), 6 * c * 4, hipMemcpyHostToDevice));
            auto run = [&] { return vs_bn_bwd_recompute(VS_BF16, bdy, nullptr, bz, bst, bst + c, bst + 2 * c, bst + 3 * c, 1, bdx, nullptr, bst + 4 * c, bst + 5 * c, rows, c, bws, wsb, st); };
            auto timeit = [&] { for (int i = 0; i < 3; ++i) run(); CK(hipEventRecord(e0, st)); for (int i = 0; i < reps; ++i) run(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f / reps; };
            vs_set_option("bn_bwd_fused", 0);
            const float t3 = timeit();
            std::vector<uint16_t> r0(ny), r1(ny); CK(hipMemcpy(r0.data(), bdx, ny * 2, hipMemcpyDeviceToHost));
            vs_set_option("bn_bwd_fused", 1);
            printf("   [bn_bwd] %.1f MB tensors, c = %d: three launches %6.2f us |", ny * 2 / 1e6, c, t3);
            for (int blocks : {0, 32, 64, 128, 256}) {
                if (blocks > 128 && (2 * c / 4) * 1 > 256) continue;
                vs_set_option("bn_fused_blocks", blocks);
                vs_set_option("bn_fused_dbg", 0); const float tf = timeit();
                if (blocks == 0) { CK(hipMemcpy(r1.data(), bdx, ny * 2, hipMemcpyDeviceToHost)); size_t bad = 0; for (size_t i = 0; i < ny; ++i) bad += r0[i] != r1[i]; printf(" (fused vs plain: %zu of %zu bf16 values differ)", bad, ny); }
                vs_set_option("bn_fused_dbg", 1); const float tnb = timeit();
                vs_set_option("bn_fused_dbg", 3); const float ts1 = timeit();
                printf(" blocks %3d: %6.2f (no wait %6.2f, sweep 1 + sums only %6.2f) |", blocks, tf, tnb, ts1);
            }
            printf("\n");
            vs_set_option("bn_fused_blocks", 0); vs_set_option("bn_fused_dbg", 0);
            CK(hipFree(bdy)); CK(hipFree(bz)); CK(hipFree(bdx)); CK(hipFree(bst)); CK(hipFree(bws));
        }
        if (do_wgrad && sh.up0 != 2) {   // the library's weight gradient on the same layer (dy = the reference output, rounded)
            vs_conv_desc d{};
            d.dtype = VS_BF16; d.n = sh.N; d.hin = sh.H; d.win = sh.H; d.c0 = sh.C0; d.c1 = sh.C1; d.up0 = sh.up0; d.cout = sh.Cout;
            d.kh = d.kw = 3; d.stride = 1; d.pad = 1;
            float* ddw; CK(hipMalloc(&ddw, nw * 4));
            // correctness of the ring kernel against the plain one (same products, another summation order)
            {
                std::vector<float> r0(nw), r1(nw);
                const size_t wsb = vs_conv2d_wgrad_workspace(&d);
                void* ws; CK(hipMalloc(&ws, std::max<size_t>(wsb, 16)));
                for (int ring = 0; ring < 2; ++ring) {
                    vs_set_option("wgrad_ring", ring);
                    CK(hipMemsetAsync(ddw, 0xff, nw * 4, st));
                    if (vs_conv2d_wgrad(&d, d0, sh.C1 ? d1 : nullptr, dy, ddw, ws, wsb, st)) printf("   wgrad refused\n");
                    CK(hipStreamSynchronize(st));
                    CK(hipMemcpy((ring ? r1 : r0).data(), ddw, nw * 4, hipMemcpyDeviceToHost));
                }
                double rr = 0, mx = 0;
                for (size_t i = 0; i < nw; ++i) { rr += (double)r0[i] * r0[i]; const double dd = std::fabs((double)r0[i] - r1[i]); if (!(dd <= mx)) mx = dd; }
                rr = std::sqrt(rr / nw);
                printf("   wgrad ring vs plain: max |diff| / rms = %.2e (rms %.3g) %s\n", mx / rr, rr, mx / rr > 1e-3 ? " <-- WRONG" : "");
                CK(hipFree(ws));
            }
            const int combos[][3] = {{256, 16, 0}, {256, 16, 1}, {384, 32, 1}, {512, 32, 1}, {512, 64, 1}};
            for (auto& cb : combos) {
                vs_set_option("wgrad_target", cb[0]); vs_set_option("wgrad_slab_mb", cb[1]); vs_set_option("wgrad_ring", cb[2]);
                const size_t wsb = vs_conv2d_wgrad_workspace(&d);
                void* ws; CK(hipMalloc(&ws, std::max<size_t>(wsb, 16)));
                std::vector<float> tt;
                for (int r = 0; r < rounds; ++r) {
                    for (int i = 0; i < 3; ++i) vs_conv2d_wgrad(&d, d0, sh.C1 ? d1 : nullptr, dy, ddw, ws, wsb, st);
                    CK(hipEventRecord(e0, st));
                    for (int i = 0; i < reps; ++i) vs_conv2d_wgrad(&d, d0, sh.C1 ? d1 : nullptr, dy, ddw, ws, wsb, st);
                    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tt.push_back(ms * 1e3f / reps);
                }
                std::sort(tt.begin(), tt.end());
                printf("   wgrad %s target %4d slab_mb %3d: %7.2f us (kernel + slab reduce), %6.0f TFLOP/s, workspace %.1f MB\n", cb[2] ? "RING " : "plain", cb[0], cb[1],
                       tt[tt.size() / 2], 2.0 * ny * 9 * Cin / tt[tt.size() / 2] * 1e-6, wsb / 1048576.0);
                if (do_probe && cb[1] == 16) {
                    const size_t cap = 1 << 15;
                    unsigned long long* pb; CK(hipMalloc(&pb, cap * 64)); CK(hipMemset(pb, 0, cap * 64));
                    vs_debug_probe(pb, cap);
                    vs_conv2d_wgrad(&d, d0, sh.C1 ? d1 : nullptr, dy, ddw, ws, wsb, st);
                    CK(hipStreamSynchronize(st));
                    vs_debug_probe(nullptr, 0);
                    print_probe("wgrad kernel", pb, cap);
                    CK(hipFree(pb));
                }
                CK(hipFree(ws));
            }
            vs_set_option("wgrad_ring", 1);
            vs_set_option("wgrad_target", 256); vs_set_option("wgrad_slab_mb", 16);
            CK(hipFree(ddw));
        }
        fflush(stdout);
        CK(hipFree(d0)); CK(hipFree(d1)); CK(hipFree(dw)); CK(hipFree(dy)); CK(hipFree(dref));
    }
    return 0;
}
