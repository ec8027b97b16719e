#include "prof.h"

#include <cctype>
#include <cstdlib>
#include <vector>

#include "common.h"

namespace {
struct Rec { int kind, tag, variant; double flops, bytes; hipEvent_t e0, e1; };
int g_tag = -1, g_variant = 0;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_used = 0;
hipEvent_t get_event() {
    if (g_pool_used == g_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        g_pool.push_back(e);
    }
    return g_pool[g_pool_used++];
}
const char* kNames[PK_COUNT] = {"conv_fwd", "conv_dgrad", "conv_wgrad", "stem", "bn_stats", "bn_apply", "bn_bwd",
                                "pool_misc", "prepare", "head"};
}  // namespace

bool prof_on() { return g_on; }
void prof_begin(int kind, double flops, double bytes, hipStream_t s) {
    Rec r{kind, g_tag, g_variant, flops, bytes, get_event(), get_event()};
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
}
void prof_end(hipStream_t s) { (void)hipEventRecord(g_recs.back().e1, s); }
void prof_add_flops(double flops) { if (g_on && !g_recs.empty()) g_recs.back().flops += flops; }

void prof_set_tag(int tag) { g_tag = tag; }
void prof_set_variant(int v) { g_variant = v; }

// raw records (kind, tag = unit index, ms, flops, bytes) in launch order; returns the number written
extern "C" int vs_profile_read_raw(int max_n, int* kind, int* tag, int* variant, double* ms, double* flops, double* bytes) {
    int n = 0;
    for (auto& r : g_recs) {
        if (n >= max_n) break;
        if (hipEventSynchronize(r.e1) != hipSuccess) return -1;
        float t = 0;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return -1;
        kind[n] = r.kind; tag[n] = r.tag; variant[n] = r.variant; ms[n] = t; flops[n] = r.flops; bytes[n] = r.bytes;
        ++n;
    }
    return n;
}

// enable/disable; enabling clears previously collected records
extern "C" int vs_profile_enable(int on) {
    g_on = on != 0;
    if (g_on) { g_recs.clear(); g_pool_used = 0; }
    return VS_OK;
}
extern "C" int vs_profile_enabled(void) { return g_on ? 1 : 0; }
extern "C" int vs_profile_num_kinds(void) { return PK_COUNT; }
extern "C" const char* vs_profile_kind_name(int kind) { return kind >= 0 && kind < PK_COUNT ? kNames[kind] : "?"; }
// sums over all records since vs_profile_enable(1): per kind elapsed ms, algorithmic flops, algorithmic bytes, launches
extern "C" int vs_profile_read(double* ms, double* flops, double* bytes, int64_t* calls) {
    for (int k = 0; k < PK_COUNT; ++k) { ms[k] = flops[k] = bytes[k] = 0; calls[k] = 0; }
    for (auto& r : g_recs) {
        VS_CHECK_HIP(hipEventSynchronize(r.e1));
        float t = 0;
        VS_CHECK_HIP(hipEventElapsedTime(&t, r.e0, r.e1));
        ms[r.kind] += t; flops[r.kind] += r.flops; bytes[r.kind] += r.bytes; calls[r.kind] += 1;
    }
    return VS_OK;
}

// ---- runtime options (defaults can also come from the environment: VS_<NAME>) -----------------------------------------
namespace {
struct Opt { const char* name; int value; bool init; };
Opt g_opts[] = {
    // kernel families (each non-default value is exercised by a named test: tests/test_hip_baseline_sizes.py
    // test_every_kernel_choice_option_gives_the_same_training_step, tests/test_hip_ops.py, tests/test_hip_unet.py)
    {"side_stream", 1, false}, {"conv_direct", 1, false}, {"conv_nw8", 1, false}, {"conv_ring", 1, false}, {"conv_stream", 1, false},
    {"wgrad_ring", 1, false}, {"wgrad_xcd", 1, false}, {"stats_bins", 1, false}, {"fuse_bn_bwd", 1, false}, {"nl_fwd", 1, false},
    {"stem_bf16", 1, false}, {"conv_pair", 1, false},
    // launch-size thresholds and split sizes (tuned on the batch-32 step / the 512^3 prediction; tools/ab_option.py)
    {"conv_min_wgs", 512, false}, {"conv_nw8_min_wgs", 128, false}, {"conv_direct_min_px", 262144, false}, {"conv_direct_rows", 32, false},
    {"conv_direct_rows_big", 128, false}, {"conv_ring_max_wgs", 1024, false}, {"conv_stream_min_tiles", 2, false},
    {"wgrad_target", 96, false}, {"wgrad_target_plain", 256, false}, {"wgrad_slab_mb", 16, false}, {"fork_every", 2, false},
    {"bn_inline_rows", 64, false}, {"nl_max_c", 64, false}};
}
int vs_option(const char* name) {
    for (auto& o : g_opts) {
        if (strcmp(o.name, name)) continue;
        if (!o.init) {
            char env[64] = "VS_";
            for (size_t i = 0; name[i] && i < 50; ++i) env[3 + i] = (char)toupper(name[i]), env[4 + i] = 0;
            if (const char* e = getenv(env)) o.value = atoi(e);
            o.init = true;
        }
        return o.value;
    }
    return 0;
}
extern "C" int vs_set_option(const char* name, int value) {
    for (auto& o : g_opts)
        if (!strcmp(o.name, name)) { o.value = value; o.init = true; return VS_OK; }
    vs_set_error("unknown option %s", name);
    return VS_ERR_INVALID;
}
static unsigned long long* g_probe = nullptr;
static size_t g_probe_cap = 0;
unsigned long long* vs_probe_buffer(size_t need_wgs) { return (g_probe && need_wgs <= g_probe_cap) ? g_probe : nullptr; }
extern "C" int vs_debug_probe(void* buf, size_t cap_wgs) {
    g_probe = (unsigned long long*)buf;
    g_probe_cap = buf ? cap_wgs : 0;
    return VS_OK;
}
extern "C" int vs_get_option(const char* name) { return vs_option(name); }

// ---- diagnostics: what the MFMA pipes sustain on this box (bench.py's peak_crosscheck; tools/mfma_probe.hip is the stand-alone
// form).  Every wave issues iters x 8 v_mfma_f32_16x16x32_bf16 on register operands; 4-wave workgroups, waves_per_simd of them
// per CU.  Returns the launch's TFLOP/s and the mean in-kernel clock (s_memtime ticks / wall time) in GHz.
typedef __attribute__((ext_vector_type(4))) float dbg_f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 dbg_bf16x8;
__global__ __launch_bounds__(256) void debug_mfma_loop(float* out, unsigned long long* clocks, int iters) {
    dbg_f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = dbg_f32x4{0.f, 0.f, 0.f, 0.f};
    dbg_bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)((threadIdx.x + i) & 3); b[i] = (__bf16)(0.5f + (float)(i & 1)); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clocks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
extern "C" int vs_debug_mfma_rate(int iters, int waves_per_simd, double* tflops, double* clock_ghz) {
    VS_REQUIRE(iters > 0 && iters <= 1000000 && waves_per_simd >= 1 && waves_per_simd <= 8 && tflops && clock_ghz, "debug_mfma_rate: bad arguments");
    hipDeviceProp_t prop;
    int dev = 0;
    VS_CHECK_HIP(hipGetDevice(&dev));
    VS_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    const int blocks = prop.multiProcessorCount * waves_per_simd, waves = blocks * 4;
    float* out = nullptr; unsigned long long* clk = nullptr;
    VS_CHECK_HIP(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
    VS_CHECK_HIP(hipMalloc(&clk, (size_t)waves * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    VS_CHECK_HIP(hipEventCreate(&e0)); VS_CHECK_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(debug_mfma_loop, dim3(blocks), dim3(256), 0, 0, out, clk, 64);
    VS_CHECK_HIP(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(debug_mfma_loop, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    VS_CHECK_HIP(hipEventRecord(e1, 0));
    VS_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    VS_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long* h = (unsigned long long*)malloc((size_t)waves * sizeof(unsigned long long));
    VS_CHECK_HIP(hipMemcpy(h, clk, (size_t)waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0.0;
    for (int i = 0; i < waves; ++i) mean += (double)h[i];
    mean /= waves;
    free(h);
    (void)hipFree(out); (void)hipFree(clk); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *tflops = (double)waves * iters * 8 * 2.0 * 16 * 16 * 32 / ((double)ms * 1e-3) * 1e-12;
    *clock_ghz = mean / ((double)ms * 1e-3) * 1e-9;
    return VS_OK;
}
