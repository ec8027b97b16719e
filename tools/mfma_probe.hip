// Diagnostics (not part of the library): what v_mfma_f32_16x16x32_bf16 really sustains on this box.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o gpurun_out/mfma_probe && gpurun_out/mfma_probe
// Every wave issues `iters` x 8 MFMAs on 8 independent accumulators from register operands (no memory in the loop), for
// 1 / 2 / 4 / 8 waves per SIMD on all 256 CUs.  Prints TFLOP/s (wall clock over the launch) and the clocks one MFMA occupies
// a SIMD (s_memtime around the loop / MFMAs issued per SIMD) - the denominator a "fraction of MFMA peak" should use.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__global__ __launch_bounds__(256) void mfma_loop(float* out, unsigned long long* clocks, int iters) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
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
    if ((threadIdx.x & 63) == 0) clocks[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    const int iters = 20000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
    for (int wps : {1, 2, 4, 8}) {                 // waves per SIMD: blocks of 4 waves (one per SIMD), wps blocks per CU
        const int blocks = cus * wps, waves = blocks * 4;
        float* out; unsigned long long* clk;
        hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
        hipMalloc(&clk, (size_t)waves * sizeof(unsigned long long));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, clk, 100);      // warm-up
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(waves);
        hipMemcpy(h.data(), clk, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double mean = 0.0;
        for (auto v : h) mean += (double)v;
        mean /= waves;
        const double flops = (double)waves * iters * 8 * 2.0 * 16 * 16 * 32;
        // s_memtime / readcyclecounter ticks at a constant 100 MHz on this part: convert with the wall time instead
        const double mfma_per_simd = (double)iters * 8 * wps;
        printf("waves/SIMD %d: %.3f ms, %.1f TFLOP/s, %.2f us per 1000 MFMAs of one SIMD (= %.1f clocks per MFMA at %d MHz), counter ticks per wave %.0f\n",
               wps, ms, flops / ms * 1e-9, ms * 1e3 / mfma_per_simd * 1000.0, ms * 1e-3 / mfma_per_simd * prop.clockRate * 1e3, prop.clockRate / 1000, mean);
        hipFree(out); hipFree(clk);
    }
    return 0;
}
