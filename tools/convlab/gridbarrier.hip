// What a grid-wide barrier costs on this part against a kernel boundary - the number VERDICT r3 item 3 asks for before a
// persistent multi-layer kernel is designed (one 512-thread workgroup per CU, 256 workgroups; conv -> statistics -> barrier ->
// normalise-on-load conv -> ...).  Stand-alone (no library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/convlab/gridbarrier.hip -o tools/convlab/bin/gridbarrier
// Arms, each timed with HIP events over `reps` phases and checked for correctness where data moves:
//   boundary      : `reps` dependent launches of the phase body as its own 256 x 512 kernel
//   flat          : ONE launch, phases separated by a one-counter barrier (lane 0: release fence, atomic add, relaxed sc1 poll, acquire fence)
//   xcd           : the same with per-group counters (blockIdx & 7, the observed XCD round-robin - speed only) feeding a top counter
// Phase bodies: `empty` (nothing), `handoff` (every workgroup writes KB kilobytes, after the barrier reads the KB kilobytes of the
// workgroup 37 places on - another XCD - and checks every word: a layer's activations crossing the barrier).
// Every spin is bounded: a barrier that does not complete sets an error flag and the kernel drains.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int NT = 512;
constexpr unsigned kSpinLimit = 1u << 22;

struct Sync {
    unsigned* flat;      // one counter
    unsigned* grp;       // 8 group counters, 32 words apart
    unsigned* top;       // group leaders
    unsigned* gen;       // 8 generation words, 32 words apart
    unsigned* err;
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void barrier_flat(const Sync& s, unsigned phase, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(s.flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (phase + 1) * nwg;
        unsigned spins = 0;
        while (ld_relaxed(s.flat) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit) { *s.err = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

__device__ __forceinline__ void barrier_xcd(const Sync& s, unsigned phase, unsigned nwg) {
    __syncthreads();
    const unsigned g = blockIdx.x & 7, per = nwg >> 3;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(s.grp + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        if (old == (phase + 1) * per - 1) {          // last of its group: tell the top, wait for all eight, release the group
            __hip_atomic_fetch_add(s.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_relaxed(s.top) < (phase + 1) * 8) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) { *s.err = 2; break; }
            }
            __hip_atomic_store(s.gen + g * 32, phase + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_relaxed(s.gen + g * 32) < phase + 1) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > kSpinLimit) { *s.err = 3; break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// phase body: write my slice of buffer (phase & 1), value = f(phase, word index)
__device__ __forceinline__ void body_write(uint4* buf, int words16, unsigned phase) {
    uint4* mine = buf + ((size_t)(phase & 1) * gridDim.x + blockIdx.x) * words16;
    for (int i = threadIdx.x; i < words16; i += NT) {
        const unsigned v = phase * 2654435761u + blockIdx.x * 977u + (unsigned)i;
        mine[i] = make_uint4(v, v + 1, v + 2, v + 3);
    }
}
__device__ __forceinline__ unsigned body_check(const uint4* buf, int words16, unsigned phase) {
    const unsigned src = (blockIdx.x + 37) % gridDim.x;
    const uint4* theirs = buf + ((size_t)(phase & 1) * gridDim.x + src) * words16;
    unsigned bad = 0;
    for (int i = threadIdx.x; i < words16; i += NT) {
        const unsigned v = phase * 2654435761u + src * 977u + (unsigned)i;
        const uint4 q = theirs[i];
        bad += (q.x != v) + (q.y != v + 1) + (q.z != v + 2) + (q.w != v + 3);
    }
    return bad;
}

template <int MODE>   // 0 flat, 1 xcd
__global__ __launch_bounds__(NT) void persistent_kernel(Sync s, uint4* buf, int words16, int reps, unsigned* bad_out) {
    unsigned bad = 0;
    for (int ph = 0; ph < reps; ++ph) {
        if (words16) body_write(buf, words16, ph);
        if (MODE == 0) barrier_flat(s, ph, gridDim.x); else barrier_xcd(s, ph, gridDim.x);
        if (words16) bad += body_check(buf, words16, ph);
    }
    if (bad) atomicAdd(bad_out, bad);
}

__global__ __launch_bounds__(NT) void phase_kernel(uint4* buf, int words16, unsigned phase, unsigned* bad_out) {
    unsigned bad = 0;
    if (words16 && phase) bad = body_check(buf, words16, phase - 1);      // what the previous launch wrote
    if (words16) body_write(buf, words16, phase);
    if (bad) atomicAdd(bad_out, bad);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int nwg = prop.multiProcessorCount;          // one workgroup per CU
    printf("device %s, %d CUs; %d workgroups x %d threads, %d phases per arm\n", prop.name, nwg, nwg, NT, reps);
    unsigned* words;
    CK(hipMalloc(&words, 4096 * sizeof(unsigned)));
    Sync s{words, words + 64, words + 64 + 8 * 32 + 32, words + 1024, words + 2048};
    unsigned* bad = words + 2049;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int kb : {0, 16, 64}) {
        const int words16 = kb * 1024 / 16;
        uint4* buf = nullptr;
        if (kb) CK(hipMalloc(&buf, (size_t)2 * nwg * kb * 1024));
        for (int arm = 0; arm < 3; ++arm) {
            std::vector<float> us;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(words, 0, 4096 * sizeof(unsigned)));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                if (arm == 0) {
                    for (int ph = 0; ph < reps; ++ph) hipLaunchKernelGGL(phase_kernel, dim3(nwg), dim3(NT), 0, 0, buf, words16, (unsigned)ph, bad);
                } else if (arm == 1) {
                    hipLaunchKernelGGL(persistent_kernel<0>, dim3(nwg), dim3(NT), 0, 0, s, buf, words16, reps, bad);
                } else {
                    hipLaunchKernelGGL(persistent_kernel<1>, dim3(nwg), dim3(NT), 0, 0, s, buf, words16, reps, bad);
                }
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                unsigned h[2];
                CK(hipMemcpy(h, words + 2048, 2 * sizeof(unsigned), hipMemcpyDeviceToHost));
                if (h[0] || h[1]) { printf("  FAILED: arm %d kb %d err %u bad words %u\n", arm, kb, h[0], h[1]); return 1; }
                us.push_back(ms * 1e3f / reps);
            }
            std::sort(us.begin(), us.end());
            printf("%-9s %3d KB per workgroup and phase: %.2f us per phase (median of 5; min %.2f max %.2f)\n",
                   arm == 0 ? "boundary" : (arm == 1 ? "flat" : "xcd"), kb, us[2], us[0], us[4]);
        }
        if (buf) CK(hipFree(buf));
    }
    return 0;
}
