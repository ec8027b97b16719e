// In-library kernel timing with HIP events on the launch stream (used by bench.py for the roofline block).
// Disabled by default: when off, a scope costs one branch.
#pragma once
#include <hip/hip_runtime.h>

enum ProfKind {
    PK_CONV_FWD = 0,   // conv_igemm_kernel, forward convolutions
    PK_CONV_DGRAD,     // conv_igemm_kernel fed with flipped/transposed weights (data gradient)
    PK_CONV_WGRAD,     // conv_wgrad_kernel + its slab reduction
    PK_STEM,           // stem forward / weight gradient
    PK_BN_STATS, PK_BN_APPLY, PK_BN_BWD,
    PK_POOL_MISC,      // maxpool, upsample-backward, zero stuffing, layout transforms
    PK_PREPARE,        // weight copies / BN folding
    PK_HEAD,           // segmentation head conv (fwd), bias gradient
    PK_COUNT
};

bool prof_on();
void prof_begin(int kind, double flops, double bytes, hipStream_t s);
void prof_end(hipStream_t s);
void prof_add_flops(double flops);   // more algorithmic work for the record that is open (a launch that turned out to cover two layers)
void prof_set_tag(int tag);  // attached to subsequent records (unit index)
void prof_set_variant(int v);  // kernel instantiation code of the next records (0 = n/a)
int vs_option(const char* name);
// Phase probe (tools/conv_probe.py): when set, instrumented kernels store 8 x u64 per workgroup (100 MHz timestamps of
// their phases + hardware id).  Null in normal operation.
unsigned long long* vs_probe_buffer(size_t need_wgs);  // runtime option (vs_set_option / environment VS_<NAME>)

struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(int kind, double flops, double bytes, hipStream_t st) : s(st), on(prof_on()) {
        if (on) prof_begin(kind, flops, bytes, s);
    }
    ~ProfScope() {
        if (on) prof_end(s);
    }
};
