#include "prof.h"

#include <vector>

#include "common.h"

namespace {
struct Rec { int kind, tag; double flops, bytes; hipEvent_t e0, e1; };
int g_tag = -1;
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
    Rec r{kind, g_tag, flops, bytes, get_event(), get_event()};
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
}
void prof_end(hipStream_t s) { (void)hipEventRecord(g_recs.back().e1, s); }

void prof_set_tag(int tag) { g_tag = tag; }

// raw records (kind, tag = unit index, ms, flops, bytes) in launch order; returns the number written
extern "C" int vs_profile_read_raw(int max_n, int* kind, int* tag, double* ms, double* flops, double* bytes) {
    int n = 0;
    for (auto& r : g_recs) {
        if (n >= max_n) break;
        if (hipEventSynchronize(r.e1) != hipSuccess) return -1;
        float t = 0;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return -1;
        kind[n] = r.kind; tag[n] = r.tag; ms[n] = t; flops[n] = r.flops; bytes[n] = r.bytes;
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
