// RCCL behind the C ABI (vs_comm_init / allreduce_sum / allreduce_max_u32 / broadcast): the collectives of the
// two data-parallel splits - gradient all-reduce (fp32 sum), cross-GPU merge of the packed (probability, direction, label) keys
// (uint32 max, or reduce-scatter so that every rank unpacks only its shard) and the initial weight / volume broadcast - on the
// caller's HIP stream, one communicator per rank (one process per GPU).  The reference has no multi-GPU path at all; this is
// the transport a caller without torch.distributed would use (volume-segmantics_amd/dist.py routes through it when
// VOLSEG_COMM=rccl).  librccl is opened lazily (dlopen): the library loads, and everything else works, where RCCL is absent.
#include <dlfcn.h>

#include <string>

#include "common.h"

namespace {

// the slice of rccl.h this file needs (ABI of NCCL 2.x / RCCL): opaque communicator, 128-byte unique id, enums
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclFloat32 = 7 };
enum { ncclSum = 0, ncclMax = 2 };

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

// why librccl could not be used: captured ONCE, right where the dlopen / dlsym failed (dlerror() clears itself when read)
std::string& rccl_load_error() { static std::string e; return e; }

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r.handle ? &r : nullptr;
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
        const char* e = dlerror();
        rccl_load_error() += std::string(rccl_load_error().empty() ? "" : "; ") + (e ? e : "dlopen failed");
    }
    if (!r.handle) return nullptr;
#define VS_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym))
    VS_SYM(GetUniqueId, "ncclGetUniqueId"); VS_SYM(CommInitRank, "ncclCommInitRank"); VS_SYM(CommDestroy, "ncclCommDestroy");
    VS_SYM(AllReduce, "ncclAllReduce"); VS_SYM(ReduceScatter, "ncclReduceScatter"); VS_SYM(AllGather, "ncclAllGather");
    VS_SYM(Broadcast, "ncclBroadcast"); VS_SYM(GetErrorString, "ncclGetErrorString");
#undef VS_SYM
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.ReduceScatter || !r.AllGather || !r.Broadcast) {
        rccl_load_error() = "librccl was opened but a required nccl* symbol is missing";
        dlclose(r.handle);
        r.handle = nullptr;
        return nullptr;
    }
    return &r;
}

int fail(const char* what, int code) {
    Rccl* r = rccl();
    vs_set_error("%s: RCCL error %d (%s)", what, code, (r && r->GetErrorString) ? r->GetErrorString(code) : "?");
    return VS_ERR_HIP;
}

}  // namespace

struct vs_comm {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
};

#define VS_NEED_RCCL(r)                                                                   \
    Rccl* r = rccl();                                                                     \
    if (!r) { vs_set_error("vs_comm: librccl could not be used (%s)", rccl_load_error().c_str()); return VS_ERR_UNSUPPORTED; }

extern "C" int vs_comm_unique_id(char id[128]) {
    VS_REQUIRE(id, "comm_unique_id: null pointer");
    VS_NEED_RCCL(r);
    ncclUniqueId u;
    const int rc = r->GetUniqueId(&u);
    if (rc != ncclSuccess) return fail("comm_unique_id", rc);
    memcpy(id, u.internal, 128);
    return VS_OK;
}

// one communicator per rank on the CURRENT device; id = rank 0's vs_comm_unique_id, carried to the others by the launcher
extern "C" int vs_comm_init(vs_comm_t** out, int nranks, int rank, const char id[128]) {
    VS_REQUIRE(out && id && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad arguments");
    VS_NEED_RCCL(r);
    ncclUniqueId u;
    memcpy(u.internal, id, 128);
    vs_comm* c = new vs_comm();
    c->nranks = nranks; c->rank = rank;
    const int rc = r->CommInitRank(&c->comm, nranks, u, rank);
    if (rc != ncclSuccess) { delete c; return fail("comm_init", rc); }
    *out = c;
    return VS_OK;
}

extern "C" void vs_comm_destroy(vs_comm_t* c) {
    if (!c) return;
    Rccl* r = rccl();
    if (r && c->comm) (void)r->CommDestroy(c->comm);
    delete c;
}

extern "C" int vs_comm_size(const vs_comm_t* c) { return c ? c->nranks : 0; }
extern "C" int vs_comm_rank(const vs_comm_t* c) { return c ? c->rank : -1; }

// gradients: in-place fp32 sum over ranks (the caller scales by 1 / nranks or folds it into the optimiser step)
extern "C" int vs_comm_allreduce_sum_f32(vs_comm_t* c, float* buf, int64_t n, void* stream) {
    VS_REQUIRE(c && buf && n >= 0, "comm_allreduce_sum_f32: bad arguments");
    VS_NEED_RCCL(r);
    const int rc = r->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream);
    return rc == ncclSuccess ? VS_OK : fail("comm_allreduce_sum_f32", rc);
}
// packed keys (fp16 probability << 16 | 15 - direction << 8 | label): the max IS the reference's merge
extern "C" int vs_comm_allreduce_max_u32(vs_comm_t* c, uint32_t* buf, int64_t n, void* stream) {
    VS_REQUIRE(c && buf && n >= 0, "comm_allreduce_max_u32: bad arguments");
    VS_NEED_RCCL(r);
    const int rc = r->AllReduce(buf, buf, (size_t)n, ncclUint32, ncclMax, c->comm, (hipStream_t)stream);
    return rc == ncclSuccess ? VS_OK : fail("comm_allreduce_max_u32", rc);
}
// the same merge with every rank keeping only its 1 / nranks of the voxels (send: nranks * n_per_rank keys, recv: n_per_rank)
extern "C" int vs_comm_reduce_scatter_max_u32(vs_comm_t* c, const uint32_t* send, uint32_t* recv, int64_t n_per_rank, void* stream) {
    VS_REQUIRE(c && send && recv && n_per_rank >= 0, "comm_reduce_scatter_max_u32: bad arguments");
    VS_NEED_RCCL(r);
    const int rc = r->ReduceScatter(send, recv, (size_t)n_per_rank, ncclUint32, ncclMax, c->comm, (hipStream_t)stream);
    return rc == ncclSuccess ? VS_OK : fail("comm_reduce_scatter_max_u32", rc);
}
extern "C" int vs_comm_allgather(vs_comm_t* c, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
    VS_REQUIRE(c && send && recv && bytes_per_rank >= 0, "comm_allgather: bad arguments");
    VS_NEED_RCCL(r);
    const int rc = r->AllGather(send, recv, (size_t)bytes_per_rank, ncclUint8, c->comm, (hipStream_t)stream);
    return rc == ncclSuccess ? VS_OK : fail("comm_allgather", rc);
}
extern "C" int vs_comm_broadcast(vs_comm_t* c, void* buf, int64_t bytes, int root, void* stream) {
    VS_REQUIRE(c && buf && bytes >= 0 && root >= 0 && root < c->nranks, "comm_broadcast: bad arguments");
    VS_NEED_RCCL(r);
    const int rc = r->Broadcast(buf, buf, (size_t)bytes, ncclUint8, root, c->comm, (hipStream_t)stream);
    return rc == ncclSuccess ? VS_OK : fail("comm_broadcast", rc);
}
