// RCCL behind the C ABI (ginsim_comm_*): the ONE exchange of the Monte-Carlo path -- every rank's 28-double statistics record
// to every rank -- as an all-gather enqueued on the context's own HIP stream, right behind the on-device
// reduction that produces the record and in front of the copy into pinned host memory.  No host synchronisation, no
// framework tensor, in the hot loop.  The reference has no counterpart (gnss_ins_sim/sim/ins_sim.py:490 is a serial loop).
//
// librccl is resolved with dlopen / dlsym on first use: a copy that is already in the process (PyTorch ships one) is reused,
// otherwise the ROCm one is loaded; nothing of RCCL is touched by single-GPU runs.
#include "comm.hpp"

#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

namespace ginsim {

namespace {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2: /opt/rocm/include/rccl/rccl.h:40-43, 187, 220, 260, 467, 678)
typedef struct { char internal[128]; } ncclUniqueId;
typedef void* ncclComm_t;
typedef int ncclResult_t;
constexpr int kNcclDouble = 8;      // ncclFloat64 / ncclDouble

struct Api {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;           // optional: what the communicator itself says
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char err[256] = "";
};

Api& api() {
    static Api a = [] {
        Api r;
        void* h = nullptr;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)         // a copy already mapped into the process first (same communicator library as the host framework)
            if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        for (const char* n : names) {
            if (h) break;
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!h) {
            snprintf(r.err, sizeof(r.err), "librccl not found: %s", dlerror());
            return r;
        }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(h, "ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
        r.CommCuDevice = reinterpret_cast<decltype(r.CommCuDevice)>(dlsym(h, "ncclCommCuDevice"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
        if (!r.ok) snprintf(r.err, sizeof(r.err), "librccl lacks a symbol of the NCCL 2 API");
        return r;
    }();
    return a;
}

thread_local char g_msg[320];
const char* fail(const char* what, ncclResult_t rc) {
    snprintf(g_msg, sizeof(g_msg), "%s: %s", what, api().GetErrorString ? api().GetErrorString(rc) : "RCCL error");
    return g_msg;
}

}  // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0;
};

const char* comm_probe() {
    Api& a = api();
    return a.ok ? nullptr : a.err;
}

const char* comm_unique_id(unsigned char* id128) {
    Api& a = api();
    if (!a.ok) return a.err;
    ncclUniqueId id;
    const ncclResult_t rc = a.GetUniqueId(&id);
    if (rc != 0) return fail("ncclGetUniqueId", rc);
    memcpy(id128, id.internal, 128);
    return nullptr;
}

const char* comm_create(int nranks, int rank, const unsigned char* id128, Comm** out) {
    Api& a = api();
    if (!a.ok) return a.err;
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    Comm* c = new Comm();
    c->nranks = nranks;
    c->rank = rank;
    const ncclResult_t rc = a.CommInitRank(&c->comm, nranks, id, rank);
    if (rc != 0) {
        delete c;
        return fail("ncclCommInitRank", rc);
    }
    *out = c;
    return nullptr;
}

void comm_destroy(Comm* c) {
    if (!c) return;
    if (c->comm && api().ok) (void)api().CommDestroy(c->comm);
    delete c;
}

int comm_nranks(const Comm* c) { return c ? c->nranks : 0; }

// what the COMMUNICATOR answers (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), not what it was asked for: -1 where the
// library lacks the query
const char* comm_query(const Comm* c, int* nranks, int* rank, int* device) {
    *nranks = *rank = *device = -1;
    if (!c || !c->comm) return "no communicator";
    Api& a = api();
    ncclResult_t rc = 0;
    if (a.CommCount && (rc = a.CommCount(c->comm, nranks)) != 0) return fail("ncclCommCount", rc);
    if (a.CommUserRank && (rc = a.CommUserRank(c->comm, rank)) != 0) return fail("ncclCommUserRank", rc);
    if (a.CommCuDevice && (rc = a.CommCuDevice(c->comm, device)) != 0) return fail("ncclCommCuDevice", rc);
    return nullptr;
}

const char* comm_allgather_f64(Comm* c, const double* send, double* recv, size_t count, hipStream_t stream) {
    const ncclResult_t rc = api().AllGather(send, recv, count, kNcclDouble, c->comm, stream);
    return rc != 0 ? fail("ncclAllGather", rc) : nullptr;
}

}  // namespace ginsim
