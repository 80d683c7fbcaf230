// comm.hip -- RCCL communicator behind the C ABI (hm_comm_*): the collectives of the row-sharded update run from the
// library itself, on the context's stream, in place on the plan's own device buffers.  No PyTorch in the process.
//
// Reference: the only parallel layer of the reference is the process-pool map of notebooks/tools/utils.py:201-224 (members
// are independent, "they don't communicate back from child processes" utils.py:226-228); the reductions below are what that
// map turns into once the update's rows stay on their GPU.
//
// librccl.so.1 is opened lazily with dlopen: single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include "common.h"

struct hm_comm {
    hm_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    const char* (*GetLastError)(ncclComm_t) = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
    if (g_rccl.handle) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    HM_REQUIRE(h, "RCCL not found (librccl.so.1): %s", dlerror());
#define SYM(field, name)                                                          \
    do {                                                                          \
        *(void**)(&g_rccl.field) = dlsym(h, name);                                \
        HM_REQUIRE(g_rccl.field, "librccl.so.1 lacks the symbol %s", name);       \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(CommGetAsyncError, "ncclCommGetAsyncError");
    SYM(AllReduce, "ncclAllReduce");
    SYM(AllGather, "ncclAllGather");
    SYM(Broadcast, "ncclBroadcast");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    *(void**)(&g_rccl.GetLastError) = dlsym(h, "ncclGetLastError");  // optional
    g_rccl.handle = h;
    return 0;
}

#define HM_NCCL(c, call)                                                                                      \
    do {                                                                                                      \
        ncclResult_t r_ = (call);                                                                             \
        if (r_ != ncclSuccess) {                                                                              \
            const char* more_ = g_rccl.GetLastError ? g_rccl.GetLastError(c) : "";                            \
            hm_set_error("%s failed: %s%s%s", #call, g_rccl.GetErrorString(r_), more_ && *more_ ? " -- " : "", \
                         more_ ? more_ : "");                                                                 \
            return 3;                                                                                         \
        }                                                                                                     \
    } while (0)

int nccl_type(int dtype, ncclDataType_t* t) {
    switch (dtype) {
        case 64: *t = ncclFloat64; return 0;
        case 32: *t = ncclFloat32; return 0;
        case 1: *t = ncclInt32; return 0;
        case 8: *t = ncclUint8; return 0;
    }
    hm_set_error("hm_comm: dtype must be 64 (double), 32 (float), 1 (int32) or 8 (bytes), got %d", dtype);
    return 2;
}
}  // namespace

// librccl opened and its symbols bound, nothing else: what every rank other than 0 checks before anyone enters ncclCommInitRank
// (ncclGetUniqueId starts a bootstrap listener thread and opens a port: only the rank whose id is used may call it)
extern "C" int hm_comm_probe(void) { return rccl_load(); }

extern "C" int hm_comm_unique_id(char* id_out) {
    HM_REQUIRE(id_out, "hm_comm_unique_id: NULL argument");
    static_assert(sizeof(ncclUniqueId) == HM_COMM_ID_BYTES, "HM_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    HM_NCCL(nullptr, g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int hm_comm_create(hm_ctx* ctx, int rank, int world_size, const char* unique_id, hm_comm** out) {
    HM_REQUIRE(ctx && unique_id && out, "hm_comm_create: NULL argument");
    HM_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "hm_comm_create: rank %d outside [0,%d)", rank, world_size);
    int rc = rccl_load();
    if (rc) return rc;
    HM_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    hm_comm* c = new hm_comm();
    c->ctx = ctx; c->rank = rank; c->world = world_size;
    // RCCL 2.27 prints a version banner with printf to stdout while rank 0 initialises; callers (bench.py) own stdout
    // (one JSON line): for the duration of the call C stdout is pointed at stderr, flushed, and restored.
    fflush(stdout);
    const int saved_out = dup(STDOUT_FILENO);
    if (saved_out >= 0) (void)dup2(STDERR_FILENO, STDOUT_FILENO);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world_size, id, rank);
    fflush(stdout);
    if (saved_out >= 0) {
        (void)dup2(saved_out, STDOUT_FILENO);
        close(saved_out);
    }
    if (r != ncclSuccess) {
        const char* more = g_rccl.GetLastError ? g_rccl.GetLastError(nullptr) : "";
        hm_set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s%s%s", rank, world_size, ctx->device, g_rccl.GetErrorString(r),
                     more && *more ? " -- " : "", more ? more : "");
        delete c;
        return 3;
    }
    *out = c;
    return 0;
}

extern "C" void hm_comm_destroy(hm_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

extern "C" int hm_comm_rank(hm_comm* c) { return c ? c->rank : 0; }
extern "C" int hm_comm_world_size(hm_comm* c) { return c ? c->world : 1; }

// In-place all-reduce of n elements at device address buf, stream-ordered behind whatever the context's stream holds
// (the producing phase of an update plan) and in front of whatever follows: no host synchronisation.
extern "C" int hm_comm_all_reduce(hm_comm* c, void* buf, long long n, int dtype, int op) {
    HM_REQUIRE(c && buf && n >= 0, "hm_comm_all_reduce: bad arguments");
    HM_REQUIRE(op == HM_COMM_SUM || op == HM_COMM_MAX, "hm_comm_all_reduce: op must be HM_COMM_SUM or HM_COMM_MAX");
    ncclDataType_t t;
    int rc = nccl_type(dtype, &t);
    if (rc) return rc;
    if (n == 0) return 0;
    HM_HIP(hipSetDevice(c->ctx->device));
    HM_NCCL(c->comm, g_rccl.AllReduce(buf, buf, (size_t)n, t, op == HM_COMM_SUM ? ncclSum : ncclMax, c->comm, c->ctx->stream));
    return 0;
}

// In-place all-gather: rank r's n_per_rank elements lie at buf + r * n_per_rank (elements of `dtype`) on entry; on exit
// every rank holds all world_size blocks.
extern "C" int hm_comm_all_gather(hm_comm* c, void* buf, long long n_per_rank, int dtype) {
    HM_REQUIRE(c && buf && n_per_rank >= 0, "hm_comm_all_gather: bad arguments");
    ncclDataType_t t;
    int rc = nccl_type(dtype, &t);
    if (rc) return rc;
    if (n_per_rank == 0) return 0;
    const size_t esz = dtype == 64 ? 8 : (dtype == 8 ? 1 : 4);
    HM_HIP(hipSetDevice(c->ctx->device));
    const char* mine = (const char*)buf + (size_t)c->rank * (size_t)n_per_rank * esz;
    HM_NCCL(c->comm, g_rccl.AllGather(mine, buf, (size_t)n_per_rank, t, c->comm, c->ctx->stream));
    return 0;
}

extern "C" int hm_comm_broadcast(hm_comm* c, void* buf, long long n, int dtype, int root) {
    HM_REQUIRE(c && buf && n >= 0 && root >= 0 && root < c->world, "hm_comm_broadcast: bad arguments");
    ncclDataType_t t;
    int rc = nccl_type(dtype, &t);
    if (rc) return rc;
    if (n == 0) return 0;
    HM_HIP(hipSetDevice(c->ctx->device));
    HM_NCCL(c->comm, g_rccl.Broadcast(buf, buf, (size_t)n, t, root, c->comm, c->ctx->stream));
    return 0;
}

// Fuse the collectives issued between the two calls into one RCCL launch (ncclGroupStart / ncclGroupEnd).
extern "C" int hm_comm_group_start(hm_comm* c) {
    HM_REQUIRE(c, "hm_comm_group_start: NULL communicator");
    HM_NCCL(c->comm, g_rccl.GroupStart());
    return 0;
}

extern "C" int hm_comm_group_end(hm_comm* c) {
    HM_REQUIRE(c, "hm_comm_group_end: NULL communicator");
    HM_NCCL(c->comm, g_rccl.GroupEnd());
    return 0;
}

// Wait for everything queued on the context's stream (collectives included) and report an asynchronous RCCL error.
extern "C" int hm_comm_sync(hm_comm* c) {
    HM_REQUIRE(c, "hm_comm_sync: NULL communicator");
    HM_HIP(hipSetDevice(c->ctx->device));
    HM_HIP(hipStreamSynchronize(c->ctx->stream));
    ncclResult_t async = ncclSuccess;
    HM_NCCL(c->comm, g_rccl.CommGetAsyncError(c->comm, &async));
    HM_REQUIRE(async == ncclSuccess, "RCCL asynchronous error: %s", g_rccl.GetErrorString(async));
    return 0;
}
