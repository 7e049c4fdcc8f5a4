// comm.hip — row-shard reductions over the GPUs of one node: RCCL all-reduce over xGMI.
//
// The reference is single-process (no collectives exist); row
// sharding is new.  One process per GPU; the host broadcasts the 128-byte ncclUniqueId by
// whatever transport it has, then every cross-rank quantity on the path — per-gene
// (cnt, sum, sumsq, N), the k x l blocks Z^T Y of each subspace iteration, the l x l Gram
// matrices — is summed with ONE ncclAllReduce(f64, sum) each on the ctx stream.  These
// messages are <= ~1 MB: latency-bound on the fully connected xGMI mesh, so no bucketing.
//
// librccl is dlopen()ed on first use so that the library (and the CPU-side symbol tests)
// load without touching RCCL's initialisers.
#include <dlfcn.h>

#include "common.hpp"

namespace srx {

// Minimal RCCL surface (rccl.h types restated; ABI-stable since NCCL 2.x).
typedef struct { char internal[SRX_UNIQUE_ID_BYTES]; } rcclUniqueId;
typedef int rcclResult;
enum { rcclFloat64 = 8, rcclSum = 0 };

struct RcclApi {
    void* h = nullptr;
    rcclResult (*GetUniqueId)(rcclUniqueId*) = nullptr;
    rcclResult (*CommInitRank)(ncclComm**, int, rcclUniqueId, int) = nullptr;
    rcclResult (*CommDestroy)(ncclComm*) = nullptr;
    rcclResult (*AllReduce)(const void*, void*, size_t, int, int, ncclComm*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(rcclResult) = nullptr;
    rcclResult (*GetVersion)(int*) = nullptr;
};

static RcclApi g_rccl;

static int32_t load_rccl(srx_ctx* ctx) {
    if (g_rccl.h) return SRX_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail(ctx, SRX_E_RCCL, "cannot dlopen librccl: %s", dlerror());
    RcclApi a;
    a.h = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    a.GetVersion = (decltype(a.GetVersion))dlsym(h, "ncclGetVersion");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce)
        return fail(ctx, SRX_E_RCCL, "librccl lacks the expected nccl* symbols");
    g_rccl = a;
    return SRX_OK;
}

static const char* rccl_err(rcclResult r) {
    return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error";
}

int32_t allreduce_f64(srx_ctx* ctx, double* d_buf, size_t count) {
    if (count == 0) return SRX_OK;
    if (ctx->host_allreduce) {
        // host transport: the buffer goes down, the caller sums it over the ranks in place, it comes back
        std::vector<double> h(count);
        SRX_TRY(d2h(ctx, h.data(), d_buf, count * sizeof(double)));
        const int32_t rc = ctx->host_allreduce(ctx->host_allreduce_user, h.data(), (uint64_t)count);
        if (rc != 0) return fail(ctx, SRX_E_RCCL, "host all-reduce callback failed (%d) on %zu doubles", rc, count);
        return h2d(ctx, d_buf, h.data(), count * sizeof(double));
    }
    if (!ctx->comm) return SRX_OK;
    rcclResult r = g_rccl.AllReduce(d_buf, d_buf, count, rcclFloat64, rcclSum, ctx->comm, ctx->stream);
    if (r != 0) return fail(ctx, SRX_E_RCCL, "ncclAllReduce(f64, %zu) failed: %s", count, rccl_err(r));
    return SRX_OK;
}

bool comm_is_rccl(const srx_ctx* ctx) { return ctx->comm != nullptr && !ctx->host_allreduce; }

int32_t allreduce_f64_on(srx_ctx* ctx, double* d_buf, size_t count, hipStream_t stream) {
    if (count == 0) return SRX_OK;
    if (!comm_is_rccl(ctx)) return fail(ctx, SRX_E_ARG, "allreduce_f64_on needs an RCCL communicator");
    rcclResult r = g_rccl.AllReduce(d_buf, d_buf, count, rcclFloat64, rcclSum, ctx->comm, stream);
    if (r != 0) return fail(ctx, SRX_E_RCCL, "ncclAllReduce(f64, %zu) failed: %s", count, rccl_err(r));
    return SRX_OK;
}

}  // namespace srx

using namespace srx;

extern "C" {

int32_t srx_comm_unique_id(void* id_out_128) {
    if (!id_out_128) return fail(nullptr, SRX_E_ARG, "null id buffer");
    SRX_TRY(load_rccl(nullptr));
    rcclUniqueId id;
    memset(&id, 0, sizeof id);
    rcclResult r = g_rccl.GetUniqueId(&id);
    if (r != 0) return fail(nullptr, SRX_E_RCCL, "ncclGetUniqueId failed: %s", rccl_err(r));
    memcpy(id_out_128, &id, SRX_UNIQUE_ID_BYTES);
    return SRX_OK;
}

int32_t srx_comm_init(srx_ctx* ctx, int32_t n_ranks, int32_t rank, const void* id_128) {
    if (!ctx || !id_128 || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(ctx, SRX_E_ARG, "srx_comm_init: bad arguments");
    if (ctx->comm) return fail(ctx, SRX_E_ARG, "communicator already initialised");
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    // a 1-rank communicator is still created: the same RCCL calls then run on a single GPU, which
    // is how the collective path is exercised by the 1-GPU test-suite
    SRX_TRY(load_rccl(ctx));
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    rcclUniqueId id;
    memcpy(&id, id_128, SRX_UNIQUE_ID_BYTES);
    rcclResult r = g_rccl.CommInitRank(&ctx->comm, n_ranks, id, rank);
    if (r != 0) {
        ctx->comm = nullptr;
        ctx->n_ranks = 1;
        ctx->rank = 0;
        return fail(ctx, SRX_E_RCCL, "ncclCommInitRank(%d/%d) failed: %s", rank, n_ranks, rccl_err(r));
    }
    return SRX_OK;
}

int32_t srx_comm_init_host(srx_ctx* ctx, int32_t n_ranks, int32_t rank, srx_host_allreduce_fn fn, void* user) {
    if (!ctx || !fn || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(ctx, SRX_E_ARG, "srx_comm_init_host: bad arguments");
    if (ctx->comm || ctx->host_allreduce) return fail(ctx, SRX_E_ARG, "communicator already initialised");
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    ctx->host_allreduce = fn;
    ctx->host_allreduce_user = user;
    return SRX_OK;
}

int32_t srx_comm_info(srx_ctx* ctx, int32_t* kind_out, int32_t* n_ranks_out, int32_t* rccl_version_out,
                      int32_t* ranks_seen_out) {
    if (!ctx) return fail(nullptr, SRX_E_ARG, "null ctx");
    const int kind = ctx->comm ? 1 : (ctx->host_allreduce ? 2 : 0);
    if (kind_out) *kind_out = kind;
    if (n_ranks_out) *n_ranks_out = ctx->n_ranks;
    if (rccl_version_out) {
        int v = 0;
        if (g_rccl.h && g_rccl.GetVersion) (void)g_rccl.GetVersion(&v);
        *rccl_version_out = v;
    }
    if (ranks_seen_out) {
        double one = 1.0, *d = nullptr;
        SRX_TRY(scratch(ctx, "comm_probe", 64, (void**)&d));
        SRX_TRY(h2d(ctx, d, &one, sizeof one));
        SRX_TRY(allreduce_f64(ctx, d, 1));
        SRX_TRY(d2h(ctx, &one, d, sizeof one));
        *ranks_seen_out = (int32_t)(one + 0.5);
    }
    return SRX_OK;
}

int32_t srx_comm_overlap_info(srx_ctx* ctx, int32_t* split_exchanges_out, int32_t* cu_masked_out) {
    if (!ctx) return fail(nullptr, SRX_E_ARG, "null ctx");
    if (split_exchanges_out) *split_exchanges_out = (int32_t)ctx->gram_splits;
    if (cu_masked_out) *cu_masked_out = (ctx->gram_splits > 0 && ctx->gram_stream_masked) ? 1 : 0;
    return SRX_OK;
}

int32_t srx_comm_destroy(srx_ctx* ctx) {
    if (!ctx) return fail(nullptr, SRX_E_ARG, "null ctx");
    ctx->host_allreduce = nullptr;
    ctx->host_allreduce_user = nullptr;
    if (ctx->comm && g_rccl.CommDestroy) {
        (void)hipStreamSynchronize(ctx->stream);
        g_rccl.CommDestroy(ctx->comm);
    }
    ctx->comm = nullptr;
    ctx->n_ranks = 1;
    ctx->rank = 0;
    return SRX_OK;
}

}  // extern "C"
