// common.hpp — internal definitions shared by the translation units of libsrx_hip.so.
// gfx950 / CDNA4 only: wave = 64 lanes, 256 CUs in 8 XCDs, 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/srx.h"

// rccl.h is only needed by comm.hip / the all-reduce helper; keep the handle opaque here.
struct ncclComm;

namespace srx {

constexpr int kWave = 64;

// ---- device-side helpers -----------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
// index of this wave in the grid, as a wave-uniform value the compiler KNOWS is uniform: what is derived from it (row
// numbers, row pointers, row scales) lives in SGPRs and comes in by scalar loads instead of 64 identical vector loads
__device__ __forceinline__ uint64_t global_wave_id() {
    return (uint64_t)blockIdx.x * (blockDim.x / kWave) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;  // every lane holds the total
}
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        T o = __shfl_xor(v, off, kWave);
        v = o < v ? o : v;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        T o = __shfl_xor(v, off, kWave);
        v = o > v ? o : v;
    }
    return v;
}

// ---- profiling accumulators ---------------------------------------------------------------
struct ProfAcc {
    double ms = 0.0;
    uint64_t launches = 0;
    double bytes = 0.0, aux_bytes = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace srx

// ---- the opaque handles of srx.h -----------------------------------------------------------
struct srx_ctx {
    int device = 0;
    int n_cus = 256;
    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;       // collectives issued beside the compute stream (launch_gram); events for fork / join
    hipEvent_t comm_fork = nullptr, comm_join = nullptr;
    // sharded rows: the second half of the Gram kernel runs here, off the CUs left to the collective (launch_gram)
    hipStream_t gram_stream = nullptr;
    bool gram_stream_masked = false;
    uint32_t* d_gram_mode = nullptr;         // copy of the value statistics the last stripe kernel decided its mode from (srx_gram_mode_info)
    int gram_mode_state = 0;                 // 0: none (no launch / no rows on this rank), 1: f64 atomics forced, 2: decided from d_gram_mode
    bool gram_mode_f32 = false;
    uint32_t gram_splits = 0;                // Gram exchanges run in the split arrangement (srx_comm_overlap_info)
    hipEvent_t gram_fork = nullptr, gram_join = nullptr;
    std::string err;
    // RCCL (one process per GPU)
    ncclComm* comm = nullptr;
    int n_ranks = 1, rank = 0;
    // caller-supplied sum over ranks (srx_comm_init_host): an application that already has a transport (MPI, its
    // own sockets) reduces the few small f64 buffers of the path itself; also how the sharded path is tested with
    // several ranks on ONE GPU, where RCCL refuses duplicate devices
    srx_host_allreduce_fn host_allreduce = nullptr;
    void* host_allreduce_user = nullptr;
    // profiling
    uint32_t prof_mask = 0;
    srx::ProfAcc prof[SRX_K_COUNT_];
    std::vector<hipEvent_t> event_pool;
    // named scratch buffers that only ever grow (no hipMalloc inside the steady-state path)
    struct Scratch { void* p = nullptr; size_t bytes = 0; };
    std::map<std::string, Scratch> scratch;
    // Device buffers of matrices that come and go (the row tiles of a backed session: a hipMalloc / hipFree pair of ~0.6 GB
    // per array and tile) are recycled instead of returned: after some tens of such pairs a single hipMalloc took 2.5 s
    // (profiles/r02_c5_backed.json: sweep 2 at 25 instead of 63 GB/s).  `pool_on` counts the open backed sessions.
    int pool_on = 0;
    std::vector<std::pair<void*, size_t>> pool;
    // pinned host staging for small D2H/H2D blocks
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    // asynchronous read-back slots of the PCA driver (residual + status per Rayleigh–Ritz step)
    static constexpr int kAsyncSlots = 4;
    // hipGraph cache of the PCA driver: the subspace iteration is ~100 small dependent launches per pipeline
    // step; captured once per (shape, schedule, buffers) and replayed with one hipGraphLaunch per segment
    std::map<std::string, hipGraphExec_t> graphs;
    bool graphs_off = false;                 // capture failed once (or SRX_NO_GRAPH): plain launches from then on
    bool capturing = false;                  // ProfScope and friends stay out of a capture
    double* pin_async = nullptr;             // kAsyncSlots x 8 doubles, pinned
    // H2D workers of srx_matrix_upload / the backed sessions (ctx.hip): each owns a stream, two pinned staging
    // buffers and two events, and moves its own contiguous share of an array
    struct UpWorker {
        hipStream_t stream = nullptr;
        void* pin[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
    };
    std::vector<UpWorker> up_workers;
    hipStream_t direct_stream = nullptr;     // upload_on: values the caller holds in pinned memory go straight from there (no staging copy)
    hipEvent_t async_ev[kAsyncSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t d2h_ev = nullptr;        // d2h_begin / d2h_end: a small read-back the host waits for while the stream goes on
    void* pin_d2h = nullptr;            // ... its own pinned slot
    bool d2h_pending = false;
};

struct srx_pca_state {           // what the last srx_pca / srx_pipeline left in HBM
    bool valid = false;
    uint32_t k = 0, n_pc = 0;
    uint32_t rounds = 1;             // deflation rounds of the solve
    std::vector<int> round_counts;   // components resolved by each round (sums to n_pc)
    double* d_scores = nullptr;      // n_rows x n_pc, row-major f64
    size_t scores_cap = 0;
    std::vector<double> components;  // k x n_pc (host copy; small)
    std::vector<double> evr, mean, std_;
    std::vector<uint64_t> sel;
    srx_pca_info info{};
    // Deferred host copies ("results stay on the device until fetched"): the solve leaves its small results in
    // `d_small` (carved from the d_scores allocation) and the host vectors above are produced by the first
    // fetch — pca_materialize() in pca.hip.  Layout in doubles:
    //   rounds x [ V k*64 | theta 64 | sgn 64 ] | mu k | sd k | trace 1 | pad 1 | sel_rank (int32) k
    double* d_small = nullptr;
    bool host_pending = false;
    bool dev_sel = false;            // mu / sd / trace / selection are in d_small (else in the pend_* fields)
    std::vector<double> pend_mu, pend_sd;      // slot order
    double pend_trace = 0.0;
};

struct srx_mat {
    srx_ctx* ctx = nullptr;
    uint64_t n_rows = 0, n_cols = 0, nnz = 0;
    int32_t dtype = SRX_F32;   // logical dtype (DynCsrMatrix variant)
    int32_t store = SRX_STORE_F32;
    bool pooled = false;           // buffers go back to ctx->pool when the matrix is freed
    bool store_auto = false;   // SRX_STORE_AUTO at creation: normalize_total / log1p promote f32 storage to f64 where the
                               // reference's DynCsrMatrix variant becomes F64 (scale/mod.rs:74-83, transform/mod.rs:48-55)
    int64_t* d_indptr = nullptr;
    int32_t* d_indices = nullptr;
    void* d_values = nullptr;
    uint64_t row_offset = 0;
    // gene tiling for the LDS-privatised per-gene passes (built lazily, pattern-only)
    int n_tiles = 0;
    int tile_genes = 0;
    int64_t* d_tile_ptr = nullptr;  // (n_tiles-1) x n_rows absolute positions
    // 16-bit mirror of `d_indices` (n_cols <= 65536; pattern-only, built with the tiles, inherited by clones):
    // the three passes that stream the column indices of the WHOLE matrix (gene moments, HVG count / fill)
    // read 2 bytes per non-zero instead of 4
    uint16_t* d_idx16 = nullptr;
    // per-gene non-zero counts of THIS shard: pattern-only (built by the first moments pass or srx_matrix_prepare,
    // inherited by clones); with them the moments passes drop the count atomic
    uint32_t* d_cnt_pat = nullptr;
    bool cnt_pat_valid = false;
    // per-gene moment cache, keyed by the value version
    uint64_t version = 1;
    uint64_t moments_version = 0;
    uint64_t* d_cnt = nullptr;      // n_cols (global after all-reduce)
    double* d_sum = nullptr;
    double* d_sq = nullptr;
    uint64_t n_rows_global = 0;     // valid with moments
    double* d_row_sum = nullptr;    // n_rows f64, filled by the normalise pass
    // pipeline: the matrix is still RAW and the in-place normalise(lazy_target) + log1p is owed (launch_writeback, pca.hip)
    bool lazy_pending = false;
    double lazy_target = 0.0;
    // CSC storage (DynCscMatrix): the arrays above are the CSR of X^T (n_rows = n_vars, n_cols = n_obs) and the
    // entry points exchange Row and Column (csc.hip)
    bool csc = false;
    srx_pca_state pca;
};

namespace srx {

// ---- error plumbing ------------------------------------------------------------------------
extern thread_local std::string g_tls_err;

inline int32_t fail(srx_ctx* ctx, int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_tls_err = buf;
    if (ctx) ctx->err = buf;
    return code;
}

#define SRX_HIP(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return srx::fail((ctx), e__ == hipErrorOutOfMemory ? SRX_E_OOM : SRX_E_HIP,      \
                             "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),         \
                             __FILE__, __LINE__);                                            \
    } while (0)

#define SRX_TRY(expr)                          \
    do {                                       \
        int32_t rc__ = (expr);                 \
        if (rc__ != SRX_OK) return rc__;       \
    } while (0)

// ---- scratch / staging ---------------------------------------------------------------------
int32_t scratch(srx_ctx* ctx, const char* name, size_t bytes, void** out);
int32_t pinned(srx_ctx* ctx, size_t bytes, void** out);
int32_t d2h(srx_ctx* ctx, void* host, const void* dev, size_t bytes);   // via pinned, synchronises
// the same small copy in two halves: `begin` queues it (pinned staging + an event), `end` waits for THAT event only — what the
// caller queues between the two runs on the device while the host is woken
int32_t d2h_begin(srx_ctx* ctx, const void* dev, size_t bytes);
int32_t d2h_end(srx_ctx* ctx, void* host, size_t bytes);
int32_t h2d(srx_ctx* ctx, void* dev, const void* host, size_t bytes);
int32_t d2h_rows(srx_ctx* ctx, void* host, const void* dev, uint64_t rows, size_t width, size_t dev_pitch);   // strided rows -> dense

// ---- profiling -------------------------------------------------------------------------------
struct ProfScope {
    srx_ctx* ctx;
    int cls;
    hipStream_t stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // alg_bytes: the minimum data the launches must move; aux_bytes: auxiliary structures of this
    // implementation read or written besides (owner records, ...) — reported separately, never part of the roofline figure
    ProfScope(srx_ctx* c, int cls_, double alg_bytes, hipStream_t stream_ = nullptr, double aux_bytes = 0.0);
    ~ProfScope();
};

// ---- roctx ranges (rocprofiler-sdk-roctx, dlopen()ed on first use; a no-op when the library is absent) --------------
// One range per pipeline stage, so that a `rocprofv3 --marker-trace` timeline shows normalise / moments / select /
// compact / gram / iterate / transform beside the kernels.
struct Range {
    bool on;
    explicit Range(const char* name);
    ~Range();
};

// ---- cross-rank sum (RCCL) --------------------------------------------------------------------
// In-place f64 sum over all ranks on ctx->stream; no-op for a single rank.
int32_t allreduce_f64(srx_ctx* ctx, double* d_buf, size_t count);
// the same sum on another stream of the context (RCCL communicator only; comm.hip): the Gram triangle's first half is summed
// over the ranks while the second half is still being computed
int32_t allreduce_f64_on(srx_ctx* ctx, double* d_buf, size_t count, hipStream_t stream);
bool comm_is_rccl(const srx_ctx* ctx);

// ---- internal entry points shared between translation units -----------------------------------
// The fused normalise + log1p transform applied ON THE FLY by a pass that reads raw values: y = ln_1p(f64(v) * scale_r),
// scale_r = (s_r == 0) ? 0 : target / s_r from the row sums (scale/mod.rs:9-15).  `row_sum` null = identity.
struct RowXf {
    const double* row_sum = nullptr;
    double target = 0.0;
    bool write_back = false;      // moments pass only: store the transformed value in place (at the storage precision)
};
int32_t upload_on(srx_ctx* ctx, const srx_csr* h, int32_t store, hipStream_t stream, srx_mat** out);   // ctx.hip
int32_t tiles_from_idx16(srx_mat* m, hipStream_t stream);     // genes.hip: 32-bit indices + gene-tile cuts from an uploaded 16-bit mirror
hipError_t dev_malloc(srx_ctx* ctx, void** p, size_t bytes);     // ctx.hip: hipMalloc, or a recycled buffer while ctx->pool_on
void pool_clear(srx_ctx* ctx);
int32_t ensure_tiles(srx_mat* m);
int32_t promote_to_f64(srx_mat* m);                   // ctx.hip: f32 storage -> f64 storage, values unchanged
int32_t ensure_pattern_counts(srx_mat* m);            // genes.hip
// device-resident result of FeatureSelection::HighlyVariable(n) (genes.hip), consumed by the PCA driver
struct HvgDev {
    int k = 0, n_words = 0;
    int32_t* d_sel_rank = nullptr;   // k gene ids in variance-rank order (what select_features returns)
    uint32_t* d_bits = nullptr;      // n_words selection bits, then n_words prefix counts
    double *d_mu = nullptr, *d_sd = nullptr, *d_dinv = nullptr, *d_tr = nullptr, *d_trace = nullptr;   // slot order
    int* d_status = nullptr;         // bit 0: NaN variance
};
int32_t select_hvg_device(srx_mat* m, uint64_t n, int center, int scale, HvgDev& out);
int32_t ensure_moments(srx_mat* m);   // fills d_cnt/d_sum/d_sq (global) for the current values
int32_t moments_accumulate(srx_mat* m, double* d_acc, RowXf xf);     // backed mode: this tile's (cnt,sum,sumsq,N) += into d_acc
int32_t moments_install(srx_mat* m, double* d_packed);     // all-reduce d_packed and make it m's global moments
int32_t launch_row_apply(srx_mat* m, double target, hipStream_t stream);   // in-place write-back from the row sums in m->d_row_sum
int32_t launch_normalize(srx_mat* m, double target, bool do_norm, bool do_log, hipStream_t stream = nullptr,
                         bool precise = false, int wgs_per_cu = 0);
int32_t launch_row_sums(srx_mat* m);
int32_t ensure_moments_xf(srx_mat* m, RowXf xf);      // moments of the TRANSFORMED values from the raw matrix
inline void touch(srx_mat* m) { m->version++; m->pca.valid = false; }

inline bool is_f32(const srx_mat* m) { return m->store == SRX_STORE_F32; }
// direction as the STORED matrix sees it: a CSC matrix is the CSR of the transpose
inline int32_t eff_dir(const srx_mat* m, int32_t d) { return (m->csc && (d == SRX_ROW || d == SRX_COLUMN)) ? 1 - d : d; }
int32_t transpose_device(srx_mat* m, srx_mat** out);      // csc.hip: CSR of the transpose of the stored matrix
inline size_t val_bytes(const srx_mat* m) { return is_f32(m) ? 4 : 8; }

}  // namespace srx
