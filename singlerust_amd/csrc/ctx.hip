// ctx.hip — context, scratch memory, profiling hooks and the device-resident CSR
// (upload / alloc / download) of libsrx_hip.so.
//
// Device layout of X (SURVEY.md §8 "device layout"): int64 row offsets, int32 column
// indices, f32 or f64 values — the narrowing of the reference's usize indices
// (nalgebra_sparse::CsrMatrix) happens once, here.
#include "common.hpp"
#include <chrono>

#include <dlfcn.h>

#include <algorithm>
#include <thread>

namespace srx {

thread_local std::string g_tls_err;

int32_t scratch(srx_ctx* ctx, const char* name, size_t bytes, void** out) {
    auto& s = ctx->scratch[name];
    if (s.bytes < bytes) {
        if (s.p) SRX_HIP(ctx, hipFree(s.p));
        s.p = nullptr;
        s.bytes = 0;
        size_t want = bytes + bytes / 8 + 256;
        SRX_HIP(ctx, hipMalloc(&s.p, want));
        s.bytes = want;
    }
    *out = s.p;
    return SRX_OK;
}

int32_t pinned(srx_ctx* ctx, size_t bytes, void** out) {
    if (ctx->pinned_bytes < bytes) {
        if (ctx->pinned) SRX_HIP(ctx, hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
        size_t want = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 4;
        SRX_HIP(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
        ctx->pinned_bytes = want;
    }
    *out = ctx->pinned;
    return SRX_OK;
}

static int32_t parallel_h2d(srx_ctx* ctx, const void* src, void* dst, uint64_t n, bool narrow, size_t elem_bytes,
                            uint64_t n_cols, bool* bad);
static int32_t parallel_d2h(srx_ctx* ctx, void* dst, const void* src, size_t bytes);
static int32_t parallel_d2h_rows(srx_ctx* ctx, void* dst, const void* src, uint64_t rows, size_t width, size_t dev_pitch);

// Small/medium blocks go through the pinned staging buffer so the copy is a true async DMA
// ordered on the ctx stream; large blocks (values of the whole matrix, the score matrix) go through the
// transfer workers (several pinned double-buffer pipelines side by side).
int32_t d2h(srx_ctx* ctx, void* host, const void* dev, size_t bytes) {
    if (bytes == 0) return SRX_OK;
    if (bytes <= (64u << 20)) {
        void* p;
        SRX_TRY(pinned(ctx, bytes, &p));
        SRX_HIP(ctx, hipMemcpyAsync(p, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(host, p, bytes);
    } else {
        SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SRX_TRY(parallel_d2h(ctx, host, dev, bytes));
    }
    return SRX_OK;
}

// `rows` rows of `width` bytes, `dev_pitch` bytes apart on the device, to a dense host matrix
// A small read-back the host waits for while the stream goes on.  Its own pinned slot (it used to stage through
// ctx->pinned, which any d2h() in between overwrites and any larger pinned() request frees) and a pending flag: one at a time.
constexpr size_t kSplitReadBytes = 256;
int32_t d2h_begin(srx_ctx* ctx, const void* dev, size_t bytes) {
    if (bytes > kSplitReadBytes) return fail(ctx, SRX_E_ARG, "d2h_begin: %zu bytes exceed the split read-back slot", bytes);
    if (ctx->d2h_pending) {              // (an error return between a begin and its end: the abandoned copy is waited out)
        SRX_HIP(ctx, hipEventSynchronize(ctx->d2h_ev));
        ctx->d2h_pending = false;
    }
    if (!ctx->pin_d2h) SRX_HIP(ctx, hipHostMalloc(&ctx->pin_d2h, kSplitReadBytes));
    if (!ctx->d2h_ev) SRX_HIP(ctx, hipEventCreateWithFlags(&ctx->d2h_ev, hipEventDisableTiming));
    SRX_HIP(ctx, hipMemcpyAsync(ctx->pin_d2h, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SRX_HIP(ctx, hipEventRecord(ctx->d2h_ev, ctx->stream));
    ctx->d2h_pending = true;
    return SRX_OK;
}
int32_t d2h_end(srx_ctx* ctx, void* host, size_t bytes) {
    if (!ctx->d2h_pending) return fail(ctx, SRX_E_ARG, "d2h_end without d2h_begin");
    ctx->d2h_pending = false;
    SRX_HIP(ctx, hipEventSynchronize(ctx->d2h_ev));
    memcpy(host, ctx->pin_d2h, bytes);
    return SRX_OK;
}

int32_t d2h_rows(srx_ctx* ctx, void* host, const void* dev, uint64_t rows, size_t width, size_t dev_pitch) {
    if (dev_pitch == width) return d2h(ctx, host, dev, rows * width);
    if (rows == 0 || width == 0) return SRX_OK;
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return parallel_d2h_rows(ctx, host, dev, rows, width, dev_pitch);
}

int32_t h2d(srx_ctx* ctx, void* dev, const void* host, size_t bytes) {
    if (bytes == 0) return SRX_OK;
    if (bytes <= (64u << 20)) {
        void* p;
        // the staging buffer may still feed an earlier async copy
        SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SRX_TRY(pinned(ctx, bytes, &p));
        memcpy(p, host, bytes);
        SRX_HIP(ctx, hipMemcpyAsync(dev, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    } else {
        SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SRX_TRY(parallel_h2d(ctx, host, dev, bytes, false, 1, 0, nullptr));
    }
    return SRX_OK;
}

// ---- roctx ---------------------------------------------------------------------------------------
namespace {
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
struct Roctx {
    roctx_push_fn push = nullptr;
    roctx_pop_fn pop = nullptr;
    Roctx() {
        if (getenv("SRX_NO_ROCTX")) return;
        void* h = nullptr;
        for (const char* n : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return;
        push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
        pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
Roctx& roctx() {
    static Roctx r;
    return r;
}
}  // namespace
Range::Range(const char* name) : on(false) {
    Roctx& r = roctx();
    if (r.push) {
        r.push(name);
        on = true;
    }
}
Range::~Range() {
    if (on) roctx().pop();
}

// ---- profiling -------------------------------------------------------------------------------
static hipEvent_t take_event(srx_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(srx_ctx* c, int cls_, double alg_bytes, hipStream_t stream_, double aux_bytes) : ctx(c), cls(cls_), stream(stream_ ? stream_ : c->stream) {
    if (!(ctx->prof_mask & (1u << cls)) || ctx->capturing) return;      // no event nodes inside a captured graph
    e0 = take_event(ctx);
    e1 = take_event(ctx);
    if (!e0 || !e1) { e0 = e1 = nullptr; return; }
    ctx->prof[cls].bytes += alg_bytes;
    ctx->prof[cls].aux_bytes += aux_bytes;
    ctx->prof[cls].launches += 1;
    (void)hipEventRecord(e0, stream);
}
ProfScope::~ProfScope() {
    if (!e0) return;
    (void)hipEventRecord(e1, stream);
    ctx->prof[cls].pending.emplace_back(e0, e1);
}

static int32_t prof_drain(srx_ctx* ctx) {
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) SRX_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));
    if (ctx->gram_stream) SRX_HIP(ctx, hipStreamSynchronize(ctx->gram_stream));
    for (int c = 0; c < SRX_K_COUNT_; ++c) {
        for (auto& pr : ctx->prof[c].pending) {
            float ms = 0.f;
            SRX_HIP(ctx, hipEventElapsedTime(&ms, pr.first, pr.second));
            ctx->prof[c].ms += ms;
            ctx->event_pool.push_back(pr.first);
            ctx->event_pool.push_back(pr.second);
        }
        ctx->prof[c].pending.clear();
    }
    return SRX_OK;
}

// ---- upload kernels ------------------------------------------------------------------------------
__global__ void k_narrow_indices(const uint64_t* __restrict__ src, int32_t* __restrict__ dst,
                                 uint64_t n, uint64_t n_cols, int* __restrict__ flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (; i < n; i += stride) {
        uint64_t c = src[i];
        bad |= (c >= n_cols);
        dst[i] = (int32_t)c;
    }
    if (bad) atomicOr(flag, 1);
}

template <typename S, typename D>
__global__ void k_convert_values(const S* __restrict__ src, D* __restrict__ dst, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (D)src[i];
}

// One wave per row: column indices strictly increasing inside the row (canonical CSR; the
// reference's CsrNonCanonical arm is todo!(), src/shared/statistics/mod.rs:11) and row
// offsets monotone.
__global__ void k_validate_rows(const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx,
                                uint64_t n_rows, uint64_t nnz, int* __restrict__ flag) {
    uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    int lane = lane_id();
    bool bad = false;
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        int64_t lo = indptr[r], hi = indptr[r + 1];
        if (lo > hi || lo < 0 || (uint64_t)hi > nnz) { bad = true; continue; }
        for (int64_t p = lo + lane; p + 1 < hi; p += kWave) bad |= !(idx[p] < idx[p + 1]);
    }
    if (bad) atomicOr(flag, 2);
}

static int grid_for(uint64_t n, int block, int cap) {
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (uint64_t)cap) g = cap;
    return (int)g;
}

template <typename S>
static int32_t convert_chunked(srx_ctx* ctx, hipStream_t stream, const char* tmp_tag, const void* host, void* d_dst,
                               uint64_t n, bool to_f32) {
    const uint64_t chunk = 32ull << 20;  // elements per staging chunk
    void* d_tmp;
    SRX_TRY(scratch(ctx, tmp_tag, (n < chunk ? n : chunk) * sizeof(S) + 16, &d_tmp));
    for (uint64_t off = 0; off < n; off += chunk) {
        uint64_t cnt = n - off < chunk ? n - off : chunk;
        SRX_HIP(ctx, hipMemcpy(d_tmp, (const S*)host + off, cnt * sizeof(S), hipMemcpyHostToDevice));
        int g = grid_for(cnt, 256, 4096);
        if (to_f32)
            hipLaunchKernelGGL((k_convert_values<S, float>), dim3(g), dim3(256), 0, stream,
                               (const S*)d_tmp, (float*)d_dst + off, cnt);
        else
            hipLaunchKernelGGL((k_convert_values<S, double>), dim3(g), dim3(256), 0, stream,
                               (const S*)d_tmp, (double*)d_dst + off, cnt);
        SRX_HIP(ctx, hipStreamSynchronize(stream));
    }
    return SRX_OK;
}

// hipMalloc, or — while a backed session is open on the context — a recycled buffer of about the size (best fit among those
// at most a quarter larger; fresh buffers of 1 MiB and more get 3 % of headroom so that the next tile's slightly different
// size still fits).  The pool only holds what a session's tiles give back: it is capped (kPoolCapBytes, oldest buffers freed
// first), and when the device runs out of memory it is emptied and the allocation tried again.
constexpr size_t kPoolCapBytes = 24ull << 30;
static size_t pool_bytes(const srx_ctx* ctx) {
    size_t t = 0;
    for (auto& b : ctx->pool) t += b.second;
    return t;
}
hipError_t dev_malloc(srx_ctx* ctx, void** p, size_t bytes) {
    if (ctx->pool_on) {
        int best = -1;
        for (size_t i = 0; i < ctx->pool.size(); ++i) {
            const size_t cap = ctx->pool[i].second;
            if (cap >= bytes && cap <= bytes + bytes / 4 + (1u << 20) && (best < 0 || cap < ctx->pool[(size_t)best].second)) best = (int)i;
        }
        if (best >= 0) {
            *p = ctx->pool[(size_t)best].first;
            ctx->pool.erase(ctx->pool.begin() + best);
            return hipSuccess;
        }
    }
    const size_t want = (ctx->pool_on && bytes >= (1u << 20)) ? bytes + bytes / 32 + 65536 : bytes;
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess && !ctx->pool.empty()) {         // out of memory with recycled buffers parked: give them back, try again
        (void)hipGetLastError();
        pool_clear(ctx);
        e = hipMalloc(p, want);
    }
    return e;
}
static void dev_release(srx_mat* m, void* p) {
    if (!p) return;
    srx_ctx* ctx = m->ctx;
    size_t cap = 0;
    // recycled only while a session is open NOW (a matrix made during a session and freed after it frees its buffers)
    if (m->pooled && ctx && ctx->pool_on > 0 && hipMemPtrGetInfo(p, &cap) == hipSuccess && cap > 0) {
        ctx->pool.emplace_back(p, cap);
        while (ctx->pool.size() > 1 && pool_bytes(ctx) > kPoolCapBytes) {       // oldest first
            (void)hipFree(ctx->pool.front().first);
            ctx->pool.erase(ctx->pool.begin());
        }
    } else {
        (void)hipFree(p);
    }
}
void pool_clear(srx_ctx* ctx) {
    for (auto& b : ctx->pool) (void)hipFree(b.first);
    ctx->pool.clear();
}

static void free_mat_buffers(srx_mat* m) {
    if (!m) return;
    dev_release(m, m->d_indptr);
    dev_release(m, m->d_indices);
    dev_release(m, m->d_values);
    dev_release(m, m->d_tile_ptr);
    dev_release(m, m->d_idx16);
    dev_release(m, m->d_cnt);
    dev_release(m, m->d_cnt_pat);
    dev_release(m, m->d_sum);
    dev_release(m, m->d_sq);
    dev_release(m, m->d_row_sum);
    (void)hipFree(m->pca.d_scores);
}

static int32_t resolve_store(int32_t dtype, int32_t store) {
    if (store == SRX_STORE_F32 || store == SRX_STORE_F64) return store;
    switch (dtype) {
        case SRX_I32: case SRX_U32: case SRX_F64: return SRX_STORE_F64;
        default: return SRX_STORE_F32;
    }
}


__global__ void k_rebase_indptr(int64_t* __restrict__ indptr, uint64_t n, int64_t base) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) indptr[i] -= base;
}

// ---- host -> device in parallel ----------------------------------------------------------------------------
// A pageable hipMemcpy moves ~26-31 GB/s on this box: one thread copies into the runtime's staging buffer while the
// DMA engine waits.  Here kUpWorkers (8; SRX_UP_WORKERS) threads each take a contiguous share of the array and run their own two-buffer
// pipeline (fill a pinned buffer — narrowing u64 indices to i32 and bounds-checking them on the way, so only half
// the index bytes cross PCIe — then hipMemcpyAsync on the worker's stream while the other buffer is being filled).
static const int kUpWorkers = [] {
    const char* e = getenv("SRX_UP_WORKERS");          // tuning switch
    const int v = e ? atoi(e) : 8;                   // 4: 40, 8: 55, 16: 39-50, 32: 33 GB/s of host data (400k x 20k matrix)
    return v < 1 ? 1 : v > 64 ? 64 : v;
}();
constexpr size_t kUpChunkBytes = 4u << 20;          // per staging buffer (pinned: workers x 2 x 4 MiB)

static int32_t ensure_up_workers(srx_ctx* ctx) {
    if (!ctx->up_workers.empty()) return SRX_OK;
    std::vector<srx_ctx::UpWorker> ws(kUpWorkers);
    for (auto& w : ws) {
        SRX_HIP(ctx, hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            SRX_HIP(ctx, hipHostMalloc(&w.pin[b], kUpChunkBytes, hipHostMallocDefault));
            SRX_HIP(ctx, hipEventCreateWithFlags(&w.ev[b], hipEventDisableTiming));
        }
    }
    ctx->up_workers = std::move(ws);
    return SRX_OK;
}

// `n` elements from host `src` to device `dst`.  narrow: src is u64[n], dst i32[n] (elem_bytes 4) or u16[n] (elem_bytes 2),
// values >= n_cols set *bad; otherwise a plain copy of n * elem_bytes bytes.
static int32_t parallel_h2d(srx_ctx* ctx, const void* src, void* dst, uint64_t n, bool narrow, size_t elem_bytes,
                            uint64_t n_cols, bool* bad) {
    if (n == 0) return SRX_OK;
    SRX_TRY(ensure_up_workers(ctx));
    const int device = ctx->device;
    const uint64_t per_chunk = kUpChunkBytes / elem_bytes;
    // small arrays: fewer workers (a share should be worth a thread)
    int nw = (int)std::min<uint64_t>(kUpWorkers, (n + per_chunk - 1) / per_chunk);
    if (nw < 1) nw = 1;
    const uint64_t share = ((n + nw - 1) / nw + 15) & ~(uint64_t)15;
    std::vector<hipError_t> err(nw, hipSuccess);
    std::vector<int> oob(nw, 0);
    auto run = [&](int t) {
        srx_ctx::UpWorker& w = ctx->up_workers[t];
        hipError_t e = hipSetDevice(device);
        const uint64_t e0 = std::min<uint64_t>(n, (uint64_t)t * share), e1 = std::min<uint64_t>(n, e0 + share);
        int buf = 0;
        bool used[2] = {false, false};
        int bad_local = 0;
        for (uint64_t off = e0; off < e1 && e == hipSuccess; off += per_chunk, buf ^= 1) {
            const uint64_t cnt = std::min<uint64_t>(per_chunk, e1 - off);
            if (used[buf]) e = hipEventSynchronize(w.ev[buf]);          // the DMA out of this buffer has finished
            if (e != hipSuccess) break;
            if (narrow && elem_bytes == sizeof(uint16_t)) {
                const uint64_t* s = static_cast<const uint64_t*>(src) + off;
                uint16_t* d = static_cast<uint16_t*>(w.pin[buf]);
                uint64_t over = 0;
                for (uint64_t i = 0; i < cnt; ++i) {
                    const uint64_t v = s[i];
                    over |= (uint64_t)(v >= n_cols);
                    d[i] = (uint16_t)v;
                }
                bad_local |= (int)over;
            } else if (narrow) {
                const uint64_t* s = static_cast<const uint64_t*>(src) + off;
                int32_t* d = static_cast<int32_t*>(w.pin[buf]);
                uint64_t over = 0;
                for (uint64_t i = 0; i < cnt; ++i) {
                    const uint64_t v = s[i];
                    over |= (uint64_t)(v >= n_cols);
                    d[i] = (int32_t)v;
                }
                bad_local |= (int)over;
            } else {
                memcpy(w.pin[buf], static_cast<const char*>(src) + off * elem_bytes, cnt * elem_bytes);
            }
            e = hipMemcpyAsync(static_cast<char*>(dst) + off * elem_bytes, w.pin[buf], cnt * elem_bytes, hipMemcpyHostToDevice,
                               w.stream);
            if (e == hipSuccess) e = hipEventRecord(w.ev[buf], w.stream);
            used[buf] = true;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(w.stream);
        err[t] = e;
        oob[t] = bad_local;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nw; ++t) th.emplace_back(run, t);
    run(0);
    for (auto& x : th) x.join();
    for (int t = 0; t < nw; ++t) {
        if (err[t] != hipSuccess) return fail(ctx, SRX_E_HIP, "H2D worker %d: %s", t, hipGetErrorString(err[t]));
        if (bad && oob[t]) *bad = true;
    }
    return SRX_OK;
}

// the same in the other direction: DMA into a pinned buffer, then a memcpy out of it while the next DMA runs
static int32_t parallel_d2h(srx_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return SRX_OK;
    SRX_TRY(ensure_up_workers(ctx));
    const int device = ctx->device;
    int nw = (int)std::min<uint64_t>(kUpWorkers, (bytes + kUpChunkBytes - 1) / kUpChunkBytes);
    if (nw < 1) nw = 1;
    const uint64_t share = ((bytes + nw - 1) / nw + 255) & ~(uint64_t)255;
    std::vector<hipError_t> err(nw, hipSuccess);
    auto run = [&](int t) {
        srx_ctx::UpWorker& w = ctx->up_workers[t];
        hipError_t e = hipSetDevice(device);
        const uint64_t b0 = std::min<uint64_t>(bytes, (uint64_t)t * share), b1 = std::min<uint64_t>(bytes, b0 + share);
        // chunk c is DMAed into buffer c & 1 while chunk c - 1 is copied out of the other one
        uint64_t prev_off = 0, prev_cnt = 0;
        int buf = 0;
        for (uint64_t off = b0; off < b1 && e == hipSuccess; off += kUpChunkBytes, buf ^= 1) {
            const uint64_t cnt = std::min<uint64_t>(kUpChunkBytes, b1 - off);
            e = hipMemcpyAsync(w.pin[buf], static_cast<const char*>(src) + off, cnt, hipMemcpyDeviceToHost, w.stream);
            if (e == hipSuccess) e = hipEventRecord(w.ev[buf], w.stream);
            if (prev_cnt && e == hipSuccess) {
                e = hipEventSynchronize(w.ev[buf ^ 1]);
                if (e == hipSuccess) memcpy(static_cast<char*>(dst) + prev_off, w.pin[buf ^ 1], prev_cnt);
            }
            prev_off = off;
            prev_cnt = cnt;
        }
        if (prev_cnt && e == hipSuccess) {
            e = hipEventSynchronize(w.ev[buf ^ 1]);
            if (e == hipSuccess) memcpy(static_cast<char*>(dst) + prev_off, w.pin[buf ^ 1], prev_cnt);
        }
        err[t] = e;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nw; ++t) th.emplace_back(run, t);
    run(0);
    for (auto& x : th) x.join();
    for (int t = 0; t < nw; ++t)
        if (err[t] != hipSuccess) return fail(ctx, SRX_E_HIP, "D2H worker %d: %s", t, hipGetErrorString(err[t]));
    return SRX_OK;
}

// ... and of `rows` rows of `width` bytes that lie `dev_pitch` bytes apart on the device (the score rows, padded to whole
// 128-byte pieces) into a dense host matrix: the workers take row ranges, the DMA engine does the strided read
// (hipMemcpy2DAsync into the dense pinned buffer), the memcpy out of the other buffer runs under it.
static int32_t parallel_d2h_rows(srx_ctx* ctx, void* dst, const void* src, uint64_t rows, size_t width, size_t dev_pitch) {
    if (rows == 0 || width == 0) return SRX_OK;
    if (width > kUpChunkBytes) return fail(ctx, SRX_E_ARG, "D2H: a row of %zu bytes exceeds the staging buffers", width);
    SRX_TRY(ensure_up_workers(ctx));
    const int device = ctx->device;
    const uint64_t per_chunk = kUpChunkBytes / width;                   // rows per staging buffer
    int nw = (int)std::min<uint64_t>(kUpWorkers, (rows + per_chunk - 1) / per_chunk);
    if (nw < 1) nw = 1;
    const uint64_t share = (rows + nw - 1) / nw;
    std::vector<hipError_t> err(nw, hipSuccess);
    auto run = [&](int t) {
        srx_ctx::UpWorker& w = ctx->up_workers[t];
        hipError_t e = hipSetDevice(device);
        const uint64_t r0 = std::min<uint64_t>(rows, (uint64_t)t * share), r1 = std::min<uint64_t>(rows, r0 + share);
        uint64_t prev_row = 0, prev_cnt = 0;
        int buf = 0;
        for (uint64_t r = r0; r < r1 && e == hipSuccess; r += per_chunk, buf ^= 1) {
            const uint64_t cnt = std::min<uint64_t>(per_chunk, r1 - r);
            e = hipMemcpy2DAsync(w.pin[buf], width, static_cast<const char*>(src) + r * dev_pitch, dev_pitch, width, cnt,
                                 hipMemcpyDeviceToHost, w.stream);
            if (e == hipSuccess) e = hipEventRecord(w.ev[buf], w.stream);
            if (prev_cnt && e == hipSuccess) {
                e = hipEventSynchronize(w.ev[buf ^ 1]);
                if (e == hipSuccess) memcpy(static_cast<char*>(dst) + prev_row * width, w.pin[buf ^ 1], prev_cnt * width);
            }
            prev_row = r;
            prev_cnt = cnt;
        }
        if (prev_cnt && e == hipSuccess) {
            e = hipEventSynchronize(w.ev[buf ^ 1]);
            if (e == hipSuccess) memcpy(static_cast<char*>(dst) + prev_row * width, w.pin[buf ^ 1], prev_cnt * width);
        }
        err[t] = e;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nw; ++t) th.emplace_back(run, t);
    run(0);
    for (auto& x : th) x.join();
    for (int t = 0; t < nw; ++t)
        if (err[t] != hipSuccess) return fail(ctx, SRX_E_HIP, "D2H worker %d: %s", t, hipGetErrorString(err[t]));
    return SRX_OK;
}

// H2D of a host CSR slice with the u64 -> i32 index narrowing, the value conversion and the canonical-CSR
// validation, every kernel on `stream`.  `h->indptr` may be a window of a larger row-offset array (a row tile of
// a backed matrix: indptr[0] != 0, indices / values pointing at the tile's first entry) — it is rebased on the
// device.  With a stream other than ctx->stream the staging buffers are separate ones, so a tile can be
// uploaded while the previous one is still being worked on.
int32_t upload_on(srx_ctx* ctx, const srx_csr* h, int32_t store, hipStream_t stream, srx_mat** out) {
    *out = nullptr;
    const bool side = stream != ctx->stream;
    const char* tmp_tag = side ? "upload_tmp_side" : "upload_tmp";
    const uint64_t base = h->indptr[0];
    if (h->indptr[h->n_rows] - base != h->nnz)
        return fail(ctx, SRX_E_FORMAT, "X is not a CSR matrix: row_offsets do not span nnz");
    // Every row offset is checked HERE, on the host, before anything on the device walks a row by them: the 16-bit upload path
    // (k_narrow16_tiles<WIDEN>) reads and writes indptr[r] .. indptr[r + 1] ahead of k_validate_rows, and a non-monotone or
    // out-of-range interior offset would send it outside the index arrays.  8 bytes per row: ~1 ms at 1.3M rows.
    {
        uint64_t bad = 0, prev = base;
        const uint64_t hi = base + h->nnz;
        for (uint64_t r = 1; r <= h->n_rows; ++r) {
            const uint64_t v = h->indptr[r];
            bad |= (uint64_t)(v < prev) | (uint64_t)(v > hi);
            prev = v;
        }
        if (bad) return fail(ctx, SRX_E_FORMAT, "X is not a CSR matrix: row_offsets are not monotone within [0, nnz]");
    }
    srx_mat* m = nullptr;
    constexpr bool trace = false;                 // (development: reports uploads that take > 100 ms, by phase)
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tt[6] = {0, 0, 0, 0, 0, 0};
    tt[0] = trace ? now() : 0.0;
    SRX_TRY(srx_matrix_alloc(ctx, h->n_rows, h->n_cols, h->nnz, h->dtype, store, &m));
    tt[1] = trace ? now() : 0.0;
    bool direct_values = false;          // a DMA out of the caller's pinned values is in flight on ctx->direct_stream
    auto bail = [&](int32_t rc) {
        if (direct_values) (void)hipStreamSynchronize(ctx->direct_stream);
        srx_matrix_free(m);
        return rc;
    };
    hipError_t e = hipMemcpy(m->d_indptr, h->indptr, (h->n_rows + 1) * sizeof(int64_t), hipMemcpyHostToDevice);
    tt[2] = trace ? now() : 0.0;
    if (e != hipSuccess) return bail(fail(ctx, SRX_E_HIP, "H2D indptr: %s", hipGetErrorString(e)));
    if (base)
        hipLaunchKernelGGL(k_rebase_indptr, dim3((unsigned)((h->n_rows + 256) / 256)), dim3(256), 0, stream, m->d_indptr,
                           h->n_rows + 1, (int64_t)base);

    int* d_flag;
    int32_t rc = scratch(ctx, side ? "flag_side" : "flag", 64, (void**)&d_flag);
    if (rc) return bail(rc);
    (void)hipMemsetAsync(d_flag, 0, sizeof(int), stream);

    // indices (u64 -> i32 on the way into the pinned staging buffers: 4 of the 8 bytes cross PCIe; with at most 65536 columns
    // u64 -> u16: 2 of the 8 — the 16-bit mirror the column passes walk arrives ready-made, and the 32-bit indices + the
    // gene-tile cuts of every row are made from it on the device, genes.hip) and, when the dtype is the storage type, the
    // values: the H2D workers
    bool f32 = is_f32(m);
    const bool plain_values = (h->dtype == SRX_F32 && f32) || (h->dtype == SRX_F64 && !f32);
    if (h->nnz) {
        bool bad_col = false;
        // values the caller holds in PINNED memory (hipHostMalloc / hipHostRegister: a backed reader's tile buffers) need no
        // staging copy: one DMA straight out of them, under the workers' index narrowing
        if (plain_values && h->nnz * val_bytes(m) >= (8u << 20)) {
            // (first AND last byte: a registration that covers only a prefix of the array is not good enough; and if the copy
            //  cannot be queued all the same the values go through the staging workers like pageable ones — ADVICE r4)
            hipPointerAttribute_t at, at_end;
            const char* last = static_cast<const char*>(h->values) + h->nnz * val_bytes(m) - 1;
            if (hipPointerGetAttributes(&at, h->values) == hipSuccess && at.type == hipMemoryTypeHost &&
                hipPointerGetAttributes(&at_end, last) == hipSuccess && at_end.type == hipMemoryTypeHost) {
                if (!ctx->direct_stream) e = hipStreamCreateWithFlags(&ctx->direct_stream, hipStreamNonBlocking);
                if (e == hipSuccess)
                    e = hipMemcpyAsync(m->d_values, h->values, h->nnz * val_bytes(m), hipMemcpyHostToDevice, ctx->direct_stream);
                direct_values = e == hipSuccess;
                if (!direct_values) (void)hipGetLastError();
                e = hipSuccess;
            } else {
                (void)hipGetLastError();           // a plain pointer: not an error
            }
        }
        if (h->n_cols <= 65536) {
            e = dev_malloc(ctx, (void**)&m->d_idx16, (h->nnz + 16) * sizeof(uint16_t));
            if (e != hipSuccess) return bail(fail(ctx, SRX_E_HIP, "upload: %s", hipGetErrorString(e)));
            rc = parallel_h2d(ctx, h->indices, m->d_idx16, h->nnz, true, sizeof(uint16_t), h->n_cols, &bad_col);
            if (!rc && !bad_col) rc = tiles_from_idx16(m, stream);
        } else {
            rc = parallel_h2d(ctx, h->indices, m->d_indices, h->nnz, true, sizeof(int32_t), h->n_cols, &bad_col);
        }
        if (rc) return bail(rc);
        if (bad_col) return bail(fail(ctx, SRX_E_BOUNDS, "column index out of bounds (>= n_cols = %llu)",
                                      (unsigned long long)h->n_cols));
        tt[3] = trace ? now() : 0.0;
        bool staged_values = plain_values && !direct_values;
        if (direct_values) {
            e = hipStreamSynchronize(ctx->direct_stream);
            direct_values = false;
            if (e != hipSuccess) {                 // the DMA out of the caller's pages failed: the staging workers instead
                (void)hipGetLastError();
                staged_values = true;
            }
        }
        if (staged_values) {
            rc = parallel_h2d(ctx, h->values, m->d_values, h->nnz, false, val_bytes(m), 0, nullptr);
            if (rc) return bail(rc);
        }
        tt[4] = trace ? now() : 0.0;
    }
    (void)tmp_tag;
    if (!plain_values && h->nnz) {
        switch (h->dtype) {
            case SRX_I8:  rc = convert_chunked<int8_t>(ctx, stream, tmp_tag, h->values, m->d_values, h->nnz, f32); break;
            case SRX_I16: rc = convert_chunked<int16_t>(ctx, stream, tmp_tag, h->values, m->d_values, h->nnz, f32); break;
            case SRX_I32: rc = convert_chunked<int32_t>(ctx, stream, tmp_tag, h->values, m->d_values, h->nnz, f32); break;
            case SRX_U8:  rc = convert_chunked<uint8_t>(ctx, stream, tmp_tag, h->values, m->d_values, h->nnz, f32); break;
            case SRX_U16: rc = convert_chunked<uint16_t>(ctx, stream, tmp_tag, h->values, m->d_values, h->nnz, f32); break;
            case SRX_U32: rc = convert_chunked<uint32_t>(ctx, stream, tmp_tag, h->values, m->d_values, h->nnz, f32); break;
            case SRX_F32: rc = convert_chunked<float>(ctx, stream, tmp_tag, h->values, m->d_values, h->nnz, f32); break;
            case SRX_F64: rc = convert_chunked<double>(ctx, stream, tmp_tag, h->values, m->d_values, h->nnz, f32); break;
            default: rc = fail(ctx, SRX_E_DTYPE, "dtype %d is not supported for this operation", h->dtype);
        }
        if (rc) return bail(rc);
    }
    // validate
    hipLaunchKernelGGL(k_validate_rows, dim3(grid_for(h->n_rows * kWave, 256, 8192)), dim3(256), 0, stream,
                       m->d_indptr, m->d_indices, h->n_rows, h->nnz, d_flag);
    int flag = 0;
    e = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = hipMemcpy(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return bail(fail(ctx, SRX_E_HIP, "validate CSR: %s", hipGetErrorString(e)));
    if (flag & 1) return bail(fail(ctx, SRX_E_BOUNDS, "column index out of bounds (>= n_cols = %llu)",
                                   (unsigned long long)h->n_cols));
    if (flag & 2) return bail(fail(ctx, SRX_E_FORMAT, "X is not a canonical CSR matrix (unsorted/duplicate column indices)"));
    if (trace && now() - tt[0] > 100.0)
        fprintf(stderr, "[upload] SLOW: alloc %.1f indptr %.1f indices %.1f values %.1f validate %.1f ms\n", tt[1] - tt[0], tt[2] - tt[1],
                tt[3] - tt[2], tt[4] - tt[3], now() - tt[4]);
    *out = m;
    return SRX_OK;
}

}  // namespace srx

using namespace srx;

extern "C" {

int32_t srx_abi_version(void) { return SRX_ABI_VERSION; }

int32_t srx_device_count(int32_t* n_out) {
    if (!n_out) return fail(nullptr, SRX_E_ARG, "srx_device_count: null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *n_out = 0;
        return fail(nullptr, SRX_E_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *n_out = n;
    return SRX_OK;
}

int32_t srx_ctx_create(int32_t device_id, srx_ctx** out) {
    if (!out) return fail(nullptr, SRX_E_ARG, "srx_ctx_create: null output");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, SRX_E_HIP, "no HIP device available (%s); libsrx_hip has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, SRX_E_ARG, "device_id %d out of range [0,%d)", device_id, n);
    srx_ctx* ctx = new srx_ctx();
    ctx->device = device_id;
    hipDeviceProp_t prop;
    e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;                                  // nothing else is owned yet
        return fail(nullptr, SRX_E_HIP, "srx_ctx_create: %s", hipGetErrorString(e));
    }
    ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = ctx;
    return SRX_OK;
}

void srx_ctx_destroy(srx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    srx_comm_destroy(ctx);
    for (auto& kv : ctx->scratch) (void)hipFree(kv.second.p);
    pool_clear(ctx);
    for (int c = 0; c < SRX_K_COUNT_; ++c)
        for (auto& pr : ctx->prof[c].pending) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (auto& kv : ctx->graphs) (void)hipGraphExecDestroy(kv.second);
    if (ctx->pin_async) (void)hipHostFree(ctx->pin_async);
    for (auto& uw : ctx->up_workers) {
        for (int b = 0; b < 2; ++b) {
            if (uw.pin[b]) (void)hipHostFree(uw.pin[b]);
            if (uw.ev[b]) (void)hipEventDestroy(uw.ev[b]);
        }
        if (uw.stream) (void)hipStreamDestroy(uw.stream);
    }
    if (ctx->direct_stream) (void)hipStreamDestroy(ctx->direct_stream);
    if (ctx->d2h_ev) (void)hipEventDestroy(ctx->d2h_ev);
    if (ctx->pin_d2h) (void)hipHostFree(ctx->pin_d2h);
    if (ctx->d_gram_mode) (void)hipFree(ctx->d_gram_mode);
    for (auto e : ctx->async_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    if (ctx->comm_fork) (void)hipEventDestroy(ctx->comm_fork);
    if (ctx->comm_join) (void)hipEventDestroy(ctx->comm_join);
    if (ctx->gram_stream) (void)hipStreamDestroy(ctx->gram_stream);
    if (ctx->gram_fork) (void)hipEventDestroy(ctx->gram_fork);
    if (ctx->gram_join) (void)hipEventDestroy(ctx->gram_join);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int32_t srx_ctx_synchronize(srx_ctx* ctx) {
    if (!ctx) return fail(nullptr, SRX_E_ARG, "null ctx");
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}

const char* srx_last_error(const srx_ctx* ctx) {
    if (ctx) return ctx->err.c_str();
    return g_tls_err.c_str();
}

int32_t srx_partition_rows(const uint64_t* indptr, uint64_t n_rows, int32_t n_ranks,
                           uint64_t* cut_out) {
    if (!indptr || !cut_out || n_ranks < 1) return fail(nullptr, SRX_E_ARG, "srx_partition_rows: bad args");
    // cut[r] = first row whose start offset reaches r/n_ranks of the non-zeros (binary search
    // on indptr): balances bytes walked per GPU, not row counts.
    uint64_t base = indptr[0], nnz = indptr[n_rows] - base;
    cut_out[0] = 0;
    for (int r = 1; r < n_ranks; ++r) {
        uint64_t target = base + (uint64_t)(((__uint128_t)nnz * (unsigned)r) / (unsigned)n_ranks);
        uint64_t lo = cut_out[r - 1], hi = n_rows;
        while (lo < hi) {
            uint64_t mid = lo + (hi - lo) / 2;
            if (indptr[mid] < target) lo = mid + 1; else hi = mid;
        }
        cut_out[r] = lo;
    }
    cut_out[n_ranks] = n_rows;
    return SRX_OK;
}

int32_t srx_matrix_alloc(srx_ctx* ctx, uint64_t n_rows, uint64_t n_cols, uint64_t nnz,
                         int32_t dtype, int32_t store, srx_mat** out) {
    if (!ctx || !out) return fail(ctx, SRX_E_ARG, "srx_matrix_alloc: null argument");
    *out = nullptr;
    if (dtype < SRX_I8 || dtype > SRX_F64)
        return fail(ctx, SRX_E_DTYPE, "dtype %d is not supported for this operation", dtype);
    if (n_cols >= (1ull << 31)) return fail(ctx, SRX_E_BOUNDS, "n_cols %llu exceeds int32 device indices",
                                            (unsigned long long)n_cols);
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    srx_mat* m = new srx_mat();
    m->ctx = ctx;
    m->n_rows = n_rows; m->n_cols = n_cols; m->nnz = nnz;
    m->dtype = dtype;
    m->store = resolve_store(dtype, store);
    m->store_auto = store == SRX_STORE_AUTO;
    m->n_rows_global = n_rows;
    m->pooled = ctx->pool_on > 0;
    hipError_t e;
    e = dev_malloc(ctx, (void**)&m->d_indptr, (n_rows + 1) * sizeof(int64_t));
    // +16 elements: the streaming kernels read 16-byte vectors that may run past a row's last entry
    if (e == hipSuccess) e = dev_malloc(ctx, (void**)&m->d_indices, (nnz + 16) * sizeof(int32_t));
    if (e == hipSuccess) e = dev_malloc(ctx, &m->d_values, (nnz + 16) * val_bytes(m));
    if (e == hipSuccess) e = hipMemsetAsync(m->d_indices + nnz, 0, 16 * sizeof(int32_t), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync((char*)m->d_values + nnz * val_bytes(m), 0, 16 * val_bytes(m), ctx->stream);
    if (e != hipSuccess) {
        free_mat_buffers(m);
        delete m;
        return fail(ctx, e == hipErrorOutOfMemory ? SRX_E_OOM : SRX_E_HIP, "hipMalloc for CSR failed: %s",
                    hipGetErrorString(e));
    }
    *out = m;
    return SRX_OK;
}

int32_t srx_matrix_upload(srx_ctx* ctx, const srx_csr* h, int32_t store, srx_mat** out) {
    if (!ctx || !h || !out) return fail(ctx, SRX_E_ARG, "srx_matrix_upload: null argument");
    *out = nullptr;
    if (!h->indptr || (h->nnz && (!h->indices || !h->values)))
        return fail(ctx, SRX_E_ARG, "srx_matrix_upload: null CSR slice");
    if (h->indptr[h->n_rows] - h->indptr[0] != h->nnz || h->indptr[0] != 0)
        return fail(ctx, SRX_E_FORMAT, "X is not a CSR matrix: row_offsets do not span nnz");
    return upload_on(ctx, h, store, ctx->stream, out);
}

int32_t srx_matrix_device_ptrs(srx_mat* m, void** indptr, void** indices, void** values) {
    if (!m) return fail(nullptr, SRX_E_ARG, "null matrix");
    if (indptr) *indptr = m->d_indptr;
    if (indices) *indices = m->d_indices;
    if (values) *values = m->d_values;
    touch(m);  // the caller may write through these pointers
    m->moments_version = 0;
    if (m->d_tile_ptr) { (void)hipFree(m->d_tile_ptr); m->d_tile_ptr = nullptr; }
    if (m->d_idx16) { (void)hipFree(m->d_idx16); m->d_idx16 = nullptr; }
    m->n_tiles = 0;
    m->cnt_pat_valid = false;
    return SRX_OK;
}

int32_t srx_matrix_info(const srx_mat* m, srx_mat_info* out) {
    if (!m || !out) return fail(nullptr, SRX_E_ARG, "null argument");
    out->n_rows = m->csc ? m->n_cols : m->n_rows;      // the shape of X, whatever the storage
    out->n_cols = m->csc ? m->n_rows : m->n_cols;
    out->nnz = m->nnz;
    out->dtype = m->dtype; out->store = m->store;
    out->row_offset = m->row_offset;
    out->n_rows_global = m->n_rows_global;
    return SRX_OK;
}

int32_t srx_matrix_set_shard(srx_mat* m, uint64_t row_offset) {
    if (!m) return fail(nullptr, SRX_E_ARG, "null matrix");
    m->row_offset = row_offset;
    return SRX_OK;
}

int32_t srx_matrix_download_values(srx_mat* m, void* out, int32_t dtype_out) {
    if (!m || !out) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    srx_ctx* ctx = m->ctx;
    if (dtype_out != SRX_F32 && dtype_out != SRX_F64)
        return fail(ctx, SRX_E_DTYPE, "values can be fetched as F32 or F64 only");
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    bool same = (dtype_out == SRX_F32) == is_f32(m);
    if (same) return d2h(ctx, out, m->d_values, m->nnz * val_bytes(m));
    void* d_tmp;
    size_t ob = dtype_out == SRX_F32 ? 4 : 8;
    SRX_TRY(scratch(ctx, "download_tmp", (m->nnz ? m->nnz : 1) * ob, &d_tmp));
    int g = grid_for(m->nnz, 256, 4096);
    if (is_f32(m))
        hipLaunchKernelGGL((k_convert_values<float, double>), dim3(g), dim3(256), 0, ctx->stream,
                           (const float*)m->d_values, (double*)d_tmp, m->nnz);
    else
        hipLaunchKernelGGL((k_convert_values<double, float>), dim3(g), dim3(256), 0, ctx->stream,
                           (const double*)m->d_values, (float*)d_tmp, m->nnz);
    return d2h(ctx, out, d_tmp, m->nnz * ob);
}

int32_t srx_prof_get_aux(srx_ctx* ctx, int32_t cls, double* aux_bytes) {
    if (!ctx || cls < 0 || cls >= SRX_K_COUNT_) return fail(ctx, SRX_E_ARG, "bad kernel class");
    if (aux_bytes) *aux_bytes = ctx->prof[cls].aux_bytes;
    return SRX_OK;
}

}  // extern "C"

namespace srx {
// Storage follows the reference's variant change: a matrix created with SRX_STORE_AUTO and held in f32 moves to f64 when
// an operation makes X a DynCsrMatrix::F64 (normalize_total on anything, log1p on a non-F32 matrix).  The values are
// widened exactly; everything derived from them stays valid.
int32_t promote_to_f64(srx_mat* m) {
    if (!m->store_auto || !is_f32(m)) return SRX_OK;
    srx_ctx* ctx = m->ctx;
    void* d_new = nullptr;
    SRX_HIP(ctx, hipMalloc(&d_new, (m->nnz + 16) * sizeof(double)));
    const unsigned g = (unsigned)std::min<uint64_t>((m->nnz + 16 + 255) / 256, 65535);
    hipLaunchKernelGGL((k_convert_values<float, double>), dim3(g ? g : 1), dim3(256), 0, ctx->stream, (const float*)m->d_values,
                       (double*)d_new, m->nnz + 16);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipFree(d_new);
        return fail(ctx, SRX_E_HIP, "promote_to_f64: conversion failed");
    }
    (void)hipFree(m->d_values);
    m->d_values = d_new;
    m->store = SRX_STORE_F64;
    return SRX_OK;
}
}  // namespace srx

extern "C" {

int32_t srx_matrix_clone(srx_mat* m, srx_mat** out) {
    if (!m || !out) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    srx_ctx* ctx = m->ctx;
    srx_mat* c = nullptr;
    SRX_TRY(srx_matrix_alloc(ctx, m->n_rows, m->n_cols, m->nnz, m->dtype, m->store, &c));
    c->row_offset = m->row_offset;
    c->csc = m->csc;
    c->store_auto = m->store_auto;
    hipError_t e = hipMemcpyAsync(c->d_indptr, m->d_indptr, (m->n_rows + 1) * sizeof(int64_t),
                                  hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_indices, m->d_indices, m->nnz * sizeof(int32_t),
                                            hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_values, m->d_values, m->nnz * val_bytes(m),
                                            hipMemcpyDeviceToDevice, ctx->stream);
    // pattern-only index structure (gene-tile cuts of every row) travels with the pattern
    if (e == hipSuccess && m->n_tiles > 0) {
        c->n_tiles = m->n_tiles;
        c->tile_genes = m->tile_genes;
        if (m->d_tile_ptr) {
            const size_t tb = (size_t)(m->n_tiles - 1) * (m->n_rows ? m->n_rows : 1) * sizeof(int64_t);
            e = hipMalloc((void**)&c->d_tile_ptr, tb);
            if (e == hipSuccess) e = hipMemcpyAsync(c->d_tile_ptr, m->d_tile_ptr, tb, hipMemcpyDeviceToDevice, ctx->stream);
        }
        if (e == hipSuccess && m->d_idx16) {
            const size_t ib = (m->nnz + 16) * sizeof(uint16_t);
            e = hipMalloc((void**)&c->d_idx16, ib);
            if (e == hipSuccess) e = hipMemcpyAsync(c->d_idx16, m->d_idx16, ib, hipMemcpyDeviceToDevice, ctx->stream);
        }
    }
    if (e == hipSuccess && m->cnt_pat_valid && m->d_cnt_pat) {          // pattern-only per-gene counts
        const size_t cb = (m->n_cols ? m->n_cols : 1) * sizeof(uint32_t);
        e = hipMalloc((void**)&c->d_cnt_pat, cb);
        if (e == hipSuccess) e = hipMemcpyAsync(c->d_cnt_pat, m->d_cnt_pat, cb, hipMemcpyDeviceToDevice, ctx->stream);
        c->cnt_pat_valid = e == hipSuccess;
    }
    // a reserved result block (srx_matrix_reserve_results) is part of the handle's layout: the clone gets its own
    if (e == hipSuccess && m->pca.scores_cap > 0) {
        e = hipMalloc((void**)&c->pca.d_scores, m->pca.scores_cap);
        if (e == hipSuccess) c->pca.scores_cap = m->pca.scores_cap;
        // ... and so is the per-cell sum buffer of the pipeline's first pass (8 bytes per cell: a hipMalloc of it inside the
        // step showed as 0.25 ms outside every kernel)
        if (e == hipSuccess && !c->d_row_sum) e = hipMalloc((void**)&c->d_row_sum, (c->n_rows ? c->n_rows : 1) * sizeof(double));
    }
    if (e != hipSuccess) {
        srx_matrix_free(c);
        return fail(ctx, SRX_E_HIP, "clone D2D: %s", hipGetErrorString(e));
    }
    *out = c;
    return SRX_OK;
}

int32_t srx_matrix_prepare(srx_mat* m) {
    if (!m) return fail(nullptr, SRX_E_ARG, "null matrix");
    SRX_HIP(m->ctx, hipSetDevice(m->ctx->device));
    SRX_TRY(ensure_tiles(m));
    return m->csc ? SRX_OK : ensure_pattern_counts(m);
}

int32_t srx_matrix_copy_values(srx_mat* dst, const srx_mat* src) {
    if (!dst || !src) return fail(nullptr, SRX_E_ARG, "null argument");
    srx_ctx* ctx = dst->ctx;
    if (dst->nnz != src->nnz || dst->n_rows != src->n_rows)
        return fail(ctx, SRX_E_ARG, "copy_values: pattern mismatch");
    if (dst->store != src->store) {
        // a SRX_STORE_AUTO work copy that normalize_total / log1p / srx_pipeline promoted to f64 being restored from its
        // pristine f32 source (or the other way round): the value buffer follows the source's storage again
        if (!dst->store_auto) return fail(ctx, SRX_E_ARG, "copy_values: storage mismatch (explicit storage on the destination)");
        // (an allocation, a stream drain and a free per call: steady-state loops that restore a work copy every step should
        //  create it with an explicit SRX_STORE_F32 / SRX_STORE_F64 — then this branch is never taken)
        void* d_new = nullptr;
        SRX_HIP(ctx, hipMalloc(&d_new, (src->nnz + 16) * val_bytes(src)));
        const hipError_t es = hipStreamSynchronize(ctx->stream);
        if (es != hipSuccess) {
            (void)hipFree(d_new);
            return fail(ctx, SRX_E_HIP, "copy_values: %s", hipGetErrorString(es));
        }
        (void)hipFree(dst->d_values);
        dst->d_values = d_new;
        dst->store = src->store;
        SRX_HIP(ctx, hipMemsetAsync((char*)d_new + src->nnz * val_bytes(src), 0, 16 * val_bytes(src), ctx->stream));
    }
    SRX_HIP(ctx, hipMemcpyAsync(dst->d_values, src->d_values, src->nnz * val_bytes(src),
                                hipMemcpyDeviceToDevice, ctx->stream));
    dst->dtype = src->dtype;
    touch(dst);
    return SRX_OK;
}

void srx_matrix_free(srx_mat* m) {
    if (!m) return;
    if (m->ctx) {
        (void)hipSetDevice(m->ctx->device);
        (void)hipStreamSynchronize(m->ctx->stream);
    }
    free_mat_buffers(m);
    delete m;
}

int32_t srx_prof_enable(srx_ctx* ctx, uint32_t class_mask) {
    if (!ctx) return fail(nullptr, SRX_E_ARG, "null ctx");
    ctx->prof_mask = class_mask;
    return SRX_OK;
}

int32_t srx_prof_reset(srx_ctx* ctx) {
    if (!ctx) return fail(nullptr, SRX_E_ARG, "null ctx");
    SRX_TRY(prof_drain(ctx));
    for (int c = 0; c < SRX_K_COUNT_; ++c) {
        ctx->prof[c].ms = 0.0;
        ctx->prof[c].launches = 0;
        ctx->prof[c].bytes = 0.0;
        ctx->prof[c].aux_bytes = 0.0;
    }
    return SRX_OK;
}

int32_t srx_prof_get(srx_ctx* ctx, int32_t cls, double* total_ms, uint64_t* launches, double* bytes) {
    if (!ctx || cls < 0 || cls >= SRX_K_COUNT_) return fail(ctx, SRX_E_ARG, "bad kernel class");
    SRX_TRY(prof_drain(ctx));
    if (total_ms) *total_ms = ctx->prof[cls].ms;
    if (launches) *launches = ctx->prof[cls].launches;
    if (bytes) *bytes = ctx->prof[cls].bytes;
    return SRX_OK;
}

}  // extern "C"
