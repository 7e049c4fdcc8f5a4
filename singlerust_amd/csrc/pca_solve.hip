// pca_solve.hip — the solver half of dim_red::pca_inplace on the GPU (pca.hip has the overview): the sparse products with a
// k x 64 panel (spmm.inl), the k x 64 subspace iteration on the device (iterate.inl, jacobi.inl: dense apply on the f64 matrix
// cores, CholeskyQR, Jacobi eigen-solve, Chebyshev filter, residuals; replayed from hipGraphs), the solver driver run_pca
// (deflation rounds, plans A / B, the transform) and srx_spmm (the raw operators, for kernel-level parity tests).
#include "pca_internal.hpp"

namespace srx {

#include "spmm.inl"

#include "iterate.inl"

// ---- launches ---------------------------------------------------------------------------------
template <typename VT, typename PT>
static int32_t launch_fwd(srx_ctx* ctx, const Tiled& c, const PT* P, const PT* cvec, PT* Y, double* scores = nullptr,
                          int n_pc = 0, int ld = 0) {
    // the output is either the N x 64 panel product (SpMM solver) or, for the transform, the N x n_pc f64 scores
    const double out_bytes = scores ? (double)c.n_rows * n_pc * 8.0 : (double)c.n_rows * L * sizeof(PT);
    const double bytes = (double)c.nnz * (4.0 + sizeof(VT)) + (double)((uint64_t)c.nt * c.n_rows + 1) * 8.0 + out_bytes +
                         (double)c.k * L * sizeof(PT);
    const size_t lds = (size_t)KT * L * sizeof(PT);
    constexpr int kRowsPerWg = (kFwdThreads / 16) * FwdCfg<PT>::kRows;
    const uint64_t n_blocks = (c.n_rows + kRowsPerWg - 1) / kRowsPerWg;
    const int per_cu = sizeof(PT) == 4 ? 2 : 1;                 // 64 KiB vs 128 KiB of LDS per workgroup
    uint64_t grid = (uint64_t)ctx->n_cus * per_cu;
    if (grid > n_blocks) grid = n_blocks;
    if (grid < 1) grid = 1;
    ProfScope ps(ctx, SRX_K_SPMM_FWD, bytes);
    SRX_HIP(ctx, hipFuncSetAttribute((const void*)k_spmm_fwd<VT, PT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    hipLaunchKernelGGL((k_spmm_fwd<VT, PT>), dim3((unsigned)grid), dim3(kFwdThreads), lds, ctx->stream, c.tptr,
                       (const GramPk<VT>*)c.tpk, c.n_rows, c.nt, c.k, P, cvec, Y, scores, n_pc, ld ? ld : n_pc);
    SRX_HIP(ctx, hipGetLastError());
    return SRX_OK;
}

// Forward product from the row-major records (k_spmm_rows); false when the panel slice does not fit the LDS (the caller
// falls back on the tile-major kernel).
// Lanes per row of the row-major forward kernel: the widest panel slice (4 Q columns of all k genes) that fits the LDS;
// 0 when even one lane's four columns do not (the caller falls back on the tile-major kernel).
template <typename PT>
static int fwd_rows_q(int k) {
    const size_t budget = 163840 - 64;
    for (int q = sizeof(PT) == 4 ? 4 : 2; q >= 1; q >>= 1)
        if ((size_t)k * 4 * q * sizeof(PT) <= budget) return q;
    return 0;
}
// (the row-major forward kernel takes any k: beyond the widest single slice it walks the genes in ranges)
template <typename VT, typename PT>
bool fwd_rows_fits(int) { return true; }
template bool fwd_rows_fits<float, float>(int);
template bool fwd_rows_fits<double, double>(int);
// Forward product by gene ranges of the whole panel, one quad per row (k_spmm_ranges): f32 entries and panels, rows in
// length order, 32-bit record offsets, at most 8 ranges (k <= 4096: beyond, a row's run per range is too short to fill chunks).
// NOT the default: conflict-free in the LDS and one walk of the matrix, and 0.89 ms against the column-slice kernel's 0.62 at c3
// (profiles/r06_pmc_spmm.md: 62 % of its wave-cycles waiting — the accumulators of a block's rows, held in registers across the
// range phases, leave no room for the 16 panel reads in flight the column-slice kernel hides its LDS latency with, and every
// phase starts with 128 KB of panel through one barrier).  SRX_FWD_RANGES=1 runs it (parity tests, the profile's table).
static bool fwd_ranges_on() {
    const char* e = getenv("SRX_FWD_RANGES");          // (read per launch: the parity tests switch kernels inside one process)
    return e && atoi(e) != 0;
}
template <int S, int NT, int D>
static int32_t launch_fwd_ranges(srx_ctx* ctx, const RowMajor& r, const float* P, const float* cvec, int n_cols, double* scores, float* Y,
                                 int ld) {
    constexpr int G = RgCfg<float>::kGenes;
    const size_t lds = (size_t)G * L * sizeof(float);
    const uint64_t n_slots = (r.n_rows + 15) / 16;
    uint64_t n_wg = (n_slots + (uint64_t)(NT / 64) * S - 1) / ((uint64_t)(NT / 64) * S);
    if (n_wg > (uint64_t)ctx->n_cus) n_wg = (uint64_t)ctx->n_cus;      // one workgroup per CU (the range fills the LDS)
    if (n_wg < 1) n_wg = 1;
    const double out_bytes = scores ? (double)r.n_rows * n_cols * 8.0 : (double)r.n_rows * L * sizeof(float);
    ProfScope ps(ctx, SRX_K_SPMM_FWD, (double)r.nnz * sizeof(GramPk<float>) + (double)(r.n_rows + 1) * 8.0 + out_bytes +
                                          (double)r.k * L * sizeof(float) + (r.perm ? (double)r.n_rows * 4.0 : 0.0));
    SRX_HIP(ctx, hipFuncSetAttribute((const void*)k_spmm_ranges<float, float, S, NT, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_spmm_ranges<float, float, S, NT, D>), dim3((unsigned)n_wg), dim3(NT), lds, ctx->stream, r.ptr,
                       (const GramPk<float>*)r.pk, (const uint32_t*)r.perm, r.n_rows, r.k, P, cvec, n_cols, scores, Y, ld);
    SRX_HIP(ctx, hipGetLastError());
    return SRX_OK;
}

template <typename VT, typename PT>
static int32_t launch_fwd_rows(srx_ctx* ctx, const RowMajor& r, const PT* P, const PT* cvec, int n_cols, double* scores, PT* Y,
                               int ld) {
    if constexpr (std::is_same<VT, float>::value && std::is_same<PT, float>::value) {
        if (fwd_ranges_on() && r.nnz + 256 < (1ull << 32) && r.k <= 8 * RgCfg<float>::kGenes && n_cols <= L)
            return launch_fwd_ranges<2, 1024, 2>(ctx, r, P, cvec, n_cols, scores, Y, ld);
    }
    const int Qr = fwd_rows_q<PT>(r.k);
    auto go = [&](auto qtag, auto rtag, auto cltag, int k_lo, int k_hi, int accumulate) -> int32_t {
        constexpr int Q = decltype(qtag)::value;
        constexpr bool RANGE = decltype(rtag)::value;
        constexpr int CL = decltype(cltag)::value;
        constexpr int C = CL * Q;
        const int n_slices = (n_cols + C - 1) / C;
        // (a gene's 16 f32 columns are 64 bytes, so every 16-byte read of a wave's 16 cells starts in bank 0 or 16: half of
        //  the LDS pipe's time goes to bank conflicts, profiles/r03_pmc_spmm.md.  A padded stride of 80 bytes was measured:
        //  0.746 against 0.745 ms — the multiplication is hidden behind the kernel's reads and stores either way)
        const int ldp = C;
        const size_t lds = (size_t)(k_hi - k_lo) * ldp * sizeof(PT);
        const uint64_t groups = kFwdRowsThreads / Q;
        uint64_t n_wg = (r.n_rows + groups - 1) / groups;
        // one workgroup per CU at a time (the panel slice fills the LDS), eight in a row: shorter workgroups even out the CUs
        // (c3, f32: 1 / 2 / 4 / 8 / 16 / 32 per CU: 0.73 / 0.71 / 0.70 / 0.69 / 0.68 / 0.77 ms; f64 panels 1.37 -> 1.30 at 8)
        uint64_t cap = std::max<uint64_t>(1, (uint64_t)ctx->n_cus * 8 / n_slices);
        if (cap > 8) cap &= ~(uint64_t)7;           // (whole rounds of the 8 XCDs: the slices of a row range share an L2)
        if (n_wg > cap) n_wg = cap;
        if (n_wg < 1) n_wg = 1;
        const double out_bytes = scores ? (double)r.n_rows * n_cols * 8.0 : (double)r.n_rows * L * sizeof(PT);
        ProfScope ps(ctx, SRX_K_SPMM_FWD, (double)r.nnz * sizeof(GramPk<VT>) + (double)(r.n_rows + 1) * 8.0 + out_bytes +
                                              (double)(k_hi - k_lo) * L * sizeof(PT) + (r.perm ? (double)r.n_rows * 4.0 : 0.0));
        SRX_HIP(ctx, hipFuncSetAttribute((const void*)k_spmm_rows<VT, PT, Q, RANGE, CL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_spmm_rows<VT, PT, Q, RANGE, CL>), dim3((unsigned)(n_wg * n_slices)), dim3(kFwdRowsThreads), lds, ctx->stream,
                           r.ptr, (const GramPk<VT>*)r.pk, (const uint32_t*)r.perm, r.n_rows, r.k, P, cvec, n_cols, scores, Y, ld, ldp,
                           k_lo, k_hi, accumulate, 0);
        SRX_HIP(ctx, hipGetLastError());
        return SRX_OK;
    };
    using No = std::false_type;
    using Yes = std::true_type;
    using C4 = std::integral_constant<int, 4>;
    using C5 = std::integral_constant<int, 5>;
    // five columns per lane where the wider slice fits the LDS, saves a pass over the matrix (n_pc = 50: 3 slices of 20 instead
    // of 4 of 16; 5 of 10 instead of 7 of 8 with f64 panels) and stays inside the panel's 64 columns
    constexpr int Qmax = sizeof(PT) == 4 ? 4 : 2;
    const int wide_slices = (n_cols + 5 * Qmax - 1) / (5 * Qmax), narrow_slices = (n_cols + 4 * Qmax - 1) / (4 * Qmax);
    // (f32 panels: 3 slices of 20 columns measured the same 0.64-0.70 ms as 4 of 16 — the launch is not bound by its passes, DESIGN.md 3c —
    //  with the LDS pipe 66 % busy instead of 51 (the 80-byte gene stride conflicts more) and 1.8 GB fetched instead of 1.2:
    //  the wide form is for f64 panels, 1.85 -> 1.35 ms; SRX_FWD_WIDE=1 forces it)
    const bool wide = Qr == Qmax && (size_t)r.k * 5 * Qmax * sizeof(PT) <= (size_t)163840 && wide_slices < narrow_slices &&
                      wide_slices * 5 * Qmax <= L && sizeof(PT) == 8;
    if (wide) return go(std::integral_constant<int, Qmax>{}, No{}, C5{}, 0, r.k, 0);
    if (Qr == 4) {
        if constexpr (sizeof(PT) == 4) return go(std::integral_constant<int, 4>{}, No{}, C4{}, 0, r.k, 0);
        else return SRX_E_ARG;
    }
    if (Qr == 2) return go(std::integral_constant<int, 2>{}, No{}, C4{}, 0, r.k, 0);
    if (Qr == 1) return go(std::integral_constant<int, 1>{}, No{}, C4{}, 0, r.k, 0);
    // wider than one slice of four columns: gene ranges of the widest slice, one launch each, the later ones accumulating
    const int per = (int)((163840 - 64) / (4 * sizeof(PT)));
    for (int k_lo = 0, i = 0; k_lo < r.k; k_lo += per, ++i)
        SRX_TRY(go(std::integral_constant<int, 1>{}, Yes{}, C4{}, k_lo, std::min(r.k, k_lo + per), i > 0 ? 1 : 0));
    return SRX_OK;
}

// rows ordered by their number of kept entries (k_spmm_rows); r.ptr must be complete
int32_t build_row_order(srx_ctx* ctx, RowMajor& r) {
    uint32_t* hist;
    SRX_TRY(scratch(ctx, "pca_rm_lenhist", kLenBins * sizeof(uint32_t), (void**)&hist));
    SRX_TRY(scratch(ctx, "pca_rm_perm", (r.n_rows ? r.n_rows : 1) * sizeof(uint32_t), (void**)&r.perm));
    SRX_HIP(ctx, hipMemsetAsync(hist, 0, kLenBins * sizeof(uint32_t), ctx->stream));
    const unsigned g = (unsigned)((r.n_rows + kLenRowsPerWg - 1) / kLenRowsPerWg + (r.n_rows ? 0 : 1));
    hipLaunchKernelGGL(k_len_hist, dim3(g), dim3(256), 0, ctx->stream, r.ptr, r.n_rows, hist);
    hipLaunchKernelGGL(k_len_scan, dim3(1), dim3(kLenBins), 0, ctx->stream, hist);
    hipLaunchKernelGGL(k_len_scatter, dim3(g), dim3(256), 0, ctx->stream, r.ptr, r.n_rows, hist, r.perm);
    SRX_HIP(ctx, hipGetLastError());
    return SRX_OK;
}

template <typename VT, typename YT>
static int32_t launch_t(srx_ctx* ctx, const Tiled& c, const YT* Y, double* T /* k*L + L */) {
    uint64_t want = (uint64_t)(2 * ctx->n_cus) / (uint64_t)c.nt;
    if (want < 1) want = 1;
    uint64_t by_rows = (c.n_rows + 255) / 256;
    if (by_rows < 1) by_rows = 1;
    const uint64_t n_rb = want < by_rows ? want : by_rows;
    const uint64_t rpb = (c.n_rows + n_rb - 1) / n_rb > 0 ? (c.n_rows + n_rb - 1) / n_rb : 1;
    double *part, *part_s;
    SRX_TRY(scratch(ctx, "pca_tpart", n_rb * (size_t)c.k * L * sizeof(double), (void**)&part));
    SRX_TRY(scratch(ctx, "pca_tpart_s", n_rb * L * sizeof(double), (void**)&part_s));
    const size_t lds = (size_t)KT * L * sizeof(double);
    const double bytes = (double)c.nnz * (4.0 + sizeof(VT)) + (double)((uint64_t)c.nt * c.n_rows + 1) * 8.0 +
                         (double)c.n_rows * L * sizeof(YT) + (double)c.k * L * 8.0;
    {
        ProfScope ps(ctx, SRX_K_SPMM_T, bytes);
        SRX_HIP(ctx, hipFuncSetAttribute((const void*)k_spmm_t<VT, YT, double>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_spmm_t<VT, YT, double>), dim3((unsigned)(n_rb * c.nt)), dim3(kTThreads), lds, ctx->stream,
                           c.tptr, (const GramPk<VT>*)c.tpk, c.n_rows, c.k, c.nt, rpb, Y, part, part_s);
        uint64_t tot = (uint64_t)c.k * L + L;
        hipLaunchKernelGGL((k_t_reduce<double>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, part,
                           part_s, c.k, n_rb, T);
    }
    SRX_HIP(ctx, hipGetLastError());
    return SRX_OK;
}

static uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

struct Work {                   // k x 64 f64 state, replicated per rank
    double *W, *Wp, *T, *A1, *A2, *small, *mu, *d, *gpart;
    double *dHG, *dM, *dM2, *dTheta, *dRho, *dColmax, *dSgn, *dDinv, *dXi;
};

static int32_t alloc_work(srx_ctx* ctx, int k, Work& w) {
    const size_t kl = (size_t)k * L;
    SRX_TRY(scratch(ctx, "pca_W", kl * 8, (void**)&w.W));
    SRX_TRY(scratch(ctx, "pca_Wp", kl * 8, (void**)&w.Wp));
    SRX_TRY(scratch(ctx, "pca_T", (kl + L) * 8, (void**)&w.T));
    SRX_TRY(scratch(ctx, "pca_A1", kl * 8, (void**)&w.A1));
    SRX_TRY(scratch(ctx, "pca_A2", kl * 8, (void**)&w.A2));
    SRX_TRY(scratch(ctx, "pca_small", (6 * L * L + 8 * L + 4 * 16 * 16) * 8, (void**)&w.small));
    SRX_TRY(scratch(ctx, "pca_mu", (size_t)k * 8, (void**)&w.mu));
    SRX_TRY(scratch(ctx, "pca_d", (size_t)k * 8, (void**)&w.d));
    SRX_TRY(scratch(ctx, "pca_g2part", (size_t)kGram2Blocks * 2 * L * L * 8, (void**)&w.gpart));
    w.dHG = w.small;                    // H then G, contiguous 2 x L x L
    w.dM = w.small + 2 * L * L;
    w.dM2 = w.small + 3 * L * L;
    w.dTheta = w.small + 4 * L * L;
    w.dRho = w.dTheta + L;
    w.dColmax = w.dRho + L;
    w.dSgn = w.dColmax + L;
    w.dDinv = w.dSgn + L;
    w.dXi = w.small + 6 * L * L + 8 * L;       // the CholeskyQR's inverted 16 x 16 diagonal blocks of R
    return SRX_OK;
}

static int32_t gram2(srx_ctx* ctx, const Work& w, const double* A, const double* B, int k) {
    int nb = (k + 31) / 32;                    // one 32-row slab per workgroup where the block count allows
    if (nb > kGram2Blocks) nb = kGram2Blocks;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_gram2_part, dim3(nb), dim3(1024), 0, ctx->stream, A, B, k, w.gpart);
    hipLaunchKernelGGL(k_gram2_reduce, dim3((2 * L * L + 255) / 256), dim3(256), 0, ctx->stream, w.gpart, nb, w.dHG);
    SRX_HIP(ctx, hipGetLastError());
    return SRX_OK;
}

// Run `enqueue` (kernel launches / async memsets and copies on ctx->stream, no host synchronisation, no
// allocation) through a cached hipGraph: captured the first time a key is seen, one hipGraphLaunch afterwards.
// The key must name everything the launches depend on (shapes, schedule, device pointers).  Any failure of the
// graph machinery switches the context back to plain launches for good.
template <typename Fn>
static int32_t graphed(srx_ctx* ctx, bool enable, const std::string& key, Fn&& enqueue) {
    if (!enable || ctx->graphs_off) return enqueue();
    auto it = ctx->graphs.find(key);
    if (it == ctx->graphs.end()) {
        if (ctx->graphs.size() >= 32) {                     // stale keys (scratch regrown): start over
            for (auto& kv : ctx->graphs) (void)hipGraphExecDestroy(kv.second);
            ctx->graphs.clear();
        }
        if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
            (void)hipGetLastError();
            ctx->graphs_off = true;
            return enqueue();
        }
        ctx->capturing = true;
        const int32_t rc = enqueue();
        ctx->capturing = false;
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(ctx->stream, &g);
        hipGraphExec_t ex = nullptr;
        if (rc == SRX_OK && e == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess) {
            (void)hipGraphDestroy(g);
            it = ctx->graphs.emplace(key, ex).first;
        } else {
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            ctx->graphs_off = true;
            return rc != SRX_OK ? rc : enqueue();
        }
    }
    SRX_HIP(ctx, hipGraphLaunch(it->second, ctx->stream));
    return SRX_OK;
}

// Block subspace iteration with Rayleigh–Ritz on span(W); `apply(W, Wp)` computes Wp = C W.
// On return w.A2 = W U holds the Ritz vectors (k x 64, leading n_pc columns meaningful),
// theta their Ritz values, w.dColmax the largest-|.| entry of each Ritz vector.
template <typename Apply>
static int32_t subspace_iterate(srx_ctx* ctx, const Work& w, int k, int l_act, const Resolved& o, Apply&& apply,
                                const void* apply_id, bool graphable, const int* d_status_sel, double& resid, int& iters,
                                bool& converged, bool acc_apply, bool resume = false, bool* bailed_first = nullptr) {
    // `resume`: the block is the one a one-round plan just left at its FIRST Ritz step (flat tail: `*bailed_first`), with the
    // speculative first filter step already queued behind it — W, A2 = Ritz vectors, theta, A1 = Y1, Wp = C Y1, the Ritz tail's
    // partial sums: everything this round's own start segment (start block, warm-up sweeps, first Ritz step: ~0.45 ms at k =
    // 1000) would recompute, for the same 64-column block.  Only the step's scalars are re-made for this round's n_pc.
    // `acc_apply` (the Gram solver): `apply` ACCUMULATES into a zeroed destination and w.T is free.  The applications of a sweep
    // then rotate through three scratch blocks (Wp, A1, T), and the kernels that are the last to read a block leave it zeroed
    // (the CholeskyQR's substitution: all three; the Ritz tail: Wp; the filter step: its Z) — the eight memset launches a solve
    // used to queue in front of its applications are gone (~40 us of a 1 ms iteration).  Invariants: after `orth` Wp, A1 and T
    // are zero; after a Ritz step Wp and T are zero, A1 = (C W) U, A2 = W U.
    const size_t kl = (size_t)k * L;
    const bool use_graph = graphable && !getenv("SRX_NO_GRAPH");
    const bool use_cheb = l_act > o.n_pc && !o.robust && !o.direct;      // both solvers: the filter only needs `apply`
    constexpr int kSlots = srx_ctx::kAsyncSlots, kSlotDoubles = 8;
    if (!ctx->pin_async) {
        SRX_HIP(ctx, hipHostMalloc((void**)&ctx->pin_async, kSlots * kSlotDoubles * sizeof(double), hipHostMallocDefault));
        for (auto& e : ctx->async_ev) SRX_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    int* d_status;
    double* d_res;
    // the status words and, behind them, the ONE 64 x 64 matrix the partial products of a block pair are summed into (k_gram1_part:
    // global f64 atomics; its single consumer — the Cholesky factorisation or the eigen-solve — reads it once and leaves it zeroed;
    // the memset that clears the status words at the start of a solve clears it too)
    constexpr size_t kStatusBytes = 256;
    SRX_TRY(scratch(ctx, "pca_status", kStatusBytes + (size_t)L * L * sizeof(double), (void**)&d_status));
    double* const d_h1 = reinterpret_cast<double*>(reinterpret_cast<char*>(d_status) + kStatusBytes);
    SRX_TRY(scratch(ctx, "pca_res", kSlots * kSlotDoubles * sizeof(double), (void**)&d_res));
    double* d_ritz;
    SRX_TRY(scratch(ctx, "pca_ritzpart", (size_t)kRitzBlocks * 3 * L * sizeof(double), (void**)&d_ritz));
    SRX_HIP(ctx, hipFuncSetAttribute((const void*)k_jacobi_eig2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(J2Lds)));
    // what a captured segment depends on besides its own schedule: shapes, options, every buffer it touches
    char key0[320];
    snprintf(key0, sizeof key0, "k%d l%d p%d w%d r%d n%d s%llu t%a|%p %p %p %p %p %p %p %p %p %p", k, l_act, o.power, o.warm,
             (o.robust ? 1 : 0) + (o.direct ? 2 : 0) + (acc_apply ? 4 : 0), o.n_pc,
             (unsigned long long)o.seed, o.tol, apply_id, (void*)w.W, (void*)w.Wp, (void*)w.A1, (void*)w.A2, (void*)w.small,
             (void*)w.gpart, (void*)d_status, (void*)d_status_sel, (void*)w.T);       // (d_ritz, d_res: allocated with d_status, never regrown)
    const std::string key_base(key0);

    // Everything below only ENQUEUES work: the l x l factorisations run on the device, and the one
    // number the host needs per Rayleigh–Ritz step (the residual) comes back through a pinned slot
    // and an event, read one step late so that the stream never drains.

    // orthonormalise src -> W  (CholeskyQR: G = src^T src = R^T R, W = src R^-1; src == W is fine: every
    // thread of the substitution owns one row); G lands in dHG + L*L
    auto gram1 = [&](const double* A, const double* B) -> int32_t {       // A^T B summed into d_h1 (zero between uses)
        int nb = (k + 31) / 32;
        if (nb > kGram1Blocks) nb = kGram1Blocks;
        hipLaunchKernelGGL(k_gram1_part, dim3(nb), dim3(1024), 0, ctx->stream, A, B, k, d_h1);
        SRX_HIP(ctx, hipGetLastError());
        return 1;
    };
    auto orth = [&](const double* src) -> int32_t {
        if (o.robust) {
            SRX_TRY(gram2(ctx, w, src, src, k));
            hipLaunchKernelGGL(k_chol_factor, dim3(1), dim3(1024), 0, ctx->stream, w.dHG + L * L, l_act, w.dM, w.dDinv, d_status, 1);
        } else {
            const int32_t nb = gram1(src, src);
            if (nb < 0) return nb;
            hipLaunchKernelGGL(k_chol_factor_mfma, dim3(1), dim3(1024), 0, ctx->stream, d_h1, l_act, w.dM, w.dDinv, w.dXi, d_status);
        }
        double* const zT = acc_apply ? w.T : nullptr;           // (the matrix-free solver's apply overwrites its destination and uses T itself)
        double* const zWp = acc_apply ? w.Wp : nullptr;
        double* const zA1 = acc_apply ? w.A1 : nullptr;
        if (o.robust)
            hipLaunchKernelGGL(k_trsm_rows, dim3((k + 63) / 64), dim3(64), 0, ctx->stream, src, (const double*)w.dM, (const double*)w.dDinv, k, w.W, zWp, zA1, zT);
        else
            hipLaunchKernelGGL(k_trsm_mfma, dim3((k + 16 * kTrsmTiles - 1) / (16 * kTrsmTiles)), dim3(kTrsmTiles * 64), 0, ctx->stream, src,
                               (const double*)w.dM, (const double*)w.dXi, k, w.W, zWp, zA1, zT);
        SRX_HIP(ctx, hipGetLastError());
        if (o.robust) {                         // second pass: the first one may have run on a shifted Gram matrix
            SRX_TRY(gram2(ctx, w, w.W, w.W, k));
            hipLaunchKernelGGL(k_chol_factor, dim3(1), dim3(1024), 0, ctx->stream, w.dHG + L * L, l_act, w.dM, w.dDinv, d_status, 1);
            hipLaunchKernelGGL(k_trsm_rows, dim3((k + 63) / 64), dim3(64), 0, ctx->stream, (const double*)w.W, (const double*)w.dM, (const double*)w.dDinv, k, w.W,
                               (double*)nullptr, (double*)nullptr, (double*)nullptr);
            SRX_HIP(ctx, hipGetLastError());
        }
        return SRX_OK;
    };
    // `n` applications of C starting from `src`, ping-ponging between Wp and A1 (no copies); returns where
    // the result is
    // (`after_orth`: Wp, A1 and T are zero; otherwise — after a Ritz step — Wp and T are)
    auto apply_n = [&](const double* src, int n, bool after_orth, const double** out) -> int32_t {
        const double* cur = src;
        double* const cand[3] = {w.Wp, w.A1, acc_apply ? w.T : nullptr};
        bool zero[3] = {acc_apply, acc_apply && after_orth, acc_apply};
        for (int t = 0; t < n; ++t) {
            int pick = -1;
            for (int c = 0; c < 3 && pick < 0; ++c)
                if (cand[c] && cand[c] != cur && zero[c]) pick = c;
            for (int c = 0; c < 3 && pick < 0; ++c)
                if (cand[c] && cand[c] != cur) pick = c;
            SRX_TRY(apply(cur, cand[pick], zero[pick]));
            zero[pick] = false;
            cur = cand[pick];
        }
        *out = cur;
        return SRX_OK;
    };
    // one Rayleigh–Ritz step on span(W): Wp = C W, H = W^T Wp = U diag(theta) U^T, Ritz vectors
    // A2 = W U, residuals || C v_i - theta_i v_i || in f64; slot <- (residual, status)
    // `loose`: the step after the warm-up.  Its residuals are O(1e-2) whatever the eigen-solver does (it only feeds the
    // filter's bounds and the rotated start), and the tail of the 64-column block holds clustered Ritz values that cost
    // the cyclic Jacobi two slow sweeps: it may stop at an off-diagonal norm of 1e-5 of the diagonal (7 -> 5 sweeps).  The
    // residuals are measured on the vectors actually formed, so a loosely rotated basis is judged as what it is: at the
    // default tolerances (1e-7 / 1e-9) such a step is never accepted as converged — the next, exact one decides.
    auto ritz_kernels = [&](int slot, bool loose = false, bool wp_zero = false) -> int32_t {
        SRX_TRY(apply(w.W, w.Wp, wp_zero || acc_apply));          // (Gram solver: Wp is zero after every CholeskyQR / filter step)
        {
            // H = W^T (C W) as partial sums -> eigen-solve (adds them on load) -> Ritz vectors, C x Ritz vectors, residual and
            // largest-entry partials in one pass -> the step's scalars: 4 launches (9 on the old route)
            const int32_t nb = gram1(w.W, w.Wp);
            if (nb < 0) return nb;
            // the step after the warm-up only feeds the filter's bounds and the rotated start (any invertible U spans the same
            // block): off-diagonal norm 1e-3 of the diagonal is enough — the Ritz residual it reports, 1.63e-3 at c3, is the same
            // to three digits as with 1e-5 (1.62e-3), one Jacobi sweep less; at 1e-2 it reads 1.7e-2 and the filter takes a degree more
            constexpr double loose_tol2 = 1e-6;
            // an exact step: the off-diagonal norm the solver leaves behind shows up in the Ritz residuals as at most that norm
            // over theta_npc, i.e. sqrt(tol2) x (theta_1 / theta_npc) x a few — 1e-4 of the residual tolerance keeps it two
            // orders below what the step is judged on (tol 1e-7: 1e-22, one sweep of the quadratic tail less than 1e-30)
            const double exact_tol2 = std::max(1e-30, (o.tol * 1e-4) * (o.tol * 1e-4));
            hipLaunchKernelGGL(k_jacobi_eig2, dim3(1), dim3(kJ2Threads), sizeof(J2Lds), ctx->stream, d_h1, l_act,
                               w.dM2, w.dTheta, d_status, loose ? loose_tol2 : exact_tol2,
                               o.direct ? l_act : std::min(l_act, o.n_pc + 4));      // (the guard columns' block need not be diagonal: jacobi.inl)
            hipLaunchKernelGGL(k_ritz_post, dim3(kRitzBlocks), dim3(256), 0, ctx->stream, (const double*)w.W, w.Wp,
                               (const double*)w.dM2, (const double*)w.dTheta, k, w.A1, w.A2, d_ritz, acc_apply ? 1 : 0);
            hipLaunchKernelGGL(k_resid_final, dim3(1), dim3(1024), 0, ctx->stream, (const double*)d_ritz, kRitzBlocks,
                               (const double*)w.dTheta, o.n_pc, l_act, d_status, d_status_sel, w.dRho, w.dColmax,
                               d_res + kSlotDoubles * slot);
        }
        SRX_HIP(ctx, hipGetLastError());
        return SRX_OK;
    };
    // the read-back of a Ritz step (pinned slot + event): always a plain stream operation, never captured
    auto ritz_readback = [&](int slot) -> int32_t {
        SRX_HIP(ctx, hipMemcpyAsync(ctx->pin_async + kSlotDoubles * slot, d_res + kSlotDoubles * slot,
                                    kSlotDoubles * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SRX_HIP(ctx, hipEventRecord(ctx->async_ev[slot], ctx->stream));
        return SRX_OK;
    };
    // the extra applications of C between two Rayleigh–Ritz steps (cheap dense products only; the
    // block stays well conditioned: kappa ~ (theta_1/theta_l)^power) and the next CholeskyQR
    auto advance = [&]() -> int32_t {
        // continue from the ROTATED block A1 = (C W) U (same span): its columns are close to eigenvectors,
        // so the next projected matrix is close to diagonal and its Jacobi solve takes 2-3 sweeps, not 8
        const double* res;
        SRX_TRY(apply_n(w.A1, o.power - 1, false, &res));
        return orth(res);
    };
    double spread = 1.0;               // theta_1 / theta_l of the last collected Ritz step
    auto collect = [&](int slot, double& r, double& ratio) -> int32_t {
        SRX_HIP(ctx, hipEventSynchronize(ctx->async_ev[slot]));
        r = ctx->pin_async[kSlotDoubles * slot];
        const int st = (int)ctx->pin_async[kSlotDoubles * slot + 1];
        ratio = ctx->pin_async[kSlotDoubles * slot + 2];
        spread = ctx->pin_async[kSlotDoubles * slot + 4];
        if ((int)ctx->pin_async[kSlotDoubles * slot + 3] & 1)
            return fail(ctx, SRX_E_NAN, "NaN gene variance: called `Option::unwrap()` on a `None` value (partial_cmp)");
        if (st & kStatChol) return fail(ctx, SRX_E_NOCONV, "pca: block lost rank (Cholesky pivot <= 0)");
        if (st & kStatEig) return fail(ctx, SRX_E_NOCONV, "pca: l x l eigen-solver did not converge");
        if (r != r) return fail(ctx, SRX_E_NOCONV, "pca: NaN in the Ritz residual");
        return SRX_OK;
    };
    // one sweep WITHOUT a Rayleigh–Ritz step: `power` applications of C, then CholeskyQR
    auto plain_sweep = [&]() -> int32_t {
        const double* res;
        SRX_TRY(apply_n(w.W, o.power, true, &res));
        return orth(res);
    };

    // segment "start": random block, CholeskyQR2, warm-up sweeps (the first Ritz residuals are O(1) whatever
    // happens — no Rayleigh–Ritz step to learn that), first Ritz step
    auto seg_start = [&]() -> int32_t {
        if (o.direct) {                    // W = I (k x k, k = l_act): H = C itself, Ritz pairs = eigenpairs whatever the rank
            SRX_HIP(ctx, hipMemsetAsync(d_status, 0, kStatusBytes + (size_t)L * L * sizeof(double), ctx->stream));
            hipLaunchKernelGGL(k_identity_block, dim3((unsigned)((kl + 255) / 256)), dim3(256), 0, ctx->stream, k, w.W);
            SRX_HIP(ctx, hipGetLastError());
            if (acc_apply) SRX_HIP(ctx, hipMemsetAsync(w.Wp, 0, kl * 8, ctx->stream));      // (no start block, no CholeskyQR: nothing has zeroed it)
            return ritz_kernels(0, false);
        }
        // With a warm-up sweep the random block goes straight into C^power: the CholeskyQR that ends the sweep is the first
        // orthonormalisation the block needs (the conditioning of C^power W is that of the operator's spectrum whether or not
        // the Gaussian W — kappa ~ 1.4 at k = 2000, l = 64 — was orthonormalised first).  Without one (matrix-free solver,
        // robust mode) the Rayleigh-Ritz step needs an orthonormal block: CholeskyQR2 on the random start.
        const bool start_orth = o.robust || o.warm < 1;
        // (Gram solver: the start block goes into W and the three scratch blocks start zeroed; with start_orth the block goes
        //  into Wp, and the CholeskyQR that follows leaves the three zeroed)
        // (the status words and the 64 x 64 projected matrix behind them start from zero: the start block's kernel does it)
        hipLaunchKernelGGL(k_init_block, dim3((unsigned)((kl + 255) / 256)), dim3(256), 0, ctx->stream, o.seed, k, l_act,
                           start_orth ? w.Wp : w.W, acc_apply && !start_orth ? w.Wp : (double*)nullptr,
                           acc_apply ? w.A1 : (double*)nullptr, acc_apply ? w.T : (double*)nullptr, reinterpret_cast<uint32_t*>(d_status),
                           (int)((kStatusBytes + (size_t)L * L * sizeof(double)) / 4));
        if (start_orth) {
            SRX_TRY(orth(w.Wp));
            SRX_TRY(orth(w.W));
        }
        for (int sweep = 0; sweep < o.warm; ++sweep) SRX_TRY(plain_sweep());
        return ritz_kernels(0, true);
    };
    // Chebyshev filter after the first Ritz step (Gram solver: w.T is free and `apply` has no collective).
    // Speculative part: Y1 and Z = C Y1 (needed whatever the degree turns out to be, d >= 2).
    const unsigned cheb_grid = (unsigned)((kl + 255) / 256);
    auto cheb_spec = [&]() -> int32_t {
        hipLaunchKernelGGL(k_cheb_first, dim3(cheb_grid), dim3(256), 0, ctx->stream, w.A1, (const double*)w.A2,
                           (const double*)w.dTheta, l_act, kl);
        SRX_HIP(ctx, hipGetLastError());
        return apply(w.A1, w.Wp, acc_apply);          // (Gram solver: the Ritz tail left Wp zeroed)
    };
    // the rest of a degree-d filter (Z = C Y1 is in Wp, cur = A1, prev = A2) and the CholeskyQR behind it
    auto cheb_filter = [&](int d) -> int32_t {
        double *cur = w.A1, *prev = w.A2;
        for (int j = 1; j < d; ++j) {
            if (j > 1) SRX_TRY(apply(cur, w.Wp, true));          // (the step before left Wp zeroed)
            hipLaunchKernelGGL(k_cheb_step, dim3(cheb_grid), dim3(256), 0, ctx->stream, w.Wp,
                               (const double*)cur, prev, (const double*)w.dTheta, l_act, kl);
            double* t = cur;
            cur = prev;
            prev = t;
        }
        hipLaunchKernelGGL(k_cheb_scale, dim3(cheb_grid), dim3(256), 0, ctx->stream, cur, (const double*)w.dTheta, l_act, d, kl);
        SRX_HIP(ctx, hipGetLastError());
        return orth(cur);
    };
    // ... then, where ONE filter of the degree the block can take will not reach the tolerance anyway (flat tails: theta_l / theta_npc
    // = 0.97 gains 10x per degree-10 filter), a SECOND filter straight behind the CholeskyQR — C W, Y1 = a C W - W, C Y1, the
    // recurrence, the CholeskyQR — and only then the Ritz step: the projected Gram product, the eigen-solve, the Ritz tail and a
    // host decision less per pair of filters.  The bounds and the column scales are the last Ritz step's (the block still is that
    // step's Ritz vectors, filtered and re-orthonormalised); the CholeskyQR between the two takes the leading directions out of
    // the guard columns as a Ritz step would.
    auto cheb_rest = [&](int d, int d2, int slot) -> int32_t {
        SRX_TRY(cheb_filter(d));
        if (d2 > 0) {
            SRX_TRY(apply(w.W, w.Wp, acc_apply));                // (Gram solver: the CholeskyQR left Wp, A1, T zeroed)
            hipLaunchKernelGGL(k_cheb_first2, dim3(cheb_grid), dim3(256), 0, ctx->stream, w.Wp, (const double*)w.W, w.A1, w.A2,
                               (const double*)w.dTheta, l_act, kl);
            SRX_HIP(ctx, hipGetLastError());
            SRX_TRY(apply(w.A1, w.Wp, true));
            SRX_TRY(cheb_filter(d2));
        }
        return ritz_kernels(slot, false, true);          // (Wp: zeroed by the last filter step, untouched by the CholeskyQR)
    };
    // segment "next": [advance] + (m - 1) plain sweeps + a Ritz step into `slot`
    auto seg_next = [&](bool with_advance, int m, int slot) -> int32_t {
        if (with_advance) SRX_TRY(advance());
        for (int sI = 1; sI < m; ++sI) SRX_TRY(plain_sweep());
        return ritz_kernels(slot);
    };

    resid = INFINITY;
    converged = false;
    iters = 0;                         // sweeps after the warm-up
    int n_ritz = 0, slot = 0;
    int q_applied = o.warm * o.power + 1;          // applications of C the block has seen (warm-up + first Ritz step)
    double r_last = INFINITY, rate_meas = 0.0;
    int sweeps_since = 0;
    if (bailed_first) *bailed_first = false;
    if (resume) {
        hipLaunchKernelGGL(k_resid_final, dim3(1), dim3(1024), 0, ctx->stream, (const double*)d_ritz, kRitzBlocks,
                           (const double*)w.dTheta, o.n_pc, l_act, d_status, d_status_sel, w.dRho, w.dColmax, d_res + kSlotDoubles * slot);
        SRX_HIP(ctx, hipGetLastError());
    } else {
        SRX_TRY(graphed(ctx, use_graph, key_base + "|start", seg_start));
    }
    SRX_TRY(ritz_readback(slot));
    for (;;) {
        ++iters;
        ++n_ritz;
        const bool first = n_ritz == 1;
        const bool cheb = use_cheb;
        if (cheb && first) {
            if (!resume) SRX_TRY(graphed(ctx, use_graph, key_base + "|cheb0", cheb_spec));   // speculative: Y1, C Y1 (resume: already there)
        } else if (first && !o.direct) SRX_TRY(graphed(ctx, use_graph, key_base + "|adv", advance));   // speculative: completes this sweep
        double r, ratio;
        SRX_TRY(collect(slot, r, ratio));
        resid = r;
        if (getenv("SRX_PCA_TRACE")) {
            fprintf(stderr, "[srx pca] sweep %d (ritz step %d): residual %.3e, theta_l/theta_npc %.3e, %d Jacobi sweeps\n", iters + o.warm,
                    n_ritz, r, ratio, (int)ctx->pin_async[kSlotDoubles * slot + 5]);
            // the eigen-solve's off-diagonal norm (wanted pairs) over the diagonal's at every sweep's measurement (a drain: trace only)
            int st[64];
            if (d2h(ctx, st, d_status, sizeof st) == SRX_OK) {
                fprintf(stderr, "[srx pca]   Jacobi off / diag per sweep:");
                for (int q = 0; q <= (int)ctx->pin_async[kSlotDoubles * slot + 5] && q < 32; ++q) {
                    float f;
                    memcpy(&f, &st[16 + q], 4);
                    fprintf(stderr, " %.2e", (double)f);
                }
                fprintf(stderr, "\n");
            }
        }
        if (r <= o.tol) {
            converged = true;
            break;
        }
        if (iters >= o.max_iter) break;
        if (first && o.bail_ratio > 0.0 && ratio > o.bail_ratio) {
            if (getenv("SRX_PCA_TRACE")) fprintf(stderr, "[srx pca] flat tail (theta_l / theta_npc = %.3f): leaving the round to the safe plan\n", ratio);
            if (bailed_first) *bailed_first = cheb;         // (the block and the queued first filter step are the next plan's to continue from)
            break;                                 // converged stays false
        }
        if (cheb) {
            // degree: T_d(t_a) >= 4 r / tol with t_a = (2 theta_npc - b) / b = 2 / ratio - 1
            const double ta = ratio > 0 && ratio < 1 ? 2.0 / ratio - 1.0 : 1.0;
            int d = 3;
            if (ta > 1.0) d = (int)std::ceil(std::acosh(std::max(4.0 * r / o.tol, 1.0)) / std::acosh(ta) - 1e-9);
            // The block captures eigenvector j up to an error ~ (b / lambda_j)^q after q applications of C, and a
            // degree-d filter multiplies that error (relative to the column's own component) by ~ (lambda_j / b)^d:
            // with d <= q the leading eigenvectors cannot swamp the other columns.  A degree-12 filter on a block
            // that had seen ONE application (SpMM solver, no warm-up) collapsed it ("block lost rank").
            if (d > q_applied) d = q_applied;
            if (d > 12) d = 12;                // T_12 of the largest t stays far inside f64; harder spectra take more rounds
            // The filter multiplies the component of every column along the leading eigenvector by T_d(t_1), t_1 =
            // 2 theta_1 / theta_l - 1, and the guard columns' own components by ~1: whatever rounding-level trace of v_1
            // a guard column carries (1e-16) must stay small against the column itself, or the block collapses onto the
            // leading directions and the next CholeskyQR finds a pivot <= 0.  T_d(t_1) <= 1e14 <=> d <= 32.9 / acosh(t_1):
            // no limit in practice when the block's spectrum spans less than 5x, 6 at 30x, 5 at 100x.
            {
                const double t1 = 2.0 * (spread > 1.0 ? spread : 1.0) - 1.0;
                const int d_safe = t1 > 1.0 + 1e-9 ? (int)std::floor(32.9 / std::acosh(t1)) : 12;
                if (d > d_safe) d = d_safe;
            }
            if (d < 2) d = 2;
            // the degree the tolerance asks for against the degree this filter may have: a second one behind it when one cannot do
            int d2 = 0;
            if (ta > 1.0) {
                const int d_need = (int)std::ceil(std::acosh(std::max(4.0 * r / o.tol, 1.0)) / std::acosh(ta) - 1e-9);
                if (d_need > d + 1) {
                    const double t1 = 2.0 * (spread > 1.0 ? spread : 1.0) - 1.0;
                    const int d_safe = t1 > 1.0 + 1e-9 ? (int)std::floor(32.9 / std::acosh(t1)) : 12;
                    d2 = std::min(std::min(d_need - d, 12), std::min(d_safe, q_applied + d));
                    if (d2 < 2) d2 = 0;
                }
            }
            q_applied += d + d2;               // d - 1 (+ d2) applications in the filter(s) + the one of the Ritz step
            iters += (d + d2 + o.power - 1) / o.power;      // counted in sweep equivalents (max_iter bounds applications of C)
            slot = (slot + 1) % kSlots;
            char kn[64];
            snprintf(kn, sizeof kn, "|cheb f%d d%d e%d s%d", first ? 1 : 0, d, d2, slot);
            SRX_TRY(graphed(ctx, use_graph, key_base + kn, [&]() -> int32_t {
                if (!first) SRX_TRY(cheb_spec());      // later rounds: nothing was queued speculatively
                return cheb_rest(d, d2, slot);
            }));
            SRX_TRY(ritz_readback(slot));
            if (getenv("SRX_PCA_TRACE")) {
                if (d2) fprintf(stderr, "[srx pca] Chebyshev filters of degree %d and %d, a CholeskyQR between them (t_a = %.3f)\n", d, d2, ta);
                else fprintf(stderr, "[srx pca] Chebyshev filter of degree %d (t_a = %.3f)\n", d, ta);
            }
            r_last = INFINITY;                 // the filter's gain says nothing about the rate of plain sweeps
            sweeps_since = 0;
            continue;
        }
        if (r_last < INFINITY && sweeps_since > 0 && r < r_last) rate_meas = std::pow(r / r_last, 1.0 / sweeps_since);
        double rate = rate_meas > 0.0 ? rate_meas : std::pow(ratio < 1.0 ? ratio : 1.0, (double)o.power);
        if (!(rate > 1e-8)) rate = 1e-8;
        if (rate > 0.9) rate = 0.9;
        int m = (int)std::ceil(std::log(o.tol / r) / std::log(rate) - 1e-9);
        if (m < 1) m = 1;
        if (m > 6) m = 6;
        if (iters + m > o.max_iter) m = o.max_iter - iters;
        iters += m - 1;
        slot = (slot + 1) % kSlots;
        {
            const bool with_adv = !first;      // the first step's half-sweep was queued speculatively
            char kn[64];
            snprintf(kn, sizeof kn, "|next a%d m%d s%d", with_adv ? 1 : 0, m, slot);
            SRX_TRY(graphed(ctx, use_graph, key_base + kn, [&]() { return seg_next(with_adv, m, slot); }));
        }
        SRX_TRY(ritz_readback(slot));
        r_last = r;
        sweeps_since = m;
    }
    return SRX_OK;
}

// Components per deflation round when more than L - 8 are asked of a k > L problem (the block keeps 16 guard
// columns), and the number of rounds; the last round takes everything that is left once <= L dimensions remain.
constexpr int kPcaPerRound = 48;
constexpr int kPcaPerRoundSafe = 32;       // the fallback plan: >= 32 guard columns per round (c2, 1000 HVGs, 50 components:
                                           // 12 -> 19.6 ms, 16 -> 16.3, 24 -> 15.5, 32 -> 10.7)
// Components per round when at most `per` are asked of one round.
static std::vector<int> plan_rounds(int k /* dimension of the operator's range */, int n_pc, int per) {
    std::vector<int> counts;
    if (k <= L) {                            // the block spans the whole range: one exact round
        counts.push_back(n_pc);
        return counts;
    }
    int done = 0;
    while (done < n_pc) {
        const int take = (k - done <= L) ? n_pc - done : std::min(per, n_pc - done);
        counts.push_back(take);
        done += take;
    }
    return counts;
}

// M (64 x 64, row-major) <- diag(s) M: row r scaled by s[r]
__global__ void k_scale_rows(double* __restrict__ M, const double* __restrict__ s) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < L * L) M[e] *= s[e / L];
}
__global__ void k_sub_inplace(double* __restrict__ a, const double* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] -= b[i];
}
// columns [at, at + n) of the locked block <- the leading n Ritz vectors / values of a finished round
__global__ void k_lock_columns(double* __restrict__ Vl, double* __restrict__ thl, const double* __restrict__ V,
                               const double* __restrict__ theta, int k, int at, int n) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)k * n) return;
    const int i = (int)(e / n), c = (int)(e % n);
    Vl[(size_t)i * L + at + c] = V[(size_t)i * L + c];
    if (i == 0) thl[at + c] = theta[c];
}

// C -= V diag(theta) V^T over the first n columns of V (k x 64): the resolved eigenpairs leave the operator.
// theta_c * (v_ic * v_jc) is symmetric in (i, j) to the last bit, so C stays exactly symmetric.
__global__ void k_deflate(double* __restrict__ C, int k, const double* __restrict__ V, const double* __restrict__ theta, int n) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)k * k) return;
    const int i = (int)(e / k), j = (int)(e % k);
    const double* vi = V + (size_t)i * L;
    const double* vj = V + (size_t)j * L;
    double s = 0.0;
    for (int c = 0; c < n; ++c) s += theta[c] * (vi[c] * vj[c]);
    C[e] -= s;
}

// Size of a matrix's result allocation: the N x n_pc f64 scores followed by the block of small results
// (layout: srx_pca_state::d_small) for the most rounds either plan can take.
// Row stride (in doubles) of the score matrix IN HBM: n_pc rounded up to a whole number of 128-byte pieces (16 doubles), so
// that the 16-column piece a panel-slice workgroup of the forward SpMM writes for a cell is its own two aligned 64-byte lines
// — with the reference's N x n_pc layout (400-byte rows at n_pc = 50) every piece straddled lines shared with another
// slice's workgroup.  srx_result_fetch hands out the dense N x n_pc matrix (obsm["X_pca"], dim_red/mod.rs:105-106) whatever
// the stride in HBM is.
int scores_ld(int n_pc) {
    return (n_pc + 15) / 16 * 16;            // (c3, n_pc = 50: the forward SpMM 0.640 -> 0.616 ms against the dense layout)
}
void result_layout(uint64_t n_rows, int k, int n_pc, int dim, size_t& score_bytes, size_t& small_doubles,
                   std::vector<int>* plan_a_out, std::vector<int>* plan_b_out) {
    const size_t kl = (size_t)k * L;
    const std::vector<int> pa = plan_rounds(dim, n_pc, n_pc <= L - 8 ? n_pc : kPcaPerRound);
    const int n_b = (n_pc + kPcaPerRoundSafe - 1) / kPcaPerRoundSafe;
    const std::vector<int> pb = plan_rounds(dim, n_pc, (n_pc + n_b - 1) / n_b);
    const int rounds_cap = (int)std::max(pa.size(), pb.size());
    score_bytes = (n_rows ? n_rows : 1) * (size_t)scores_ld(n_pc) * 8;
    small_doubles = (size_t)rounds_cap * (kl + 2 * L) + 2 * (size_t)k + 2 + ((size_t)k + 1) / 2;
    if (plan_a_out) *plan_a_out = pa;
    if (plan_b_out) *plan_b_out = pb;
}
int32_t ensure_result_capacity(srx_ctx* ctx, srx_pca_state& st, size_t need) {
    if (st.scores_cap < need) {
        if (st.d_scores) SRX_HIP(ctx, hipFree(st.d_scores));
        st.d_scores = nullptr;
        st.scores_cap = 0;
        SRX_HIP(ctx, hipMalloc((void**)&st.d_scores, need));
        st.scores_cap = need;
    }
    return SRX_OK;
}

template <typename VT, typename PT>
int32_t run_pca(srx_ctx* ctx, const RowMajor* parts, int n_parts, const Tiled* t256p, double* gram_packed,
                       const Resolved& o, const std::vector<double>& mu, const std::vector<double>& dinv,
                       const HvgDev* hv, int l_act, double n_cells, srx_pca_state& st) {
    // `parts`: the row-major compacted rows of this rank — one for a resident matrix, one per row tile in backed mode (then
    // `gram_packed` holds the packed Gram matrix already summed over the row tiles); `t256p`: the 256-tiled view, made for
    // the matrix-free solver and for selections too wide for the row kernel's LDS panel slice
    const RowMajor* rmp = n_parts == 1 ? &parts[0] : nullptr;
    const int k = parts[0].k;
    struct { uint64_t n_rows, max_rows; } cc{0, 0};
    for (int i = 0; i < n_parts; ++i) {
        cc.n_rows += parts[i].n_rows;
        cc.max_rows = std::max(cc.max_rows, parts[i].n_rows);
    }
    const size_t kl = (size_t)k * L;
    Work w;
    st.d_small = nullptr;
    SRX_TRY(alloc_work(ctx, k, w));
    if (hv) {                          // selection made on the device: centring / scaling vectors are already there — read in place
        if (o.center) w.mu = hv->d_mu;      // (they are only ever read; two device copies per step were 9 us on the critical path)
        else SRX_HIP(ctx, hipMemsetAsync(w.mu, 0, (size_t)k * 8, ctx->stream));
        w.d = hv->d_dinv;
    } else {
        SRX_TRY(h2d(ctx, w.mu, mu.data(), (size_t)k * 8));
        SRX_TRY(h2d(ctx, w.d, dinv.data(), (size_t)k * 8));
    }
    PT *P, *cvec, *Y;
    SRX_TRY(scratch(ctx, "pca_P", (kl + L) * sizeof(PT), (void**)&P));
    cvec = P + kl;
    SRX_TRY(scratch(ctx, "pca_Y", (cc.max_rows ? cc.max_rows : 1) * (size_t)L * sizeof(PT), (void**)&Y));

    double resid = INFINITY;
    int iters = 0;
    bool converged = false;
    // DEFLATION ROUNDS on the explicit C (Gram solver).  A round resolves the next eigenpairs with the 64-column block,
    // writes their scores, and removes them from C (C -= V diag(theta) V^T), so that the next round's dominant
    // subspace is the one after them.
    //   plan A: everything in one round up to 56 components (48 per round beyond) — two Ritz steps when the spectrum
    //           decays across the block, the normal case;
    //   plan B: rounds of <= 32 components with 32+ guard columns each.  Taken when plan A breaks down or stalls: a flat
    //           tail (theta_64 / theta_50 -> 1) needs Chebyshev filters of high total degree, and with the dominant
    //           eigenvalues still in the operator (theta_1 / theta_64 ~ 20-100) a degree-12 filter amplifies the leading
    //           directions by T_12(t_1) ~ 1e20 over the guard columns — the block collapses onto them ("Cholesky pivot
    //           <= 0").  Once a round has deflated the leading eigenpairs the remaining spectrum is narrow and the same
    //           filters are harmless.
    const int n_pc = o.n_pc;
    // dimension of the operator's range: min(k, N - 1) (N when not centred); a block as wide as that is exact
    const int dim = o.direct ? k : (int)std::min<double>((double)k, n_cells - (o.center ? 1.0 : 0.0));
    std::vector<int> plan_a, plan_b;
    size_t score_bytes, small_doubles;
    result_layout(cc.n_rows, k, n_pc, dim, score_bytes, small_doubles, &plan_a, &plan_b);
    // one allocation: the scores, then the block of small results (srx_matrix_reserve_results makes it ahead of time)
    SRX_TRY(ensure_result_capacity(ctx, st, score_bytes + small_doubles * 8));
    double* const d_small = st.d_scores + score_bytes / 8;
    // A2 = W U are the Ritz vectors (ascending-gene row order); sign: largest-|.| entry positive.
    // scores = Z V (transform, pca/mod.rs:156-185): one forward SpMM per row tile with the panel D V, the f64 scores
    // written by the SpMM itself.  The Ritz vectors, values and signs move out of the (per-context) scratch into
    // the matrix's own block; their host copies are made by the first fetch (pca_materialize).
    auto finish_round = [&](int r, int col0, int n_r) -> int32_t {
        // signs, panel, centring partials and the copy of (Ritz vectors, values, signs) into the result block: one pass
        double* blk = d_small + (size_t)r * (kl + 2 * L);
        hipLaunchKernelGGL((k_make_panel_mb<PT>), dim3(kPanelBlocks), dim3(1024), 0, ctx->stream, (const double*)w.A2, (const double*)w.d,
                           (const double*)w.mu, (const double*)w.dColmax, (const double*)w.dTheta, k, P, w.gpart, w.dSgn, blk);
        hipLaunchKernelGGL((k_cvec_reduce<PT>), dim3(1), dim3(L), 0, ctx->stream, (const double*)w.gpart, kPanelBlocks, o.center, cvec);
        SRX_HIP(ctx, hipGetLastError());
        // scores = Z V: the transform from the row-major records, one launch per row tile (the tile-major kernel — 1.22 ms
        // at c3 against 0.84 — when its view was made: matrix-free solver, panel slice larger than the LDS, SRX_FWD_TILED)
        uint64_t row0 = 0;
        const int ld_s = scores_ld(n_pc);
        if (t256p) {
            SRX_TRY((launch_fwd<VT, PT>(ctx, *t256p, P, cvec, Y, st.d_scores + col0, n_r, ld_s)));
        } else {
            for (int i = 0; i < n_parts; ++i) {
                SRX_TRY((launch_fwd_rows<VT, PT>(ctx, parts[i], P, cvec, n_r, st.d_scores + row0 * (size_t)ld_s + col0, (PT*)nullptr, ld_s)));
                row0 += parts[i].n_rows;
            }
        }
        return SRX_OK;
    };
    // Runs a plan of deflation rounds with the solver's `apply`; `reset` restores the undeflated operator, `deflate`
    // removes the eigenpairs a round has resolved (w.A2 / w.dTheta, leading n columns).  SRX_E_NOCONV (breakdown) or
    // converged == false (budget spent) leave the decision to the caller.
    const bool acc_apply = o.solver == 1;            // the dense application accumulates into a zeroed block; w.T is free
    bool bailed_first = false;               // the last plan was left at its first Ritz step (flat tail) with its block intact
    auto run_plan = [&](const std::vector<int>& plan, int budget, bool robust, auto& apply, const void* apply_id, bool graphable,
                        auto& reset, auto& deflate, double bail = 0.0, bool resume = false) -> int32_t {
        if (!resume) SRX_TRY(reset());      // (resume: nothing was deflated, C is as expanded)
        resid = 0.0;
        converged = true;
        iters = 0;
        int done = 0;
        const int rounds = (int)plan.size();
        for (int r = 0; r < rounds; ++r) {
            Resolved o_r = o;
            o_r.n_pc = plan[r];
            o_r.max_iter = budget;
            o_r.seed = o.seed + (uint64_t)r;
            o_r.bail_ratio = bail;
            if (robust) {                       // CholeskyQR after every application, the same number of applications up front
                o_r.robust = true;
                o_r.warm = o.warm * o.power;
                o_r.power = 1;
                o_r.max_iter = budget * o.power;
            }
            const int l_r = rounds == 1 ? l_act : std::min(L, dim - done);        // dim - done: what is still in the operator
            double resid_r = INFINITY;
            int iters_r = 0;
            bool conv_r = false;
            {
                ProfScope ps_it(ctx, SRX_K_ITERATE, (double)k * k * 8.0);
                SRX_TRY(subspace_iterate(ctx, w, k, l_r, o_r, apply, apply_id, graphable, hv ? hv->d_status : nullptr, resid_r,
                                         iters_r, conv_r, acc_apply, resume && r == 0 && l_r == l_act, &bailed_first));
            }
            resid = std::max(resid, resid_r);
            iters += iters_r + o.warm;
            if (!conv_r) {
                converged = false;
                return SRX_OK;
            }
            SRX_TRY(finish_round(r, done, o_r.n_pc));
            if (r + 1 < rounds) SRX_TRY(deflate(done, o_r.n_pc));
            done += o_r.n_pc;
        }
        st.rounds = (uint32_t)rounds;
        st.round_counts = plan;
        return SRX_OK;
    };
    // plan A, then plan B if it stalled or broke down.  One round for everything gets a short budget before the safe
    // plan takes over; a plan A that already deflates (n_pc > 56) keeps the full one.
    auto solve = [&](auto& apply, const void* apply_id, bool graphable, auto& reset, auto& deflate) -> int32_t {
        const bool have_b = plan_b.size() > plan_a.size();
        if (getenv("SRX_PCA_ROBUST")) {                               // test switch: the last-resort mode from the start
            SRX_TRY(run_plan(have_b ? plan_b : plan_a, o.max_iter, true, apply, apply_id, graphable, reset, deflate));
            iters -= o.warm;
            return SRX_OK;
        }
        // (a one-round plan A is also given up at once when its first Ritz step shows a flat tail, theta_64 / theta_npc >
        //  0.93: such a round needs a total filter degree of 45+ and plan B gets there sooner)
        int32_t rc = run_plan(plan_a, have_b && plan_a.size() == 1 ? std::min(o.max_iter, 40) : o.max_iter, false, apply, apply_id,
                              graphable, reset, deflate, have_b && plan_a.size() == 1 ? 0.93 : 0.0);
        if (have_b && (rc == SRX_E_NOCONV || (rc == SRX_OK && !converged))) {
            if (getenv("SRX_PCA_TRACE"))
                fprintf(stderr, "[srx pca] plan A (%zu round(s)) %s at residual %.3e: rounds of <= %d components instead\n",
                        plan_a.size(), rc == SRX_OK ? "stalled" : "broke down", resid, plan_b[0]);
            const int spent = iters;
            // (left at the first Ritz step for its flat tail: the safe plan's first round continues from that block — the same 64
            //  columns, the same warm-up and Ritz step it would redo from a new random start)
            const bool resume = rc == SRX_OK && bailed_first && plan_a.size() == 1;
            rc = run_plan(plan_b, o.max_iter, false, apply, apply_id, graphable, reset, deflate, 0.0, resume);
            iters += resume ? 0 : spent;
        }
        if (rc == SRX_E_NOCONV) {
            // last resort: a block whose spectrum spans more than ~1e8 between two CholeskyQRs (small exact-rank
            // problems: k = 10 features of 6 cells have theta_1 / theta_5 ~ 1e3 and a sweep is three applications)
            if (getenv("SRX_PCA_TRACE")) fprintf(stderr, "[srx pca] breakdown again: robust mode (CholeskyQR3 after every application)\n");
            const int spent = iters;
            rc = run_plan(have_b ? plan_b : plan_a, o.max_iter, true, apply, apply_id, graphable, reset, deflate);
            iters += spent;
        }
        SRX_TRY(rc);
        iters -= o.warm;                                              // st.info adds it back once below
        return SRX_OK;
    };
    if (o.solver == 1) {
        // explicit Gram: G = A^T A once (all-reduced), C = D (G - c N mu mu^T) D dense
        double* C;
        SRX_TRY(scratch(ctx, "pca_C", (size_t)k * k * 8, (void**)&C));
        double* Pk = gram_packed;
        const size_t n_packed = gram_packed_count(k);
        if (!Pk) {
            Range r_("srx:gram");
            SRX_TRY(scratch(ctx, "pca_gpacked", n_packed * sizeof(double), (void**)&Pk));
            bool reduced = false;
            SRX_TRY(launch_gram<VT>(ctx, *rmp, Pk, &reduced, true));  // (a fresh sum: zeroed on the way; sharded rows: the exchange overlaps the second half)
            if (!reduced) SRX_TRY(allreduce_f64(ctx, Pk, n_packed));
        } else {
            SRX_TRY(allreduce_f64(ctx, Pk, n_packed));            // the one exchange of this solver: the packed upper triangle
        }
        auto reset = [&]() -> int32_t {
            hipLaunchKernelGGL(k_gram_expand, dim3((unsigned)(((size_t)k * k + 255) / 256)), dim3(256), 0, ctx->stream, Pk,
                               k, (const double*)w.d, (const double*)w.mu, o.center, n_cells, C);
            SRX_HIP(ctx, hipGetLastError());
            return SRX_OK;
        };
        auto apply = [&](const double* Win, double* Wout, bool out_zeroed) -> int32_t {
            ProfScope ps(ctx, SRX_K_DENSE, (double)k * k * 8.0 + 2.0 * k * L * 8.0);
            if (!out_zeroed) SRX_HIP(ctx, hipMemsetAsync(Wout, 0, kl * 8, ctx->stream));
            hipLaunchKernelGGL(k_dense_apply, dim3((k + 31) / 32, kDenseSplit), dim3(kDenseWaves * 64), kDenseLds, ctx->stream, C, Win, k, Wout);
            SRX_HIP(ctx, hipGetLastError());
            return SRX_OK;
        };
        auto deflate = [&](int, int n_r) -> int32_t {                  // C -= V diag(theta) V^T
            hipLaunchKernelGGL(k_deflate, dim3((unsigned)(((size_t)k * k + 255) / 256)), dim3(256), 0, ctx->stream, C, k,
                               (const double*)w.A2, (const double*)w.dTheta, n_r);
            SRX_HIP(ctx, hipGetLastError());
            return SRX_OK;
        };
        Range r_("srx:iterate");
        SRX_HIP(ctx, hipFuncSetAttribute((const void*)k_dense_apply, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDenseLds));
        SRX_TRY(solve(apply, C, true, reset, deflate));
    } else {
        if (n_parts != 1 || !t256p) return fail(ctx, SRX_E_ARG, "pca: the SpMM solver needs the matrix resident in one piece");
        const Tiled& t256 = *t256p;
        // matrix-free: Z^T (Z W) by a forward and a transposed SpMM; resolved eigenpairs are deflated IMPLICITLY,
        // W' -= V_lock (theta_lock * (V_lock^T W)) with the locked vectors in a k x 64 block (plan B locks <= 48)
        double* v_lock;
        SRX_TRY(scratch(ctx, "pca_lock", (kl + L) * 8, (void**)&v_lock));
        double* th_lock = v_lock + kl;
        int n_lock = 0;
        auto reset = [&]() -> int32_t {
            n_lock = 0;
            SRX_HIP(ctx, hipMemsetAsync(v_lock, 0, (kl + L) * 8, ctx->stream));
            return SRX_OK;
        };
        auto apply = [&](const double* Win, double* Wout, bool) -> int32_t {
            hipLaunchKernelGGL((k_make_panel<PT>), dim3(1), dim3(1024), 0, ctx->stream, Win, w.d, w.mu,
                               (const double*)nullptr, k, o.center, P, cvec);
            SRX_HIP(ctx, hipGetLastError());
            // forward product: the row-major batch-stream kernel where its widest slice (8 f64 columns of all k genes: k <= 2559)
            // fits the LDS — 8 passes over the records, 1.48 against 2.45 ms per application at c3 for the tile-major kernel,
            // which reads the matrix once whatever k is and stays the route beyond (narrower slices mean 16+ passes)
            if (parts[0].pk && fwd_rows_q<PT>(k) >= 2)
                SRX_TRY((launch_fwd_rows<VT, PT>(ctx, parts[0], P, cvec, L, (double*)nullptr, Y, L)));
            else SRX_TRY((launch_fwd<VT, PT>(ctx, t256, P, cvec, Y)));
            SRX_TRY((launch_t<VT, PT>(ctx, t256, Y, w.T)));
            SRX_TRY(allreduce_f64(ctx, w.T, kl + L));             // the one exchange per iteration
            hipLaunchKernelGGL(k_finish_t, dim3((unsigned)((kl + 255) / 256)), dim3(256), 0, ctx->stream, w.T, w.d, w.mu,
                               k, o.center, Wout);
            SRX_HIP(ctx, hipGetLastError());
            if (n_lock > 0) {
                SRX_TRY(gram2(ctx, w, v_lock, Win, k));                               // dHG <- V_lock^T W (64 x 64)
                hipLaunchKernelGGL(k_scale_rows, dim3((L * L + 255) / 256), dim3(256), 0, ctx->stream, w.dHG, (const double*)th_lock);
                hipLaunchKernelGGL(k_right_mul, dim3(128), dim3(256), 0, ctx->stream, (const double*)v_lock, (const double*)w.dHG, k, w.T);
                hipLaunchKernelGGL(k_sub_inplace, dim3((unsigned)((kl + 255) / 256)), dim3(256), 0, ctx->stream, Wout, (const double*)w.T, kl);
                SRX_HIP(ctx, hipGetLastError());
            }
            return SRX_OK;
        };
        auto deflate = [&](int, int n_r) -> int32_t {                  // the round's vectors join the locked block
            if (n_lock + n_r > L) return fail(ctx, SRX_E_ARG, "pca: more than %d locked vectors in the matrix-free solver", L);
            hipLaunchKernelGGL(k_lock_columns, dim3((unsigned)(((size_t)k * n_r + 255) / 256)), dim3(256), 0, ctx->stream, v_lock,
                               th_lock, (const double*)w.A2, (const double*)w.dTheta, k, n_lock, n_r);
            SRX_HIP(ctx, hipGetLastError());
            n_lock += n_r;
            return SRX_OK;
        };
        SRX_TRY(solve(apply, nullptr, false, reset, deflate));
    }
    st.d_small = d_small;
    st.info.n_iter = (uint32_t)(iters + o.warm);
    st.info.residual = resid;
    if (!converged)
        return fail(ctx, SRX_E_NOCONV, "pca: subspace iteration stopped at max_iter=%d with residual %.3e > tol %.3e",
                    o.max_iter, resid, o.tol);
    return SRX_OK;
}

// Defaults and limits of pca_inplace (dim_red/mod.rs:38-57) and of this solver.
int32_t resolve_opts(srx_ctx* ctx, const srx_pca_opts* opts, int k, uint64_t Ng, bool f32, Resolved& o, int& l_act) {
    // dim_red/mod.rs:38-41: column(0)/column(1) and slice(..5) panic when k < 2 or N < 5
    if (k < 2 || Ng < 5) return fail(ctx, SRX_E_SHAPE, "pca_inplace needs >= 2 selected features and >= 5 cells (k=%d, N=%llu)",
                                     k, (unsigned long long)Ng);
    int want = (!opts || opts->n_components < 0) ? 2 : opts->n_components;      // :52
    o.n_pc = std::min(want, k);
    o.center = (!opts || opts->center < 0) ? 1 : (opts->center != 0);           // :55
    o.scale = (!opts || opts->scale < 0) ? 1 : (opts->scale != 0);              // :56
    o.max_iter = (opts && opts->max_iter > 0) ? opts->max_iter : 0;          // default set below, once the solver is known
    o.tol = (opts && opts->tol > 0) ? opts->tol : 0.0;
    o.seed = opts ? opts->seed : 0;
    o.solver = opts ? opts->solver : 0;
    if (o.solver < 0 || o.solver > 2) return fail(ctx, SRX_E_ARG, "pca: solver must be 0 (auto), 1 (gram) or 2 (spmm)");
    // auto: the explicit Gram matrix as long as it fits (k <= 16384: 2 GB of f64).  At c2's size k = 6000 / 8000 cost
    // 13.6 / 20.5 ms per pipeline against 23.7 / 30 of the matrix-free iteration, and on a flat-tailed spectrum at
    // k = 9000 (general compaction route) 105 ms against 297
    if (o.solver == 0) o.solver = k <= 16384 ? 1 : 2;
    // a sweep is `power` applications of C: the same default budget of 600 applications for both solvers
    if (o.max_iter == 0) o.max_iter = o.solver == 1 ? 200 : 600;
    o.power = o.solver == 1 ? 3 : 1;
    o.warm = o.solver == 1 ? 2 : 0;
    // (Round 5, measured at c3 with the schedule overridden — warm-up sweeps x applications per sweep: 2 x 3 (this) 0.97 ms, residual
    //  1.0e-8; 1 x 6 0.90 ms, the same residuals to four digits — one CholeskyQR less, on a block of condition (theta_1 / theta_l)^6; 1 x 5
    //  0.88 ms, 4.9e-8; 1 x 4 and 2 x 2 need a third Ritz step, 1.2-1.3 ms.  The 0.07-0.1 ms are not worth a CholeskyQR on the sixth
    //  power of the spectrum's spread: kept at 2 x 3.  profiles/r05_knockouts.md)
    if (o.solver == 1 && k > 16384) return fail(ctx, SRX_E_ARG, "pca: the Gram solver holds a k x k f64 matrix; k=%d is too large", k);
    // default tolerance on the relative Ritz residual: what the arithmetic of the solver supports
    if (o.tol == 0.0) o.tol = f32 ? 1e-7 : 1e-9;
    if (o.n_pc < 1) return fail(ctx, SRX_E_ARG, "pca: n_components must be >= 1");
    if (opts && opts->block != 0 && opts->block != L) return fail(ctx, SRX_E_ARG, "pca: only block = %d is built", L);
    // Z has rank <= min(k, N - 1) (N when not centred): a block wider than that cannot stay independent under C
    const uint64_t rank_bound = std::min<uint64_t>((uint64_t)k, Ng - (o.center ? 1 : 0));
    if ((uint64_t)o.n_pc > rank_bound && !(k <= L && o.solver == 1))
        return fail(ctx, SRX_E_ARG, "pca: n_components %d exceeds the rank of the data (min(k, N%s) = %llu)", o.n_pc,
                    o.center ? " - 1" : "", (unsigned long long)rank_bound);
    l_act = (int)std::min<uint64_t>((uint64_t)L, rank_bound);
    // k <= 64 with the explicit matrix: the k x k matrix goes straight to the eigen-solver (exact for any rank)
    if (k <= L && o.solver == 1) {
        o.direct = true;
        l_act = k;
    }
    if (o.n_pc > l_act && o.solver != 1)
        return fail(ctx, SRX_E_ARG, "pca: n_components %d exceeds what the %d-column block resolves (max %d)", o.n_pc, L, l_act);
    // beyond L - 8 components the Gram solver runs deflation rounds on the explicit k x k matrix; the matrix-free
    // solver has nothing to deflate
    if (rank_bound > (uint64_t)L && o.n_pc > L - 8 && o.solver != 1)
        return fail(ctx, SRX_E_ARG, "pca: n_components %d > %d needs the Gram solver (k <= 16384, opts.solver = 1)", o.n_pc, L - 8);
    return SRX_OK;
}

// Everything the host side of the result needs stays on the device until the first fetch (pca_materialize).
int32_t stash_results(srx_ctx* ctx, srx_pca_state& st, int k, int n_pc, const HvgDev* hv,
                             const std::vector<double>& mu, const std::vector<double>& sd, double trace,
                             const std::vector<uint64_t>& selv) {
    // Everything the host side of the result needs stays on the device until the first fetch.
    const size_t kl = (size_t)k * L;
    double* sm = st.d_small + (size_t)st.rounds * (kl + 2 * L);      // behind the per-round blocks
    st.dev_sel = hv != nullptr;
    if (hv) {
        CopySegs cs;                                           // (four runtime copies were 20 us of dispatches)
        cs.src[0] = (const uint32_t*)hv->d_mu;       cs.dst[0] = (uint32_t*)sm;                          cs.words[0] = (uint32_t)k * 2;
        cs.src[1] = (const uint32_t*)hv->d_sd;       cs.dst[1] = (uint32_t*)(sm + k);                    cs.words[1] = (uint32_t)k * 2;
        cs.src[2] = (const uint32_t*)hv->d_trace;    cs.dst[2] = (uint32_t*)(sm + 2 * (size_t)k);        cs.words[2] = 2;
        cs.src[3] = (const uint32_t*)hv->d_sel_rank; cs.dst[3] = (uint32_t*)(sm + 2 * (size_t)k + 2);    cs.words[3] = (uint32_t)k;
        hipLaunchKernelGGL(k_copy_segs, dim3(16), dim3(256), 0, ctx->stream, cs);
        SRX_HIP(ctx, hipGetLastError());
        st.sel.clear();
    } else {
        st.pend_mu = mu;
        st.pend_sd = sd;
        st.pend_trace = trace;
        st.sel = selv;
    }
    st.k = (uint32_t)k;
    st.n_pc = (uint32_t)n_pc;
    st.host_pending = true;
    st.valid = true;
    return SRX_OK;
}

template int32_t run_pca<float, float>(srx_ctx*, const RowMajor*, int, const Tiled*, double*, const Resolved&, const std::vector<double>&,
                                        const std::vector<double>&, const HvgDev*, int, double, srx_pca_state&);
template int32_t run_pca<float, double>(srx_ctx*, const RowMajor*, int, const Tiled*, double*, const Resolved&, const std::vector<double>&,
                                         const std::vector<double>&, const HvgDev*, int, double, srx_pca_state&);
template int32_t run_pca<double, double>(srx_ctx*, const RowMajor*, int, const Tiled*, double*, const Resolved&, const std::vector<double>&,
                                          const std::vector<double>&, const HvgDev*, int, double, srx_pca_state&);

}  // namespace srx

using namespace srx;

extern "C" {

// Kernel-level entry point: Y = X[:, sel] * P, T = X[:, sel]^T * Y and G = X[:, sel]^T X[:, sel]
// for a caller-supplied 64-column panel (no centring / scaling).  Exists so the SpMM and Gram
// kernels can be checked against a CPU reference in isolation, and as the raw operators.
int32_t srx_spmm(srx_mat* m, const uint64_t* sel, uint64_t k64, const double* panel, double* y_out, double* t_out,
                 double* gram_out) {
    if (!m || !sel) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    if ((y_out || t_out) && !panel) return fail(m->ctx, SRX_E_ARG, "srx_spmm: panel is required for y/t");
    if (m->csc) return fail(m->ctx, SRX_E_FORMAT, "srx_spmm walks cells: convert the CSC matrix with srx_matrix_to_csr");
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    const int k = (int)k64;
    const uint64_t G = m->n_cols;
    std::vector<int32_t> remap(G, -1);
    for (int s = 0; s < k; ++s) {
        if (sel[s] >= G) return fail(ctx, SRX_E_BOUNDS, "selected feature index out of bounds");
        if (s > 0 && sel[s] <= sel[s - 1]) return fail(ctx, SRX_E_ARG, "srx_spmm: sel must be strictly ascending");
        remap[sel[s]] = s;
    }
    Tiled c256;
    RowMajor crm;
    if ((k + KG - 1) / KG <= kWave) {
        SRX_TRY(build_tiled_fused(m, remap, k, crm, &c256));
    } else {
        CompactCsr cc;
        SRX_TRY(build_compact(m, remap, k, cc, crm));
        SRX_TRY(retile(m, cc, KT, c256));
    }
    const size_t kl = (size_t)k * L;
    double* T;
    SRX_TRY(scratch(ctx, "pca_T", (kl + L) * 8, (void**)&T));
    auto run = [&](auto vt, auto pt) -> int32_t {
        using VT = decltype(vt);
        using PT = decltype(pt);
        if (y_out || t_out) {
            const Tiled& c = c256;
            PT *P, *Y;
            SRX_TRY(scratch(ctx, "pca_P", (kl + L) * sizeof(PT), (void**)&P));
            SRX_TRY(scratch(ctx, "pca_Y", (c.n_rows ? c.n_rows : 1) * (size_t)L * sizeof(PT), (void**)&Y));
            std::vector<PT> hp(kl + L, PT(0));
            for (size_t e = 0; e < kl; ++e) hp[e] = (PT)panel[e];
            SRX_TRY(h2d(ctx, P, hp.data(), (kl + L) * sizeof(PT)));
            // SRX_SPMM_ROWS=1 (kernel-level parity tests): the product from the row-major records — the kernels the transform
            // runs (k_spmm_ranges, or k_spmm_rows under SRX_FWD_RANGES=0) — instead of the tile-major view
            if (getenv("SRX_SPMM_ROWS") && atoi(getenv("SRX_SPMM_ROWS"))) {
                if (!crm.perm) SRX_TRY(build_row_order(ctx, crm));
                SRX_TRY((launch_fwd_rows<VT, PT>(ctx, crm, P, P + kl, L, (double*)nullptr, Y, L)));
            } else {
                SRX_TRY((launch_fwd<VT, PT>(ctx, c, P, P + kl, Y)));
            }
            if (y_out) {
                std::vector<PT> hy(c.n_rows * (size_t)L);
                SRX_TRY(d2h(ctx, hy.data(), Y, hy.size() * sizeof(PT)));
                for (size_t e = 0; e < hy.size(); ++e) y_out[e] = (double)hy[e];
            }
            if (t_out) {
                SRX_TRY((launch_t<VT, PT>(ctx, c, Y, T)));
                SRX_TRY(d2h(ctx, t_out, T, kl * 8));
            }
        }
        if (gram_out) {
            double *C, *Pk;
            SRX_TRY(scratch(ctx, "pca_C", (size_t)k * k * 8, (void**)&C));
            SRX_TRY(scratch(ctx, "pca_gpacked", gram_packed_count(k) * sizeof(double), (void**)&Pk));
            SRX_HIP(ctx, hipMemsetAsync(Pk, 0, gram_packed_count(k) * sizeof(double), ctx->stream));
            SRX_TRY(launch_gram<VT>(ctx, crm, Pk));
            hipLaunchKernelGGL(k_gram_expand, dim3((unsigned)(((size_t)k * k + 255) / 256)), dim3(256), 0, ctx->stream, Pk,
                               k, (const double*)nullptr, (const double*)nullptr, 0, 0.0, C);
            SRX_HIP(ctx, hipGetLastError());
            SRX_TRY(d2h(ctx, gram_out, C, (size_t)k * k * 8));
        }
        return SRX_OK;
    };
    return is_f32(m) ? run(float{}, float{}) : run(double{}, double{});
}

}  // extern "C"
