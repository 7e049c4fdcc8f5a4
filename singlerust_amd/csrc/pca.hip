// pca.hip — dim_red::pca_inplace (src/memory/processing/dim_red/mod.rs:24-94) on the GPU.
//
// The reference densifies X[:, sel] to an N x k f64 matrix (src/shared/mod.rs:230-259) and
// runs a full SVD of the standardised copy (spec: src/shared/processing/pca/mod.rs:74-154).
// Here the standardised matrix  Z = (X[:, sel] - c 1 mu^T) D   (c = center, D = diag(1/std)
// when scale) is never formed.  The top eigenpairs of C = Z^T Z are found by block subspace
// iteration with a Rayleigh–Ritz step on a k x 64 block W; two ways of applying C:
//
//  GRAM solver (default up to k = 16384)
//     G = A^T A is accumulated ONCE, exactly, in f64 (gram.inl): every kept entry times the suffix of its row, one f64 LDS
//     atomic per product, G's rows owned in stripes by the workgroups (12-byte owner records made by a bucket pass).
//     C = D (G - c N mu mu^T) D is then a dense k x k matrix and every iteration is a dense (k x k)(k x 64) product on the
//     f64 matrix cores (iterate.inl).  Across row shards: ONE all-reduce of G's packed upper triangle, half of it under the
//     kernel.  The whole iteration (CholeskyQR, Jacobi eigen-solve of the projected matrix — jacobi.inl —, Chebyshev filter,
//     residuals) runs on the device, replayed from hipGraphs; the host reads one residual per Rayleigh–Ritz step.
//  SPMM solver (matrix-free, any k)
//     forward     Y  = Z W   = A (D W) - 1 (mu^T D W)        sparse x dense panel, panel in LDS        (spmm.inl)
//     transposed  W' = Z^T Y = D (A^T Y - c mu (1^T Y))      scatter form, LDS f64 atomics
//     N m 64 lane-atomics PER ITERATION; across shards one all-reduce of the k x 64 block each.
//
// Either way the scores are one forward SpMM  Z V  (transform, pca/mod.rs:156-185).
// A is the HVG-COMPACTED matrix (compact.inl): row-major (column, value) records — what the Gram kernel and the forward
// product read — and, for the matrix-free solver, a tile-major view; everything of size k x 64 is replicated per rank and
// kept in f64.  Three translation units share pca_internal.hpp: pca_form.hip (compaction, owner records, G = A^T A and its
// exchange across row shards), pca_solve.hip (the sparse products, the k x 64 iteration, the solver driver run_pca,
// srx_spmm) and this file: feature selection plumbing, the per-matrix driver pca_device, the host copies of the results, the
// out-of-core sessions (backed.inl) and the C entry points srx_pca / srx_pipeline / srx_result_fetch.
//
// Measured on MI355X (profiles/r01_*): LDS f32 float atomics (ds_add_f32) run ~10x slower than
// ds_add_f64, so every LDS accumulation here is f64; pure-f32 accumulation also stalls at ~2e-5
// eigenvector error on close eigenvalue pairs, f64 accumulation reaches ~3e-6 with f32 values.
//
// Algorithmic bytes per launch, s_v = bytes per stored value:
//   forward     nnz_w*(4+s_v) + (n_t N+1)*8 + N*64*4 (write Y) + k*64*4 (panel)
//   transposed  nnz_w*(4+s_v) + (n_t N+1)*8 + N*64*4 (read Y)  + k*64*8 (result)
#include "pca_internal.hpp"

namespace srx {

// Host-side view of an explicit selection: ascending-gene order, remap table, and the all-cells mean / std
// (ddof 0) of the selected genes from the (global) moments of `m`.
static int32_t prepare_host_selection(srx_mat* m, const std::vector<uint64_t>& selv, const Resolved& o,
                                      std::vector<int>& order, std::vector<int>& slot_of_sel,
                                      std::vector<int32_t>& remap, std::vector<double>& mu, std::vector<double>& sd,
                                      std::vector<double>& dinv, double& trace) {
    srx_ctx* ctx = m->ctx;
    const uint64_t G = m->n_cols;
    const int k = (int)selv.size();
    const double Nd = (double)m->n_rows_global;
    // ascending-gene-order view of the selection; remap table; permutation back to selection order
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return selv[a] < selv[b]; });
    remap.assign(G, -1);
    for (int s = 0; s < k; ++s) {
        uint64_t g = selv[order[s]];
        if (g >= G) return fail(ctx, SRX_E_BOUNDS, "selected feature index %llu out of bounds (n_vars = %llu)",
                                (unsigned long long)g, (unsigned long long)G);
        if (remap[g] >= 0) return fail(ctx, SRX_E_ARG, "selected feature index %llu appears twice", (unsigned long long)g);
        remap[g] = s;
        slot_of_sel[order[s]] = s;
    }
    // all-cells column mean / std (ddof 0) of the selected genes from the one moments pass
    // (pca/mod.rs:87-91): mean = sum/N, var = sumsq/N - mean^2
    std::vector<double> hsum(G), hsq(G);
    SRX_TRY(d2h(ctx, hsum.data(), m->d_sum, G * 8));
    SRX_TRY(d2h(ctx, hsq.data(), m->d_sq, G * 8));
    for (int s = 0; s < k; ++s) {
        uint64_t g = selv[order[s]];
        double mean = hsum[g] / Nd;
        double var = hsq[g] / Nd - mean * mean;
        if (var < 0) var = 0;
        double std_ = std::sqrt(var);
        mu[s] = (o.center || o.scale) ? mean : 0.0;            // :85-119: mean/std stored only if center||scale
        sd[s] = o.scale ? std_ : 1.0;
        // zero-variance column: the reference divides by 0 (NaN, :108); treated as std 1 here
        dinv[s] = (o.scale && std_ > 0) ? 1.0 / std_ : 1.0;
        double ss = o.center ? (hsq[g] - Nd * mean * mean) : hsq[g];
        if (ss < 0) ss = 0;
        trace += dinv[s] * dinv[s] * ss;
    }
    return SRX_OK;
}

// The in-place normalise + log1p of a matrix whose pipeline has been reading it raw (RowXf) and did not get as far as the
// moments pass that stores the transformed values itself (an error on the way, or a selection too wide for the fused
// compaction): from the row sums in m->d_row_sum, on the context's stream.
static int32_t launch_writeback(srx_mat* m) {
    if (!m->lazy_pending) return SRX_OK;
    m->lazy_pending = false;
    // (the moments cached on the matrix are those of the f64 transform, not of the values as stored: the version bump of
    //  launch_row_apply retires them — a later compute_variance sees what X holds)
    return launch_row_apply(m, m->lazy_target, m->ctx->stream);
}

// Everything up to and including the scores, left in m->pca (device scores + small host vectors).
// `hvg_n` > 0: FeatureSelection::HighlyVariable(hvg_n) made on the device (the pipeline's route: no host round
// trip between the moments pass and the compaction); otherwise `sel` (nullptr = all features).
static int32_t pca_device(srx_mat* m, const uint64_t* sel, uint64_t k64, const srx_pca_opts* opts, uint64_t hvg_n = 0,
                          RowXf xf = RowXf{}) {
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    if (m->csc) {
        // compaction, Gram and SpMM walk cells: solve on the CSR of X (device transpose, csc.hip) and hand the
        // result to the CSC handle (convert_to_array_f64_csc_selected, src/shared/mod.rs:261-290, builds the same
        // dense matrix from either storage)
        srx_mat* t = nullptr;
        SRX_TRY(transpose_device(m, &t));
        const int32_t rc = pca_device(t, sel, k64, opts, hvg_n, RowXf{});
        std::swap(m->pca, t->pca);
        srx_matrix_free(t);
        return rc;
    }
    srx_pca_state& st = m->pca;
    st.valid = false;
    const uint64_t G = m->n_cols;
    // the fused compaction holds one tile counter per lane (<= 64 tiles of 128) and the rank kernel is O(G^2)
    const bool dev_sel = hvg_n > 0 && (hvg_n < G ? hvg_n : G) <= (uint64_t)kWave * KG && G <= 65536;
    std::vector<uint64_t> selv;
    if (dev_sel) selv.resize(hvg_n < G ? hvg_n : G);                          // filled after the solve
    else if (sel) selv.assign(sel, sel + k64);
    else { selv.resize(G); std::iota(selv.begin(), selv.end(), 0ull); }      // FeatureSelection::None, :154
    const int k = (int)selv.size();
    SRX_TRY(ensure_moments(m));                                                // also fixes n_rows_global
    const uint64_t Ng = m->n_rows_global;
    Resolved o;
    int l_act = 0;
    SRX_TRY(resolve_opts(ctx, opts, k, Ng, is_f32(m), o, l_act));

    std::vector<int> order(k), slot_of_sel(k);
    std::vector<int32_t> remap;
    std::vector<double> mu(k), sd(k), dinv(k);
    double trace = 0.0;
    const double Nd = (double)Ng;
    HvgDev hv;
    if (dev_sel) {
        Range r_("srx:select_hvg");
        ProfScope ps_sel(ctx, SRX_K_SELECT, (double)G * 24.0);
        SRX_TRY(select_hvg_device(m, hvg_n, o.center, o.scale, hv));
    } else {
        SRX_TRY(prepare_host_selection(m, selv, o, order, slot_of_sel, remap, mu, sd, dinv, trace));
    }
    std::vector<double> mu_eff = mu;
    if (!o.center) std::fill(mu_eff.begin(), mu_eff.end(), 0.0);

    // tile-major views of X[:, sel]: fused count/fill when the tile counters fit a wave, else the
    // general route through a row-major compacted CSR
    Tiled t256;
    RowMajor rm;
    Range r_compact("srx:compact");
    // the 256-tiled view only where something reads it
    const bool fits_rows = is_f32(m) ? fwd_rows_fits<float, float>(k) : fwd_rows_fits<double, double>(k);
    const bool need_t256 = o.solver == 2 || !fits_rows;
    if (dev_sel) {
        SRX_TRY(build_tiled_fused(m, hv.d_bits, hv.n_words, k, rm, need_t256 ? &t256 : nullptr, xf, o.solver == 1));
    } else if ((k + KG - 1) / KG <= kWave) {
        SRX_TRY(build_tiled_fused(m, remap, k, rm, need_t256 ? &t256 : nullptr, xf, o.solver == 1));
    } else {
        // the general route reads stored values: the matrix is transformed in place first
        if (xf.row_sum) SRX_TRY(launch_writeback(m));
        CompactCsr cc;
        SRX_TRY(build_compact(m, remap, k, cc, rm));
        if (need_t256) SRX_TRY(retile(m, cc, KT, t256));
    }
    if (!need_t256 && !rm.perm) SRX_TRY(build_row_order(ctx, rm));      // the transform walks rows by length (the fused compaction has made it already)
    st.info = srx_pca_info{};
    st.info.n_cells_global = Ng;
    st.info.k = (uint32_t)k;
    st.info.n_pc = (uint32_t)o.n_pc;
    st.info.block = L;
    st.info.nnz_selected = rm.nnz;
    st.info.solver = (uint32_t)o.solver;
    int32_t rc;
    const HvgDev* hvp = dev_sel ? &hv : nullptr;
    // f32 storage: f32 panels and products for the one transform of the Gram route; the matrix-free iteration runs its
    // panels in f64 (f32 products of Z W level the residuals of the small components off at ~1e-7 theta_1 / theta_i:
    // 6.6e-5 at k = 9000 on a flat-tailed matrix, however many sweeps)
    if (is_f32(m) && o.solver == 2)
        rc = run_pca<float, double>(ctx, &rm, 1, need_t256 ? &t256 : nullptr, nullptr, o, mu_eff, dinv, hvp, l_act, Nd, st);
    else if (is_f32(m)) rc = run_pca<float, float>(ctx, &rm, 1, need_t256 ? &t256 : nullptr, nullptr, o, mu_eff, dinv, hvp, l_act, Nd, st);
    else rc = run_pca<double, double>(ctx, &rm, 1, need_t256 ? &t256 : nullptr, nullptr, o, mu_eff, dinv, hvp, l_act, Nd, st);
    if ((rc != SRX_OK && rc != SRX_E_NOCONV) || !st.d_small) return rc;      // d_small unset: the solve broke down early
    SRX_TRY(stash_results(ctx, st, k, o.n_pc, dev_sel ? &hv : nullptr, mu, sd, trace, selv));
    return rc;
}

// Host copies of the last solve: components (k x n_pc, rows in SELECTION order, sign-fixed), explained variance
// ratio = theta / trace(Z^T Z) (pca/mod.rs:131-133: eigenvalues s^2/(N-1) over ALL components), mean / std per
// selected feature, the selection itself.  One D2H of the matrix's result block.
static int32_t pca_materialize(srx_mat* m) {
    srx_pca_state& st = m->pca;
    if (!st.valid || !st.host_pending) return SRX_OK;
    srx_ctx* ctx = m->ctx;
    const int k = (int)st.k, n_pc = (int)st.n_pc;
    const size_t kl = (size_t)k * L;
    const int rounds = (int)st.rounds;
    const size_t rblk = kl + 2 * L;
    const size_t small_doubles = (size_t)rounds * rblk + 2 * (size_t)k + 2 + ((size_t)k + 1) / 2;
    std::vector<double> blk(small_doubles);
    SRX_TRY(d2h(ctx, blk.data(), st.d_small, small_doubles * 8));
    const double* tail = blk.data() + (size_t)rounds * rblk;
    // component p lives in the round whose range of components holds it, at column p - first component of that round
    std::vector<int> first(rounds + 1, 0);
    for (int r = 0; r < rounds; ++r) first[r + 1] = first[r] + (r < (int)st.round_counts.size() ? st.round_counts[r] : 0);
    auto locate = [&](int p, const double*& hV, const double*& theta, const double*& sgn, int& c) {
        int r = 0;
        while (r + 1 < rounds && p >= first[r + 1]) ++r;
        c = p - first[r];
        hV = blk.data() + (size_t)r * rblk;
        theta = hV + kl;
        sgn = theta + L;
    };
    std::vector<double> mu, sd;
    double trace;
    if (st.dev_sel) {
        mu.assign(tail, tail + k);
        sd.assign(tail + k, tail + 2 * (size_t)k);
        trace = tail[2 * (size_t)k];
        const int32_t* hsel = reinterpret_cast<const int32_t*>(tail + 2 * (size_t)k + 2);
        st.sel.resize(k);
        for (int i = 0; i < k; ++i) st.sel[i] = (uint64_t)hsel[i];
    } else {
        mu = st.pend_mu;
        sd = st.pend_sd;
        trace = st.pend_trace;
    }
    // slot (ascending gene order) of every selected feature
    std::vector<int> order(k), slot_of_sel(k);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return st.sel[a] < st.sel[b]; });
    for (int sI = 0; sI < k; ++sI) slot_of_sel[order[sI]] = sI;
    st.components.assign((size_t)k * n_pc, 0.0);
    st.mean.resize(k);
    st.std_.resize(k);
    for (int i = 0; i < k; ++i) {
        const int sl = slot_of_sel[i];
        for (int p = 0; p < n_pc; ++p) {
            const double *hV, *theta, *sgn;
            int c;
            locate(p, hV, theta, sgn, c);
            st.components[(size_t)i * n_pc + p] = hV[(size_t)sl * L + c] * sgn[c];
        }
        st.mean[i] = mu[sl];
        st.std_[i] = sd[sl];
    }
    st.evr.resize(n_pc);
    for (int p = 0; p < n_pc; ++p) {
        const double *hV, *theta, *sgn;
        int c;
        locate(p, hV, theta, sgn, c);
        st.evr[p] = trace > 0 ? theta[c] / trace : 0.0;
    }
    st.host_pending = false;
    return SRX_OK;
}

}  // namespace srx

using namespace srx;

#include "backed.inl"

extern "C" {

int32_t srx_result_fetch(srx_mat* m, double* scores, double* components, double* evr, double* mean, double* std_,
                         uint64_t* hvg_idx) {
    if (!m) return fail(nullptr, SRX_E_ARG, "null matrix");
    srx_ctx* ctx = m->ctx;
    if (!m->pca.valid) return fail(ctx, SRX_E_ARG, "no PCA result on this matrix");
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    if (components || evr || mean || std_ || hvg_idx) SRX_TRY(pca_materialize(m));
    const srx_pca_state& st = m->pca;
    if (scores) SRX_TRY(d2h_rows(ctx, scores, st.d_scores, m->csc ? m->n_cols : m->n_rows, (size_t)st.n_pc * 8, (size_t)scores_ld((int)st.n_pc) * 8));
    if (components) memcpy(components, st.components.data(), st.components.size() * 8);
    if (evr) memcpy(evr, st.evr.data(), st.evr.size() * 8);
    if (mean) memcpy(mean, st.mean.data(), st.mean.size() * 8);
    if (std_) memcpy(std_, st.std_.data(), st.std_.size() * 8);
    if (hvg_idx) memcpy(hvg_idx, st.sel.data(), st.sel.size() * 8);
    return SRX_OK;
}

int32_t srx_matrix_reserve_results(srx_mat* m, uint64_t n_selected, int32_t n_components) {
    if (!m) return fail(nullptr, SRX_E_ARG, "null matrix");
    srx_ctx* ctx = m->ctx;
    if (n_components < 1 || n_selected < 1) return fail(ctx, SRX_E_ARG, "reserve_results: need n_selected >= 1 and n_components >= 1");
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n_cells = m->csc ? m->n_cols : m->n_rows, n_genes = m->csc ? m->n_rows : m->n_cols;
    const int k = (int)std::min<uint64_t>(n_selected, n_genes);
    const int n_pc = std::min(n_components, k);
    size_t score_bytes, small_doubles;
    // dim = k bounds the number of rounds from above (fewer dimensions never take more rounds)
    result_layout(n_cells, k, n_pc, k, score_bytes, small_doubles);
    SRX_TRY(ensure_result_capacity(ctx, m->pca, score_bytes + small_doubles * 8));
    // the per-cell sums of the pipeline's first pass belong to the same promise: no device allocation inside a step
    if (!m->csc && !m->d_row_sum) SRX_HIP(ctx, hipMalloc((void**)&m->d_row_sum, (m->n_rows ? m->n_rows : 1) * sizeof(double)));
    return SRX_OK;
}

int32_t srx_pca(srx_mat* m, const uint64_t* sel, uint64_t k, const srx_pca_opts* opts, double* scores,
                double* components, double* evr, double* mean, double* std_, srx_pca_info* info) {
    if (!m) return fail(nullptr, SRX_E_ARG, "null matrix");
    int32_t rc = pca_device(m, sel, k, opts);
    if (rc != SRX_OK && rc != SRX_E_NOCONV) return rc;
    if (info) *info = m->pca.info;
    int32_t rc2 = srx_result_fetch(m, scores, components, evr, mean, std_, nullptr);
    return rc2 != SRX_OK ? rc2 : rc;
}

int32_t srx_pca_loadings(const double* components, const double* std_, const uint64_t* sel, uint64_t k,
                         uint64_t n_pc, uint64_t n_vars, double* out) {
    if (!components || !std_ || !sel || !out) return fail(nullptr, SRX_E_ARG, "null argument");
    // dim_red/mod.rs:111-116: zeros(n_vars x n_pc); row sel[i] <- loadings row i;
    // loadings = components^T * std (pca/mod.rs:204-215) i.e. component[i][p] * std[i]
    for (uint64_t t = 0; t < n_vars * n_pc; ++t) out[t] = 0.0;
    for (uint64_t i = 0; i < k; ++i) {
        if (sel[i] >= n_vars) return fail(nullptr, SRX_E_BOUNDS, "feature index out of bounds");
        for (uint64_t p = 0; p < n_pc; ++p) out[sel[i] * n_pc + p] = components[i * n_pc + p] * std_[i];
    }
    return SRX_OK;
}

int32_t srx_pipeline(srx_mat* m, double target_sum, uint64_t n_hvg, const srx_pca_opts* opts,
                     srx_pipeline_result* res) {
    if (!m) return fail(nullptr, SRX_E_ARG, "null matrix");
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    auto cleanup = [&]() { for (auto& e : ev) if (e) (void)hipEventDestroy(e); };
    for (auto& e : ev)
        if (hipEventCreate(&e) != hipSuccess) {
            cleanup();
            return fail(ctx, SRX_E_HIP, "srx_pipeline: hipEventCreate failed");
        }
    int32_t rc = SRX_OK;
    // SRX_STORE_AUTO: the storage follows the reference's variant (scale/mod.rs:74-83: X becomes DynCsrMatrix::F64 at
    // normalize_total) exactly as under the three separate calls — the raw values widen to f64 first and the whole path runs at
    // f64 storage.  The f32 fast path (X ends as f32(ln_1p(f64(v) * scale))) is for handles created with SRX_STORE_F32.
    if (!m->csc) rc = promote_to_f64(m);
    if (rc != SRX_OK) {
        cleanup();
        return rc;
    }
    (void)hipEventRecord(ev[0], ctx->stream);
    // normalize_total_inplace(target, Row) + log1p_transform_inplace.  CSR: the passes that follow read the RAW matrix
    // and form y = ln_1p(v * scale_row) on the fly in f64 (RowXf) — so the per-gene moments behind HighlyVariable(n)
    // are those of the reference's f64 values whatever the storage type — and the in-place write-back of y runs on the
    // side stream beside the Gram kernel.  CSC (cells are columns): the two calls, then the stored values.
    const uint64_t take = n_hvg < m->n_cols ? n_hvg : m->n_cols;
    const bool lazy = !m->csc;
    RowXf xf;
    bool wrote_back = false;
    {
    Range r_("srx:normalize");
    if (m->csc) rc = srx_normalize_log1p_inplace(m, target_sum, nullptr);
    else {
        rc = launch_row_sums(m);
        xf.row_sum = m->d_row_sum;
        xf.target = target_sum;
        xf.write_back = true;
        m->lazy_pending = true;
        m->lazy_target = target_sum;
    }
    }
    (void)hipEventRecord(ev[1], ctx->stream);
    // per-gene moments of the transformed values (one pass, all-reduced across shards)
    if (rc == SRX_OK && !m->csc) {
        Range r_("srx:gene_moments");
        if (lazy) {
            rc = ensure_moments_xf(m, xf);
            if (xf.write_back && !m->lazy_pending) {
                // X now holds the transformed values (ensure_moments_xf did the bookkeeping of the two in-place calls as soon
                // as its kernel was in the stream, also when a later step of it failed): the passes below read them as stored
                xf = RowXf{};
                wrote_back = true;
            }
            if (rc == SRX_OK) m->moments_version = m->version;          // what pca_device's ensure_moments looks at
        } else rc = ensure_moments(m);
    }
    (void)hipEventRecord(ev[2], ctx->stream);
    // FeatureSelection::HighlyVariable(n_hvg): on the device, inside pca_device (the selection is fetched with the
    // other results once the solve is over); the host route only for shapes the device kernels do not take
    std::vector<uint64_t> sel;
    const bool dev_sel = !m->csc && n_hvg > 0 && take <= (uint64_t)kWave * KG && m->n_cols <= 65536;
    if (rc == SRX_OK && m->csc) {                 // HighlyVariable(n) with the CSC variance (csc.rs:164-177)
        uint64_t n_out = 0;
        sel.resize(n_hvg < m->n_rows ? n_hvg : m->n_rows);
        rc = srx_select_hvg(m, n_hvg, sel.data(), &n_out);
        sel.resize(rc == SRX_OK ? n_out : 0);
    } else if (rc == SRX_OK && !dev_sel) {
        std::vector<double> var;
        rc = gene_variances(m, var);
        if (rc == SRX_OK) rc = select_hvg_host(ctx, var, n_hvg, sel);
    }
    (void)hipEventRecord(ev[3], ctx->stream);
    if (rc == SRX_OK) {
        Range r_("srx:pca");
        rc = dev_sel ? pca_device(m, nullptr, 0, opts, n_hvg, xf) : pca_device(m, sel.data(), sel.size(), opts, 0, xf);
    }
    // whatever happened above, X ends up normalised and log1p'd (the two in-place calls come first in the reference)
    if (m->lazy_pending) {
        const int32_t rc_wb = launch_writeback(m);
        if (rc == SRX_OK) rc = rc_wb;
    }
    // the moments cached on the matrix are those of the f64 transform, not of the values as stored: retired (a later
    // compute_variance sees what X holds)
    if (wrote_back) m->moments_version = m->version - 1;
    (void)hipEventRecord(ev[4], ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    if (res) {
        memset(res, 0, sizeof *res);
        res->pca = m->pca.info;
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) res->ms_normalize = ms;
        if (hipEventElapsedTime(&ms, ev[1], ev[2]) == hipSuccess) res->ms_moments = ms;
        if (hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) res->ms_select = ms;
        if (hipEventElapsedTime(&ms, ev[3], ev[4]) == hipSuccess) res->ms_pca = ms;
        res->ms_compact = 0.0;
    }
    cleanup();
    return rc;
}

}  // extern "C"
