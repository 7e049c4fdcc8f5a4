// backed.inl — out-of-core ("backed") mode: the matrix lives in host memory or on disk and is visited as
// consecutive ROW TILES (src/backed/statistics/mod.rs:5-45 with ComputationMode::Chunked(size); the chunk loops
// src/shared/statistics/mod.rs:17-41,59-83 and csr.rs:48-74,112-143).  Included by pca.hip: the session reuses
// the compaction, Gram and transform launches of the resident path.
//
// The reference has the chunked form for compute_number / compute_sum only (and loses the row offset of a chunk
// in Direction::Row, csr.rs:126 — here every tile writes at its own global offset).  The same machinery carries
// the whole path: the only thing that has to stay in HBM for a 50-PC PCA over 2000 HVGs is the HVG-compacted
// matrix (7 % of the non-zeros, 8 B each) and the k x k Gram tiles, so a matrix many times larger than HBM is
// processed in two sweeps over its row tiles:
//
//   sweep 1, per tile:  H2D -> normalize_total(Row) + log1p -> per-gene (cnt, sum, sumsq) ADDED to the session
//   select:             one all-reduce of the moments, HighlyVariable(n) on the device, mean / std per gene
//   sweep 2, per tile:  H2D -> normalize + log1p -> compaction to the selected genes -> Gram tiles ADDED to the
//                       session; the tile's 256-tiled compacted view is kept (exact-size allocation)
//   solve:              one all-reduce of the Gram tiles, the subspace iteration on C, then the scores of every
//                       kept tile with one forward SpMM each
//
// Overlap: a tile is uploaded on the session's own stream with its own staging buffers while the kernels of the
// previous tile are still running on the context's stream; the previous tile is released after the upload.
// A row tile is an `srx_csr` whose `indptr` may be a WINDOW of the matrix's row-offset array (indptr[0] != 0;
// indices / values pointing at the tile's first entry).

struct srx_backed {
    srx_ctx* ctx = nullptr;
    uint64_t n_cols = 0;
    int32_t store_req = 0;
    int32_t store = -1;                 // resolved by the first tile
    hipStream_t up_stream = nullptr;
    srx_mat* acc = nullptr;             // header-only matrix: global moments, then the PCA result
    double* d_packed = nullptr;         // 3G+1: this rank's (cnt, sum, sumsq, rows) over the tiles of sweep 1
    srx_mat* prev = nullptr;            // tile whose kernels may still be running
    uint64_t rows1 = 0, nnz1 = 0, rows2 = 0, nnz_sel = 0;
    int phase = 0;                      // 0: sweep 1, 1: selected (sweep 2), 2: solved
    // selection
    bool dev_sel = false;
    srx::HvgDev hv;
    srx::Resolved o;
    int l_act = 0, k = 0;
    std::vector<uint64_t> selv;
    std::vector<int32_t> remap;
    std::vector<double> mu, sd, dinv;
    double trace = 0.0;
    // sweep 2
    std::vector<srx::RowMajor> parts;   // owned copies of the row-major compacted rows of every tile (the transform walks them)
    double* d_gram = nullptr;
    size_t n_packed = 0;
};

namespace srx {

int32_t row_number(srx_mat* m, uint32_t* out);                 // rows.hip
int32_t row_stat(srx_mat* m, int which, double* out0, double* out1);

static __global__ void k_acc_f64(double* __restrict__ acc, const double* __restrict__ x, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += x[i];
}

static void backed_release_prev(srx_backed* b) {
    if (b->prev) srx_matrix_free(b->prev);         // waits for the kernels of that tile
    b->prev = nullptr;
}

// upload on the side stream (overlaps the previous tile's kernels), then let go of the previous tile
static int32_t backed_upload(srx_backed* b, const srx_csr* tile, srx_mat** out) {
    srx_ctx* ctx = b->ctx;
    if (!tile || !tile->indptr || (tile->nnz && (!tile->indices || !tile->values)))
        return fail(ctx, SRX_E_ARG, "backed: null CSR tile");
    if (tile->n_cols != b->n_cols)
        return fail(ctx, SRX_E_SHAPE, "backed: tile has %llu columns, the matrix %llu", (unsigned long long)tile->n_cols,
                    (unsigned long long)b->n_cols);
    int32_t rc = upload_on(ctx, tile, b->store >= 0 ? b->store : b->store_req, b->up_stream, out);
    backed_release_prev(b);
    if (rc != SRX_OK) return rc;
    if (b->store < 0) b->store = (*out)->store;
    return SRX_OK;
}

// normalize_total(Row) + log1p together are applied ON THE FLY by the passes that read the tile (RowXf: f64-accurate
// values from the raw ones, nothing written back — the tile is dropped afterwards); either one alone is done in place.
static int32_t backed_transform(srx_mat* m, double target_sum, int32_t transform, RowXf& xf) {
    const bool do_norm = (transform & SRX_BACKED_NORMALIZE) != 0, do_log = (transform & SRX_BACKED_LOG1P) != 0;
    xf = RowXf{};
    if (!do_norm && !do_log) return SRX_OK;
    if (do_norm && do_log) {
        SRX_TRY(launch_row_sums(m));
        xf.row_sum = m->d_row_sum;
        xf.target = target_sum;
        return SRX_OK;
    }
    return launch_normalize(m, target_sum, do_norm, do_log);
}

static void free_parts(srx_backed* b) {
    for (auto& t : b->parts) {
        (void)hipFree(t.ptr);
        (void)hipFree(t.pk);
        (void)hipFree(t.perm);
    }
    b->parts.clear();
}

template <typename VT>
static int32_t backed_gram_tile(srx_backed* b, srx_mat* m, RowXf xf) {
    srx_ctx* ctx = b->ctx;
    RowMajor rm;
    if (b->dev_sel) SRX_TRY(build_tiled_fused(m, b->hv.d_bits, b->hv.n_words, b->k, rm, nullptr, xf, true));
    else SRX_TRY(build_tiled_fused(m, b->remap, b->k, rm, nullptr, xf, true));
    SRX_TRY(launch_gram<VT>(ctx, rm, b->d_gram));            // accumulates into the session's packed matrix
    if (!rm.perm) SRX_TRY(build_row_order(ctx, rm));          // (made inside the fused compaction where that one ran)
    // keep the row-major records of this tile for the transform: exact-size copies out of the scratch buffers
    RowMajor keep = rm;
    keep.ptr = nullptr;
    keep.pk = nullptr;
    keep.perm = nullptr;
    const size_t ptr_bytes = (rm.n_rows + 1) * sizeof(int64_t);
    const size_t pk_bytes = (rm.nnz + 64) * sizeof(GramPk<VT>);
    const size_t perm_bytes = (rm.n_rows ? rm.n_rows : 1) * sizeof(uint32_t);
    hipError_t e = hipMalloc((void**)&keep.ptr, ptr_bytes);
    if (e == hipSuccess) e = hipMalloc(&keep.pk, pk_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&keep.perm, perm_bytes);
    if (e != hipSuccess) {
        (void)hipFree(keep.ptr);
        (void)hipFree(keep.pk);
        (void)hipFree(keep.perm);
        return fail(ctx, SRX_E_OOM, "backed: keeping a compacted tile: %s", hipGetErrorString(e));
    }
    b->parts.push_back(keep);
    SRX_HIP(ctx, hipMemcpyAsync(keep.ptr, rm.ptr, ptr_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(keep.pk, rm.pk, pk_bytes - 64 * sizeof(GramPk<VT>), hipMemcpyDeviceToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemsetAsync((char*)keep.pk + pk_bytes - 64 * sizeof(GramPk<VT>), 0, 64 * sizeof(GramPk<VT>), ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(keep.perm, rm.perm, perm_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    b->nnz_sel += rm.nnz;
    return SRX_OK;
}

}  // namespace srx

extern "C" {

int32_t srx_backed_create(srx_ctx* ctx, uint64_t n_cols, int32_t store, srx_backed** out) {
    if (!ctx || !out) return fail(ctx, SRX_E_ARG, "srx_backed_create: null argument");
    *out = nullptr;
    if (n_cols >= (1ull << 31)) return fail(ctx, SRX_E_BOUNDS, "n_cols %llu exceeds int32 device indices", (unsigned long long)n_cols);
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    srx_backed* b = new srx_backed();
    b->ctx = ctx;
    ctx->pool_on += 1;             // the tiles' device buffers are recycled from here on (common.hpp)
    b->n_cols = n_cols;
    b->store_req = store;
    auto bail = [&](int32_t rc) { srx_backed_destroy(b); return rc; };
    if (hipStreamCreateWithFlags(&b->up_stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(ctx, SRX_E_HIP, "backed: stream creation failed"));
    b->acc = new srx_mat();
    b->acc->ctx = ctx;
    b->acc->n_cols = n_cols;
    const size_t n = 3 * n_cols + 1;
    if (hipMalloc((void**)&b->d_packed, n * sizeof(double)) != hipSuccess)
        return bail(fail(ctx, SRX_E_OOM, "backed: moment accumulators"));
    if (hipMemsetAsync(b->d_packed, 0, n * sizeof(double), ctx->stream) != hipSuccess)
        return bail(fail(ctx, SRX_E_HIP, "backed: memset"));
    *out = b;
    return SRX_OK;
}

void srx_backed_destroy(srx_backed* b) {
    if (!b) return;
    if (b->ctx) {
        (void)hipSetDevice(b->ctx->device);
        (void)hipStreamSynchronize(b->ctx->stream);
    }
    backed_release_prev(b);
    free_parts(b);
    (void)hipFree(b->d_packed);
    (void)hipFree(b->d_gram);
    if (b->acc) {
        (void)hipFree(b->acc->d_cnt);
        (void)hipFree(b->acc->d_sum);
        (void)hipFree(b->acc->d_sq);
        (void)hipFree(b->acc->pca.d_scores);
        delete b->acc;
    }
    if (b->up_stream) (void)hipStreamDestroy(b->up_stream);
    if (b->ctx && b->ctx->pool_on > 0 && --b->ctx->pool_on == 0) pool_clear(b->ctx);
    delete b;
}

int32_t srx_backed_stats_tile(srx_backed* b, const srx_csr* tile, double target_sum, int32_t transform,
                              uint32_t* row_number_out, double* row_sum_out) {
    if (!b) return fail(nullptr, SRX_E_ARG, "null session");
    srx_ctx* ctx = b->ctx;
    if (b->phase != 0) return fail(ctx, SRX_E_ARG, "backed: the statistics sweep is over (selection already made)");
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    srx_mat* m = nullptr;
    SRX_TRY(backed_upload(b, tile, &m));
    b->prev = m;
    // Direction::Row statistics of the RAW values belong to the tile alone: written at the caller's offset
    if (row_number_out) SRX_TRY(row_number(m, row_number_out));
    if (row_sum_out) SRX_TRY(row_stat(m, 0, row_sum_out, nullptr));
    RowXf xf;
    SRX_TRY(backed_transform(m, target_sum, transform, xf));
    SRX_TRY(moments_accumulate(m, b->d_packed, xf));
    b->rows1 += m->n_rows;
    b->nnz1 += m->nnz;
    return SRX_OK;
}

int32_t srx_backed_moments(srx_backed* b, uint64_t* cnt, double* sum, double* sumsq, uint64_t* n_rows_global) {
    if (!b) return fail(nullptr, SRX_E_ARG, "null session");
    srx_ctx* ctx = b->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    if (b->acc->moments_version != b->acc->version) {
        // the all-reduce works on a copy, so that more tiles may still be added afterwards
        double* tmp;
        const size_t n = 3 * b->n_cols + 1;
        SRX_TRY(scratch(ctx, "backed_packed", n * sizeof(double), (void**)&tmp));
        SRX_HIP(ctx, hipMemcpyAsync(tmp, b->d_packed, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        SRX_TRY(moments_install(b->acc, tmp));
        if (b->phase == 0) b->acc->moments_version = 0;       // sweep 1 still open: recompute on the next call
    }
    const uint64_t G = b->n_cols;
    if (cnt) SRX_TRY(d2h(ctx, cnt, b->acc->d_cnt, G * sizeof(uint64_t)));
    if (sum) SRX_TRY(d2h(ctx, sum, b->acc->d_sum, G * sizeof(double)));
    if (sumsq) SRX_TRY(d2h(ctx, sumsq, b->acc->d_sq, G * sizeof(double)));
    if (n_rows_global) *n_rows_global = b->acc->n_rows_global;
    return SRX_OK;
}

static int32_t backed_select_impl(srx_backed* b, uint64_t n_hvg, const uint64_t* sel, uint64_t n_sel, const srx_pca_opts* opts,
                                  uint64_t* sel_out, uint64_t* n_out);

int32_t srx_backed_select(srx_backed* b, uint64_t n_hvg, const uint64_t* sel, uint64_t n_sel, const srx_pca_opts* opts,
                          uint64_t* sel_out, uint64_t* n_out) {
    if (!b) return fail(nullptr, SRX_E_ARG, "null session");
    srx_ctx* ctx = b->ctx;
    if (b->phase != 0) return fail(ctx, SRX_E_ARG, b->phase < 0 ? "backed: the session failed during the selection; start a new one"
                                                                 : "backed: the selection was already made");
    if (b->store < 0) return fail(ctx, SRX_E_ARG, "backed: no tile was given to the statistics sweep");
    // the moment accumulators are all-reduced in place: a selection that fails half-way cannot be repeated
    b->phase = -1;
    const int32_t rc = backed_select_impl(b, n_hvg, sel, n_sel, opts, sel_out, n_out);
    if (rc == SRX_OK) b->phase = 1;
    return rc;
}

static int32_t backed_select_impl(srx_backed* b, uint64_t n_hvg, const uint64_t* sel, uint64_t n_sel, const srx_pca_opts* opts,
                                  uint64_t* sel_out, uint64_t* n_out) {
    srx_ctx* ctx = b->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    backed_release_prev(b);
    srx_mat* acc = b->acc;
    acc->store = b->store;
    acc->version = 1;
    SRX_TRY(moments_install(acc, b->d_packed));
    const uint64_t G = b->n_cols;
    const uint64_t take = n_hvg < G ? n_hvg : G;
    b->dev_sel = n_hvg > 0 && take <= (uint64_t)kWave * KG && G <= 65536;
    if (n_hvg > 0 && !b->dev_sel) {                       // HighlyVariable(n) on the host (wide matrices)
        std::vector<double> var;
        SRX_TRY(gene_variances(acc, var));
        SRX_TRY(select_hvg_host(ctx, var, n_hvg, b->selv));
    } else if (n_hvg == 0) {
        if (sel) b->selv.assign(sel, sel + n_sel);
        else { b->selv.resize(G); std::iota(b->selv.begin(), b->selv.end(), 0ull); }
    } else {
        b->selv.resize(take);                              // filled from the device once the solve is over
    }
    const int k = (int)b->selv.size();
    b->k = k;
    if ((k + KG - 1) / KG > kWave)
        return fail(ctx, SRX_E_ARG, "backed: %d selected features exceed the %d the fused compaction takes", k, kWave * KG);
    // (the scores are written by the row-major forward kernel: any k — beyond its widest LDS slice it walks gene ranges)
    SRX_TRY(resolve_opts(ctx, opts, k, acc->n_rows_global, b->store == SRX_STORE_F32, b->o, b->l_act));
    if (opts && opts->solver == 2) return fail(ctx, SRX_E_ARG, "backed: only the Gram solver works on row tiles");
    if (b->o.solver != 1) { b->o.solver = 1; b->o.power = 3; b->o.warm = 2; }
    b->mu.assign(k, 0.0); b->sd.assign(k, 0.0); b->dinv.assign(k, 0.0);
    b->trace = 0.0;
    if (b->dev_sel) {
        SRX_TRY(select_hvg_device(acc, n_hvg, b->o.center, b->o.scale, b->hv));
    } else {
        std::vector<int> order(k), slot_of_sel(k);
        SRX_TRY(prepare_host_selection(acc, b->selv, b->o, order, slot_of_sel, b->remap, b->mu, b->sd, b->dinv, b->trace));
    }
    b->n_packed = gram_packed_count(k);
    SRX_HIP(ctx, hipMalloc((void**)&b->d_gram, b->n_packed * sizeof(double)));
    SRX_HIP(ctx, hipMemsetAsync(b->d_gram, 0, b->n_packed * sizeof(double), ctx->stream));
    if (n_out) *n_out = (uint64_t)k;
    if (sel_out) {
        if (b->dev_sel) {
            std::vector<int32_t> r(k);
            SRX_TRY(d2h(ctx, r.data(), b->hv.d_sel_rank, (size_t)k * sizeof(int32_t)));
            int st = 0;
            SRX_TRY(d2h(ctx, &st, b->hv.d_status, sizeof(int)));
            if (st & 1) return fail(ctx, SRX_E_NAN, "NaN gene variance: called `Option::unwrap()` on a `None` value (partial_cmp)");
            for (int i = 0; i < k; ++i) sel_out[i] = (uint64_t)r[i];
        } else {
            memcpy(sel_out, b->selv.data(), (size_t)k * 8);
        }
    }
    return SRX_OK;
}

int32_t srx_backed_gram_tile(srx_backed* b, const srx_csr* tile, double target_sum, int32_t transform) {
    if (!b) return fail(nullptr, SRX_E_ARG, "null session");
    srx_ctx* ctx = b->ctx;
    if (b->phase != 1) return fail(ctx, SRX_E_ARG, "backed: the Gram sweep comes after srx_backed_select and before the solve");
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    srx_mat* m = nullptr;
    SRX_TRY(backed_upload(b, tile, &m));
    b->prev = m;
    if (m->store != b->store) return fail(ctx, SRX_E_DTYPE, "backed: tile storage differs from the first tile's");
    if (m->n_rows == 0) return SRX_OK;
    RowXf xf;
    SRX_TRY(backed_transform(m, target_sum, transform, xf));
    if (is_f32(m)) SRX_TRY(backed_gram_tile<float>(b, m, xf));
    else SRX_TRY(backed_gram_tile<double>(b, m, xf));
    b->rows2 += m->n_rows;
    return SRX_OK;
}

int32_t srx_backed_solve(srx_backed* b, srx_pca_info* info) {
    if (!b) return fail(nullptr, SRX_E_ARG, "null session");
    srx_ctx* ctx = b->ctx;
    if (b->phase != 1) return fail(ctx, SRX_E_ARG, "backed: nothing to solve (select first, then the Gram sweep)");
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    backed_release_prev(b);
    if (b->rows2 != b->rows1)
        return fail(ctx, SRX_E_SHAPE, "backed: the Gram sweep saw %llu rows, the statistics sweep %llu", (unsigned long long)b->rows2,
                    (unsigned long long)b->rows1);
    if (b->parts.empty()) return fail(ctx, SRX_E_SHAPE, "backed: this rank holds no rows");
    srx_mat* acc = b->acc;
    srx_pca_state& st = acc->pca;
    st.valid = false;
    const int k = b->k;
    std::vector<double> mu_eff = b->mu;
    if (!b->o.center) std::fill(mu_eff.begin(), mu_eff.end(), 0.0);
    st.info = srx_pca_info{};
    st.info.n_cells_global = acc->n_rows_global;
    st.info.k = (uint32_t)k;
    st.info.n_pc = (uint32_t)b->o.n_pc;
    st.info.block = L;
    st.info.nnz_selected = b->nnz_sel;
    st.info.solver = 1;
    const HvgDev* hvp = b->dev_sel ? &b->hv : nullptr;
    const double Nd = (double)acc->n_rows_global;
    int32_t rc;
    if (b->store == SRX_STORE_F32)
        rc = run_pca<float, float>(ctx, b->parts.data(), (int)b->parts.size(), nullptr, b->d_gram, b->o, mu_eff, b->dinv, hvp,
                                   b->l_act, Nd, st);
    else
        rc = run_pca<double, double>(ctx, b->parts.data(), (int)b->parts.size(), nullptr, b->d_gram, b->o, mu_eff, b->dinv,
                                     hvp, b->l_act, Nd, st);
    b->phase = 2;
    if ((rc != SRX_OK && rc != SRX_E_NOCONV) || !st.d_small) return rc;
    SRX_TRY(stash_results(ctx, st, k, b->o.n_pc, hvp, b->mu, b->sd, b->trace, b->selv));
    acc->n_rows = b->rows2;                      // what srx_result_fetch sizes the scores by
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_parts(b);
    if (info) *info = st.info;
    return rc;
}

int32_t srx_backed_fetch(srx_backed* b, double* scores, double* components, double* evr, double* mean, double* std_,
                         uint64_t* sel) {
    if (!b) return fail(nullptr, SRX_E_ARG, "null session");
    if (b->phase != 2) return fail(b->ctx, SRX_E_ARG, "backed: no PCA result yet");
    return srx_result_fetch(b->acc, scores, components, evr, mean, std_, sel);
}

}  // extern "C"
