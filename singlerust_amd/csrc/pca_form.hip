// pca_form.hip — the HVG-compacted matrix and its Gram matrix (first half of dim_red::pca_inplace on the GPU; pca.hip has the
// overview): the compaction of X to the selected features (compact.inl: counts, scans, fill passes; row-major entries and,
// where a consumer wants it, a tile-major view), the owner records of the Gram kernel and G = A^T A itself (gram.inl), with
// the exchange of the packed triangle across row shards.
#include "pca_internal.hpp"
#include "log1p64.hpp"

namespace srx {

#include "compact.inl"

#include "gram.inl"

// The part of the plan that depends on k alone: stripe height, stripes, owners.
static void gram_stripes_of(int k, GramPlan& g) {
    g.k = k;
    // the largest stripe height whose two stripes fit 64 KiB (two workgroups per CU); one row per stripe up to 160 KiB
    // (c3, k = 2000: SR = 4 — two workgroups per CU — 3.9 ms; SR = 8, one workgroup per CU: 4.29-4.47; SR = 2: 4.17)
    int sr = 8;
    while (sr > 1 && (size_t)sr * (size_t)(k + sr) * 8 > 65536) sr >>= 1;
    g.sr_shift = sr == 8 ? 3 : sr == 4 ? 2 : sr == 2 ? 1 : 0;
    g.n_stripes = (k + sr - 1) / sr;
    g.n_stripes += g.n_stripes & 1;
    g.n_wg = g.n_stripes / 2;
}
// Sharded rows: the packed triangle crosses the ranks in three pieces (launch_gram).  Rows [0, r_lo) and [r_hi, k) belong to
// the first half of the owners, the rows between to the second half.
static void gram_exchange_rows(const GramPlan& g, int& r_lo, int& r_hi) {
    const int SR = 1 << g.sr_shift, h = g.n_wg / 2;
    r_lo = std::min(g.k, h * SR);
    r_hi = std::min(g.k, std::max(r_lo, (g.n_stripes - h) * SR));
}
static size_t packed_row_offset(int row, int k) { return (size_t)row * (size_t)k - (size_t)row * (size_t)(row - 1) / 2; }      // of (row, row)

int32_t gram_plan(srx_ctx* ctx, int k, uint64_t n_rows, GramPlan& g, double entries_per_cell, int entry_bytes) {
    gram_stripes_of(k, g);
    const int sr = 1 << g.sr_shift;
    size_t widest = 0;
    for (int w = 0; w < g.n_wg; ++w) {
        const int a0 = w * sr, b0 = (g.n_stripes - 1 - w) * sr;
        const size_t wd = (size_t)(k - a0) + (size_t)(k - b0 > 0 ? k - b0 : 0);
        widest = std::max(widest, wd);
    }
    g.lds_bytes = (size_t)sr * widest * 8;
    if (g.lds_bytes > 163840) return fail(ctx, SRX_E_ARG, "pca: %d selected features exceed the Gram kernel's LDS stripes", k);
    g.rblk = 512u;      // (round 6, assembly core: 256 / 384 / 512 / 1024 cells: stripes 2.55 / 2.55 / 2.47 / 2.49, records 0.89 / 0.87 / 0.89 / 1.07 ms) c3: bucket pass + stripe kernel 4.89 ms with 1024-cell blocks, 4.78 with 512, 5.07 with 256
    // an owner record names its piece by a 32-bit BYTE offset from its row block's first entry (a row holds at most k entries of at
    // most 16 bytes): 512 x 16384 x 16 = 2^27 — asserted, not assumed
    if ((uint64_t)g.rblk * (uint64_t)k * 16u >= (1ull << 31))
        return fail(ctx, SRX_E_ARG, "pca: %d selected features exceed the owner records' 32-bit offsets", k);
    g.n_rblk = (n_rows + g.rblk - 1) / g.rblk;
    const int per_cu = g.lds_bytes <= 65536 ? 2 : 1;
    // chunks of consecutive row blocks, at least one block per wave and enough chunks to fill the device.  Round 5, with the assembly
    // core, c3 / f32 (72 entries of 8 bytes per cell), cells per chunk: 8k 3.35 ms, 12k 2.90, 16k 2.65, 20k 2.54, 24k 2.48, 28k 2.46, 32k
    // 2.45, 36k 2.54, 40k 2.65, 48k 2.91, 64k 3.34, 128k 3.96: every (owner, chunk) workgroup flushes its 8000 sums with global atomics
    // — fewer chunks, fewer flushes — until the owners of a chunk drift apart in it and the L2 stops serving one's reads from another's.
    // What a chunk should hold is WORK, ~9e7 scalar products (32k cells of 72 kept entries): under skewed gene densities (147 entries
    // per cell, owners of very different weight: they drift apart sooner) 8k cells take 7.4 ms where 16k took 10.7 and 32k 12.5; and no
    // more than ~19 MB of entries (f64 storage, 16-byte entries: 16k cells 4.4 ms, 32k 5.8).  (Rounds 2-4, VALU-bound kernels: flat
    // from 16k to 64k cells, 16k kept.)
    uint64_t chunk = 16384 / g.rblk;
    if (entries_per_cell > 0.0 && entry_bytes > 0) {
        const double m = entries_per_cell;
        double cells = 9.0e7 / (0.5 * m * (m + 1.0));
        cells = std::min(cells, 18.9e6 / (m * (double)entry_bytes));
        chunk = (uint64_t)(cells / g.rblk + 0.5);
        if (chunk > 65536 / g.rblk) chunk = 65536 / g.rblk;
    }
    // the fixed-point accumulators of a chunk: at most 2^16 products below 2^47 each (f64 entries; f32: below 2^31) in 64 bits, read
    // back unsigned — the cap is part of that bound, whoever changes the sizing above
    static_assert(65536ull * (1ull << 47) <= (1ull << 63), "chunk cap x product bound must fit the accumulator");
    if (chunk * g.rblk > 65536) chunk = 65536 / g.rblk;
    chunk = std::max<uint64_t>(chunk, kGramWaves);
    const uint64_t want_wgs = (uint64_t)ctx->n_cus * per_cu;
    while (chunk > kGramWaves && ((g.n_rblk + chunk - 1) / chunk) * (uint64_t)g.n_wg < want_wgs) chunk /= 2;
    g.n_chunk = (uint32_t)chunk;
    int z = (int)((g.n_rblk + chunk - 1) / chunk);
    if (z < 1) z = 1;
    g.n_z = z;
    return SRX_OK;
}
int grid_rows(const srx_ctx* ctx, uint64_t n_rows, int rows_per_block) {
    uint64_t want = (n_rows + rows_per_block - 1) / rows_per_block;
    uint64_t cap = (uint64_t)ctx->n_cus * 8;
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

// out[0..n] = exclusive scan of in[0..n), out[n] = total (also left in *total_dev).
int32_t scan_exclusive(srx_ctx* ctx, const int64_t* d_in, uint64_t n, int64_t* d_out, int64_t** total_dev) {
    const uint64_t per_block = (uint64_t)kScanBlock * kScanItems;
    const uint64_t nb = (n + per_block - 1) / per_block > 0 ? (n + per_block - 1) / per_block : 1;
    int64_t* d_bsum;
    SRX_TRY(scratch(ctx, "scan_bsum", (nb + 1) * sizeof(int64_t), (void**)&d_bsum));
    int64_t* d_total = d_bsum + nb;
    hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(kScanBlock), 0, ctx->stream, d_in, n, d_bsum);
    hipLaunchKernelGGL(k_scan_serial, dim3(1), dim3(kScanBlock), 0, ctx->stream, d_bsum, nb, d_total);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(kScanBlock), 0, ctx->stream, d_in, n, d_bsum, d_total,
                       d_out);
    SRX_HIP(ctx, hipGetLastError());
    if (total_dev) *total_dev = d_total;
    return SRX_OK;
}

// (idx, vals) of a compacted CSR -> packed row-major records
template <typename T>
__global__ void k_pack_records(const int32_t* __restrict__ idx, const T* __restrict__ vals, uint64_t n,
                               GramPk<T>* __restrict__ out) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; e < n; e += stride) {
        GramPk<T> r{};
        r.j = idx[e];
        r.v = vals[e];
        out[e] = r;
    }
}

// X[:, sel] -> row-major compacted CSR (count, scan, fill); columns renumbered by `remap`.  General route
// (more than 8192 selected features); `rm` receives the packed-record view of it.
int32_t build_compact(srx_mat* m, const std::vector<int32_t>& remap, int k, CompactCsr& c, RowMajor& rm) {
    srx_ctx* ctx = m->ctx;
    const uint64_t N = m->n_rows;
    int32_t* d_remap;
    int64_t *d_counts, *d_total;
    SRX_TRY(scratch(ctx, "pca_remap", (remap.size() ? remap.size() : 1) * sizeof(int32_t), (void**)&d_remap));
    SRX_TRY(h2d(ctx, d_remap, remap.data(), remap.size() * sizeof(int32_t)));
    SRX_TRY(scratch(ctx, "pca_counts", (N ? N : 1) * sizeof(int64_t), (void**)&d_counts));
    SRX_TRY(scratch(ctx, "pca_rm_ptr", (N + 1) * sizeof(int64_t), (void**)&c.indptr));
    const double in_bytes = (double)m->nnz * 4.0 * 2.0 + (double)(N + 1) * 8.0 * 2.0;   // idx read by count + fill
    ProfScope ps(ctx, SRX_K_COMPACT, in_bytes);
    hipLaunchKernelGGL(k_compact_count, dim3(grid_rows(ctx, N, 4)), dim3(256), 0, ctx->stream, m->d_indptr,
                       m->d_indices, d_remap, N, d_counts);
    SRX_TRY(scan_exclusive(ctx, d_counts, N, c.indptr, &d_total));
    int64_t total = 0;
    SRX_TRY(d2h(ctx, &total, d_total, sizeof(int64_t)));
    c.nnz = (uint64_t)total;
    c.n_rows = N;
    c.k = k;
    const size_t vb = val_bytes(m);
    SRX_TRY(scratch(ctx, "pca_cidx", (c.nnz ? c.nnz : 1) * sizeof(int32_t), (void**)&c.idx));
    SRX_TRY(scratch(ctx, "pca_cvals", (c.nnz ? c.nnz : 1) * vb, &c.vals));
    const size_t pb = vb == 4 ? sizeof(GramPk<float>) : sizeof(GramPk<double>);
    SRX_TRY(scratch(ctx, "pca_rm_pk", (c.nnz + 64) * pb, &rm.pk));
    // the 64 entries behind the last row are READ by the forward kernel (a row's last chunk runs past its end: value masked
    // to 0, column used as is): they must name a real column, or 0 x panel[garbage] is NaN
    SRX_HIP(ctx, hipMemsetAsync((char*)rm.pk + c.nnz * pb, 0, 64 * pb, ctx->stream));
    const unsigned pg = (unsigned)std::min<uint64_t>((c.nnz + 255) / 256 + 1, 65536);
    if (is_f32(m)) {
        hipLaunchKernelGGL((k_compact_fill<float>), dim3(grid_rows(ctx, N, 4)), dim3(256), 0, ctx->stream, m->d_indptr,
                           m->d_indices, (const float*)m->d_values, d_remap, N, c.indptr, c.idx, (float*)c.vals);
        hipLaunchKernelGGL((k_pack_records<float>), dim3(pg), dim3(256), 0, ctx->stream, c.idx, (const float*)c.vals, c.nnz,
                           (GramPk<float>*)rm.pk);
    } else {
        hipLaunchKernelGGL((k_compact_fill<double>), dim3(grid_rows(ctx, N, 4)), dim3(256), 0, ctx->stream,
                           m->d_indptr, m->d_indices, (const double*)m->d_values, d_remap, N, c.indptr, c.idx,
                           (double*)c.vals);
        hipLaunchKernelGGL((k_pack_records<double>), dim3(pg), dim3(256), 0, ctx->stream, c.idx, (const double*)c.vals, c.nnz,
                           (GramPk<double>*)rm.pk);
    }
    SRX_HIP(ctx, hipGetLastError());
    rm.n_rows = N;
    rm.nnz = c.nnz;
    rm.k = k;
    rm.ptr = c.indptr;
    if (ctx->prof_mask & (1u << SRX_K_COMPACT)) ctx->prof[SRX_K_COMPACT].bytes += (double)c.nnz * (4.0 + vb) * 3.0;
    return SRX_OK;
}

// Tile-major copy of a compacted CSR for gene tiles of kt columns (cut, scan, copy).
int32_t retile(srx_mat* m, const CompactCsr& c, int kt, Tiled& t) {
    srx_ctx* ctx = m->ctx;
    const uint64_t N = c.n_rows;
    const size_t vb = val_bytes(m);
    const std::string tag = "pca_t" + std::to_string(kt) + "_";
    t.n_rows = N;
    t.nnz = c.nnz;
    t.k = c.k;
    t.kt = kt;
    t.nt = (c.k + kt - 1) / kt;
    int64_t* d_tp = nullptr;
    if (t.nt > 1) {
        SRX_TRY(scratch(ctx, "pca_tp", (size_t)(t.nt - 1) * (N ? N : 1) * sizeof(int64_t), (void**)&d_tp));
        SRX_TRY(launch_tile_ptr(ctx, c.indptr, c.idx, N, t.nt, kt, d_tp));
    }
    const uint64_t nseg = (uint64_t)t.nt * N;
    int64_t* d_seglen;
    SRX_TRY(scratch(ctx, "pca_seglen", (nseg ? nseg : 1) * sizeof(int64_t), (void**)&d_seglen));
    SRX_TRY(scratch(ctx, (tag + "ptr").c_str(), (nseg + 1) * sizeof(int64_t), (void**)&t.tptr));
    uint64_t g = (nseg + 255) / 256;
    if (g < 1) g = 1;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(k_seglen, dim3((unsigned)g), dim3(256), 0, ctx->stream, c.indptr, d_tp, N, t.nt, d_seglen);
    SRX_TRY(scan_exclusive(ctx, d_seglen, nseg, t.tptr, nullptr));
    // +64 records of padding: the forward kernel reads 16-wide chunks unconditionally
    const size_t pb = vb == 4 ? sizeof(GramPk<float>) : sizeof(GramPk<double>);
    SRX_TRY(scratch(ctx, (tag + "pk").c_str(), (c.nnz + 64) * pb, &t.tpk));
    SRX_HIP(ctx, hipMemsetAsync((char*)t.tpk + c.nnz * pb, 0, 64 * pb, ctx->stream));
    if (is_f32(m))
        hipLaunchKernelGGL((k_retile<float>), dim3(grid_rows(ctx, N, 4)), dim3(256), 0, ctx->stream, c.indptr, d_tp, c.idx,
                           (const float*)c.vals, N, t.nt, kt, t.tptr, (GramPk<float>*)t.tpk);
    else
        hipLaunchKernelGGL((k_retile<double>), dim3(grid_rows(ctx, N, 4)), dim3(256), 0, ctx->stream, c.indptr, d_tp,
                           c.idx, (const double*)c.vals, N, t.nt, kt, t.tptr, (GramPk<double>*)t.tpk);
    SRX_HIP(ctx, hipGetLastError());
    return SRX_OK;
}

// Fast path: the 256-tiled layout and the row-major records straight from X (count, two scans, fill); needs
// <= 64 tiles of 128 columns (k <= 8192) because lane t of a wave is the counter of tile t.
static int32_t alloc_tiled(srx_mat* m, uint64_t N, uint64_t nnz, int k, int kt, Tiled& t) {
    srx_ctx* ctx = m->ctx;
    const size_t vb = val_bytes(m);
    const std::string tag = "pca_t" + std::to_string(kt) + "_";
    t.n_rows = N;
    t.nnz = nnz;
    t.k = k;
    t.kt = kt;
    t.nt = (k + kt - 1) / kt;
    const size_t pb = vb == 4 ? sizeof(GramPk<float>) : sizeof(GramPk<double>);
    SRX_TRY(scratch(ctx, (tag + "pk").c_str(), (nnz + 64) * pb, &t.tpk));
    SRX_HIP(ctx, hipMemsetAsync((char*)t.tpk + nnz * pb, 0, 64 * pb, ctx->stream));
    return SRX_OK;
}

// `d_sel`: n_words selection bits followed by n_words prefix counts, on the device (the compacted column of a
// gene is its rank among the selected genes in ascending gene order)
int32_t build_tiled_fused(srx_mat* m, const uint32_t* d_sel, int n_words, int k, RowMajor& rm, Tiled* t256p, RowXf xf, bool want_recs) {
    Tiled t256_dummy;
    Tiled& t256 = t256p ? *t256p : t256_dummy;          // the 256-tiled view is only made for the matrix-free solver
    srx_ctx* ctx = m->ctx;
    const uint64_t N = m->n_rows;
    const int nt128 = (k + KG - 1) / KG, nt256 = t256p ? (k + KT - 1) / KT : 0;
    int64_t *cntrow, *cnt256, *d_total;
    const size_t sel_lds = 2 * (size_t)n_words * sizeof(uint32_t);
    if (sel_lds > 60000) return fail(ctx, SRX_E_ARG, "pca: %llu genes exceed the LDS selection table", (unsigned long long)m->n_cols);
    const uint64_t n256 = (uint64_t)nt256 * N;
    SRX_TRY(scratch(ctx, "pca_cntrow", (N ? N : 1) * sizeof(int64_t), (void**)&cntrow));
    SRX_TRY(scratch(ctx, "pca_cnt256", (n256 ? n256 : 1) * sizeof(int64_t), (void**)&cnt256));
    SRX_TRY(scratch(ctx, "pca_rm_ptr", (N + 1) * sizeof(int64_t), (void**)&rm.ptr));
    SRX_TRY(scratch(ctx, "pca_t256_ptr", (n256 + 1) * sizeof(int64_t), (void**)&t256.tptr));
    // (A single-pass form — count, decoupled look-back over groups of 8 rows, fill from the indices still in L2; the output
    //  size known beforehand from the cached per-gene counts — was built and measured in round 3: 3.0 ms against 1.73 for
    //  count + scan + fill.  The groups have to stay small for the second walk to hit L2 (16 KB of L2 per resident
    //  workgroup), and 162 500 groups make the prefix chain the bound: 64 groups per ~1.5 us hop.  It also needs every wave
    //  of the grid resident, which the occupancy query over-promised at 8 workgroups per CU.  Not kept.)
    const double s_i = m->d_idx16 ? 2.0 : 4.0;          // bytes per column index streamed by the passes
    const size_t pb = is_f32(m) ? sizeof(GramPk<float>) : sizeof(GramPk<double>);
    // algorithmic bytes: the column indices of the whole matrix once per pass (count, fill) + row pointers in, row pointers
    // out; the KEPT values read and the compacted entries written are added below, once their number is known
    ProfScope ps(ctx, SRX_K_COMPACT, (double)m->nnz * s_i * ((!t256p && m->n_cols <= 65536) ? 1.0 : 2.0) +
                                         (double)(N + 1) * 8.0 * 2.0);
    const size_t cnt_lds = sel_lds + 4 * (size_t)kCompactRows * kWave * sizeof(uint32_t);      // + 8 x 64 counters per wave
    const bool list = !t256p && m->n_cols <= 65536 && m->d_idx16;      // (the 16-bit index mirror: there whenever n_cols <= 65536, ensure_tiles)
    uint32_t* kept = nullptr;
    if (list) {
        SRX_TRY(scratch(ctx, "pca_keptlist", (m->nnz + 64) * sizeof(uint32_t), (void**)&kept));
        // one 1024-thread workgroup per 16-bit gene table (2 bytes per gene + the entry for the lanes behind a row's end): two per
        // CU at 28k genes, one at 65536
        const size_t col_lds = ((size_t)n_words * 32 + 1) * sizeof(uint16_t);
        const int per_cu = std::max(1, std::min(2, (int)(160 * 1024 / (col_lds + 1024))));
        const uint64_t visits = (N + (uint64_t)kCompactRows * (kCountThreads / kWave) - 1) / ((uint64_t)kCompactRows * (kCountThreads / kWave));
        const unsigned cgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(visits, (uint64_t)ctx->n_cus * per_cu));
        SRX_HIP(ctx, hipFuncSetAttribute((const void*)k_rowcount_list, hipFuncAttributeMaxDynamicSharedMemorySize, (int)col_lds));
        hipLaunchKernelGGL(k_rowcount_list, dim3(cgrid), dim3(kCountThreads), col_lds, ctx->stream, m->d_indptr,
                           (const uint16_t*)m->d_idx16, d_sel, d_sel + n_words, n_words, N, m->nnz, k, cntrow, kept);
    } else if (!t256p) {
        const size_t bits_lds = (size_t)n_words * sizeof(uint32_t);
        if (m->d_idx16)
            hipLaunchKernelGGL((k_rowcount<uint16_t>), dim3(grid_rows(ctx, N, 4)), dim3(256), bits_lds, ctx->stream, m->d_indptr,
                               (const uint16_t*)m->d_idx16, d_sel, n_words, N, cntrow);
        else
            hipLaunchKernelGGL((k_rowcount<int32_t>), dim3(grid_rows(ctx, N, 4)), dim3(256), bits_lds, ctx->stream, m->d_indptr,
                               (const int32_t*)m->d_indices, d_sel, n_words, N, cntrow);
    } else if (m->d_idx16)
        hipLaunchKernelGGL((k_tcount<uint16_t>), dim3(grid_rows(ctx, N, 4)), dim3(256), cnt_lds, ctx->stream, m->d_indptr,
                           (const uint16_t*)m->d_idx16, d_sel, d_sel + n_words, n_words, N, nt128, nt256, k, cntrow, cnt256);
    else
        hipLaunchKernelGGL((k_tcount<int32_t>), dim3(grid_rows(ctx, N, 4)), dim3(256), cnt_lds, ctx->stream, m->d_indptr,
                           (const int32_t*)m->d_indices, d_sel, d_sel + n_words, n_words, N, nt128, nt256, k, cntrow, cnt256);
    SRX_TRY(scan_exclusive(ctx, cntrow, N, rm.ptr, &d_total));
    // The Gram kernel's record counts only need the row lengths: made HERE, before the read-back of the compacted size, so
    // that both numbers come back behind ONE drain of the stream (the second wait cost ~90 us of idle device per step)
    int64_t* d_nrecs = nullptr;
    rm.n_recs = -1;
    if (want_recs && N > 0) {
        GramPlan g;
        SRX_TRY(gram_plan(ctx, k, N, g));
        int64_t *blk_total, *rec_base;
        SRX_TRY(scratch(ctx, "pca_brtot", g.n_rblk * sizeof(int64_t), (void**)&blk_total));
        SRX_TRY(scratch(ctx, "pca_brbase", (g.n_rblk + 4) * sizeof(int64_t), (void**)&rec_base));
        hipLaunchKernelGGL(k_rec_count, dim3((unsigned)g.n_rblk), dim3(256), 0, ctx->stream, (const int64_t*)rm.ptr, N, g.rblk, blk_total);
        hipLaunchKernelGGL(k_rec_scan, dim3(1), dim3(1024), 0, ctx->stream, (const int64_t*)blk_total, g.n_rblk, rec_base,
                           (const int64_t*)d_total);
        SRX_HIP(ctx, hipGetLastError());
        d_nrecs = rec_base + g.n_rblk;
    }
    int64_t total = 0;
    {                                         // {records, compacted size}: ONE copy (the scan kernel put the second number next to the first)
        int64_t both[2] = {0, 0};
        SRX_TRY(d2h_begin(ctx, d_nrecs ? (const void*)d_nrecs : (const void*)d_total, d_nrecs ? sizeof both : sizeof(int64_t)));
        // the order of the rows by length (what the transform walks) needs the row pointers only: queued HERE, it runs while
        // the host is woken by the copy (the device used to idle through that round trip, ~30 us per step)
        if (!t256p) {
            rm.n_rows = N;
            SRX_TRY(build_row_order(ctx, rm));
        }
        SRX_TRY(d2h_end(ctx, both, d_nrecs ? sizeof both : sizeof(int64_t)));
        if (d_nrecs) {
            rm.n_recs = both[0];
            total = both[1];
        } else {
            total = both[0];
        }
    }
    if (t256p) {
        SRX_TRY(scan_exclusive(ctx, cnt256, n256, t256.tptr, nullptr));
        SRX_TRY(alloc_tiled(m, N, (uint64_t)total, k, KT, t256));
    }
    SRX_TRY(scratch(ctx, "pca_rm_pk", ((size_t)total + 64) * pb, &rm.pk));
    // (the 64 entries behind the last row, read by the forward kernel: zeroed by the fill kernel — the list route's by its
    //  first workgroup, the others by a memset)
    if (!list) SRX_HIP(ctx, hipMemsetAsync((char*)rm.pk + (size_t)total * pb, 0, 64 * pb, ctx->stream));
    rm.n_rows = N;
    rm.nnz = (uint64_t)total;
    rm.k = k;
    auto fill = [&](auto kern, const auto* idxp, const auto* valp, auto* rmp, auto* pk256) {
        hipLaunchKernelGGL(kern, dim3(grid_rows(ctx, N, 4)), dim3(256), sel_lds, ctx->stream, m->d_indptr, idxp, valp, d_sel,
                           d_sel + n_words, n_words, N, nt256, k, cnt256, rm.ptr, t256.tptr, xf.row_sum, xf.target, rmp, pk256);
    };
    auto fill_t = [&](auto tval, auto* rmp, auto* pk256) {
        using T = decltype(tval);
        const T* valp = (const T*)m->d_values;
        if (m->d_idx16) {
            const uint16_t* ip = (const uint16_t*)m->d_idx16;
            if (xf.row_sum) fill(k_tfill<T, uint16_t, true>, ip, valp, rmp, pk256);
            else fill(k_tfill<T, uint16_t, false>, ip, valp, rmp, pk256);
        } else {
            const int32_t* ip = (const int32_t*)m->d_indices;
            if (xf.row_sum) fill(k_tfill<T, int32_t, true>, ip, valp, rmp, pk256);
            else fill(k_tfill<T, int32_t, false>, ip, valp, rmp, pk256);
        }
    };
    if (list) {
        const unsigned g2 = (unsigned)grid_rows(ctx, N, 8);
        if (is_f32(m)) {
            if (xf.row_sum) hipLaunchKernelGGL((k_tfill_list<float, true>), dim3(g2), dim3(256), 0, ctx->stream, m->d_indptr, (const float*)m->d_values, kept, N, rm.ptr, xf.row_sum, xf.target, (GramPk<float>*)rm.pk);
            else hipLaunchKernelGGL((k_tfill_list<float, false>), dim3(g2), dim3(256), 0, ctx->stream, m->d_indptr, (const float*)m->d_values, kept, N, rm.ptr, xf.row_sum, xf.target, (GramPk<float>*)rm.pk);
        } else {
            if (xf.row_sum) hipLaunchKernelGGL((k_tfill_list<double, true>), dim3(g2), dim3(256), 0, ctx->stream, m->d_indptr, (const double*)m->d_values, kept, N, rm.ptr, xf.row_sum, xf.target, (GramPk<double>*)rm.pk);
            else hipLaunchKernelGGL((k_tfill_list<double, false>), dim3(g2), dim3(256), 0, ctx->stream, m->d_indptr, (const double*)m->d_values, kept, N, rm.ptr, xf.row_sum, xf.target, (GramPk<double>*)rm.pk);
        }
    } else if (is_f32(m)) fill_t(float{}, (GramPk<float>*)rm.pk, (GramPk<float>*)t256.tpk);
    else fill_t(double{}, (GramPk<double>*)rm.pk, (GramPk<double>*)t256.tpk);
    SRX_HIP(ctx, hipGetLastError());
    if (ctx->prof_mask & (1u << SRX_K_COMPACT))
    {
        ctx->prof[SRX_K_COMPACT].bytes += (double)total * (val_bytes(m) + (double)pb * (t256p ? 2.0 : 1.0));   // kept values read, entries written once or twice
        if (list) ctx->prof[SRX_K_COMPACT].aux_bytes += (double)total * 4.0 * 2.0;      // the list of kept entries: written, read
    }
    return SRX_OK;
}

// host-side selection (srx_pca with an explicit feature list): bitmask + prefix counts from the remap table
int32_t build_tiled_fused(srx_mat* m, const std::vector<int32_t>& remap, int k, RowMajor& rm, Tiled* t256, RowXf xf, bool want_recs) {
    srx_ctx* ctx = m->ctx;
    const int n_words = (int)((remap.size() + 31) / 32);
    std::vector<uint32_t> hsel(2 * (size_t)n_words, 0u);
    for (size_t g = 0; g < remap.size(); ++g)
        if (remap[g] >= 0) hsel[g >> 5] |= 1u << (g & 31);
    uint32_t run = 0;
    for (int w = 0; w < n_words; ++w) {
        hsel[n_words + w] = run;
        run += (uint32_t)__builtin_popcount(hsel[w]);
    }
    uint32_t* d_sel;
    SRX_TRY(scratch(ctx, "pca_selbits", (hsel.size() ? hsel.size() : 1) * sizeof(uint32_t), (void**)&d_sel));
    SRX_TRY(h2d(ctx, d_sel, hsel.data(), hsel.size() * sizeof(uint32_t)));
    return build_tiled_fused(m, d_sel, n_words, k, rm, t256, xf, want_recs);
}

// The Gram kernel's second half runs on a stream whose CU mask leaves `kCommFreeCus` CUs alone when the rows are sharded: the
// collective's workgroups (RCCL: one per channel, persistent) then find a CU with room whatever the dispatcher does with the
// stripe kernel's 10 000 queued workgroups — measured in round 3: a second stream's first kernel sat 2.8 ms in its queue
// beside that grid, stream priority or not.  6 % of the CUs cost the half launch ~0.1 ms.
constexpr int kCommFreeCus = 16;
static int32_t ensure_comm_streams(srx_ctx* ctx) {
    if (!ctx->comm_stream) {
        SRX_HIP(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
        SRX_HIP(ctx, hipEventCreateWithFlags(&ctx->comm_fork, hipEventDisableTiming));
        SRX_HIP(ctx, hipEventCreateWithFlags(&ctx->comm_join, hipEventDisableTiming));
    }
    if (!ctx->gram_stream) {
        uint32_t mask[8];
        const int n_cus = ctx->n_cus > 256 ? 256 : ctx->n_cus;
        for (int w = 0; w < 8; ++w) mask[w] = 0u;
        for (int c = 0; c < n_cus; ++c)
            if (c >= kCommFreeCus) mask[c >> 5] |= 1u << (c & 31);
        if (n_cus <= 2 * kCommFreeCus ||
            hipExtStreamCreateWithCUMask(&ctx->gram_stream, (uint32_t)((n_cus + 31) / 32), mask) != hipSuccess) {
            (void)hipGetLastError();
            SRX_HIP(ctx, hipStreamCreateWithFlags(&ctx->gram_stream, hipStreamNonBlocking));
            ctx->gram_stream_masked = false;
        } else {
            ctx->gram_stream_masked = true;
        }
        SRX_HIP(ctx, hipEventCreateWithFlags(&ctx->gram_fork, hipEventDisableTiming));
        SRX_HIP(ctx, hipEventCreateWithFlags(&ctx->gram_join, hipEventDisableTiming));
    }
    return SRX_OK;
}

// G += A^T A of the row-major compacted matrix, into the packed upper triangle `Gp` (k (k + 1) / 2 doubles; the
// caller zeroes it for a fresh sum): owner buckets, then the stripe kernel.
// `reduce` (nullable): sum the triangle over the ranks HERE, the first half of the owners' rows on the communication stream
// while the second half is still being computed (*reduce is set when that was done; otherwise the caller's all-reduce follows).
// Whether the exchange is split is decided from rank-invariant data only (k, the communicator): a rank WITHOUT rows — more
// ranks than non-empty rows, a skewed cut, a filter that emptied a shard — skips the kernels and issues the same three
// collectives with the same counts as everybody else.
// `zero_first`: a fresh sum — the triangle is zeroed on the way (by the bucket pass; a rank without rows: a memset)
template <typename VT>
int32_t launch_gram(srx_ctx* ctx, const RowMajor& rm, double* Gp, bool* reduce, bool zero_first) {
    if (reduce) *reduce = false;
    GramPlan g;
    SRX_TRY(gram_plan(ctx, rm.k, rm.n_rows, g, rm.n_rows ? (double)rm.nnz / (double)rm.n_rows : 0.0, (int)sizeof(GramPk<VT>)));
    static const bool force_split = getenv("SRX_GRAM_OVERLAP") != nullptr;      // test switch: the split with a 1-rank communicator
    const int h = g.n_wg / 2;
    const bool split = reduce && comm_is_rccl(ctx) && (ctx->n_ranks > 1 || force_split) && h >= 1 && g.n_wg - h >= 1;
    const bool empty = rm.n_rows == 0;
    // (sharded rows: the accumulation mode is decided from the statistics of ALL ranks — one small all-reduce that a rank without
    //  rows makes too, below)
    const bool stat_exchange = reduce && ctx->n_ranks > 1;
    if (zero_first && empty) SRX_HIP(ctx, hipMemsetAsync(Gp, 0, (size_t)rm.k * (rm.k + 1) / 2 * sizeof(double), ctx->stream));
    if (empty && !split && !stat_exchange) return SRX_OK;
    uint32_t* boff = nullptr;
    int64_t *blk_total = nullptr, *rec_base = nullptr;
    GramRec<VT>* recs = nullptr;
    int64_t n_recs = 0;
    if (!empty) {
        SRX_TRY(scratch(ctx, "pca_boff", g.n_rblk * (size_t)(g.n_wg + 1) * sizeof(uint32_t), (void**)&boff));
        SRX_TRY(scratch(ctx, "pca_brtot", g.n_rblk * sizeof(int64_t), (void**)&blk_total));
        SRX_TRY(scratch(ctx, "pca_brbase", (g.n_rblk + 4) * sizeof(int64_t), (void**)&rec_base));
        // SRX_K_BUCKET: record counts, their read-back, the bucket pass — the compacted matrix read once (twice through L2),
        // the records written once
        ProfScope ps(ctx, SRX_K_BUCKET, (double)rm.nnz * sizeof(GramPk<VT>) + (double)(rm.n_rows + 1) * 8.0 * 2.0);
        // how many records each block makes (a suffix longer than a wave is several), and where its records start
        if (rm.n_recs >= 0) {
            n_recs = rm.n_recs;                  // counted with the compaction (build_tiled_fused): blk_total / rec_base are filled
        } else {
            hipLaunchKernelGGL(k_rec_count, dim3((unsigned)g.n_rblk), dim3(256), 0, ctx->stream, rm.ptr, rm.n_rows, g.rblk, blk_total);
            hipLaunchKernelGGL(k_rec_scan, dim3(1), dim3(1024), 0, ctx->stream, blk_total, g.n_rblk, rec_base);
            SRX_HIP(ctx, hipGetLastError());
            SRX_TRY(d2h(ctx, &n_recs, rec_base + g.n_rblk, sizeof(int64_t)));
        }
        SRX_TRY(scratch(ctx, "pca_brecs", ((size_t)n_recs + kGramUnroll) * sizeof(GramRec<VT>), (void**)&recs));
        if (ctx->prof_mask & (1u << SRX_K_BUCKET)) ctx->prof[SRX_K_BUCKET].bytes += (double)n_recs * sizeof(GramRec<VT>);
        hipLaunchKernelGGL((k_bucket<VT>), dim3((unsigned)g.n_rblk), dim3(kBucketThreads),
                           (size_t)(g.n_wg + 1 + g.rblk + 1 + kBucketGroup) * sizeof(uint32_t), ctx->stream, rm.ptr,
                           (const GramPk<VT>*)rm.pk, rm.n_rows, g.rblk, rm.k, g.sr_shift, g.n_wg, g.n_stripes, rec_base, boff, recs,
                           reinterpret_cast<uint32_t*>(rec_base + g.n_rblk + 2), n_recs, zero_first ? Gp : (double*)nullptr,
                           (uint64_t)rm.k * (rm.k + 1) / 2);
        SRX_HIP(ctx, hipGetLastError());
    }
    // Sharded rows: the fixed-point / f64 decision from the statistics of ALL ranks' compacted values (one more all-reduce of
    // 513 doubles — 4097 for f64 entries; an empty rank takes part with nothing marked).  Only where every rank is known to make the
    // same number of calls — the resident solve (`reduce` given); a backed session's tiles decide per tile (their number may
    // differ between the ranks; the sums of two modes differ by the fixed-point quantum).
    if (stat_exchange) {
        constexpr int EB = sizeof(VT) == 4 ? 8 : 11;             // exponent bits of the statistics words (f64 entries: high words)
        constexpr int n_bins = gstat_bins<EB>();
        double* bins = nullptr;
        SRX_TRY(scratch(ctx, "pca_gstat_bins", gstat_bins<11>() * sizeof(double), (void**)&bins));
        SRX_HIP(ctx, hipMemsetAsync(bins, 0, n_bins * sizeof(double), ctx->stream));
        uint32_t* gs = empty ? nullptr : reinterpret_cast<uint32_t*>(rec_base + g.n_rblk + 2);
        if (gs) hipLaunchKernelGGL(k_gstat_onehot<EB>, dim3(1), dim3(64), 0, ctx->stream, (const uint32_t*)gs, bins);
        SRX_TRY(allreduce_f64(ctx, bins, n_bins));
        if (gs) hipLaunchKernelGGL(k_gstat_decode<EB>, dim3(1), dim3(64), 0, ctx->stream, (const double*)bins, gs);
        SRX_HIP(ctx, hipGetLastError());
    }
    if (empty && !split) return SRX_OK;
    // SRX_GRAM_F64_ATOMICS=1: no fixed point whatever the statistics say (the kernel gets no statistics)
    const bool force_f64 = getenv("SRX_GRAM_F64_ATOMICS") && atoi(getenv("SRX_GRAM_F64_ATOMICS"));
    // srx_gram_mode_info: the three words the stripe kernel decides from, COPIED (they live in a scratch buffer the next, larger
    // compaction may free, and the next k_rec_scan zeroes them: ADVICE r5) into a buffer of the context's own, on the stream,
    // behind whatever made them final; a rank without rows, and a forced mode, say so instead of leaving the last launch's words
    ctx->gram_mode_state = empty ? 0 : (force_f64 ? 1 : 2);
    if (!empty && !force_f64) {
        if (!ctx->d_gram_mode) SRX_HIP(ctx, hipMalloc((void**)&ctx->d_gram_mode, 4 * sizeof(uint32_t)));
        SRX_HIP(ctx, hipMemcpyAsync(ctx->d_gram_mode, rec_base + g.n_rblk + 2, 3 * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
        ctx->gram_mode_f32 = sizeof(VT) == 4;
    }
    // SRX_K_GRAM: the stripe kernel alone.  Algorithmic bytes = what ANY Gram kernel must move: the compacted matrix and its
    // row pointers read once, the packed triangle written once.  The owner records and block offsets are this kernel's own
    // auxiliary input (aux bytes).  Every row suffix is read once per kept entry of its row (from L2 / Infinity Cache): that
    // shows up in the PMC traffic, not here.
    ProfScope ps(ctx, SRX_K_GRAM, (double)rm.nnz * sizeof(GramPk<VT>) + (double)(rm.n_rows + 1) * 8.0 + (double)rm.k * (rm.k + 1) / 2 * 8.0,
                 nullptr, (double)n_recs * sizeof(GramRec<VT>) + (double)g.n_rblk * (g.n_wg + 1) * 4.0);
    if (!empty) SRX_HIP(ctx, hipFuncSetAttribute((const void*)k_gram_stripes<VT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes));
    auto launch = [&](int w0, int n_w, hipStream_t st) {
        if (empty) return;
        hipLaunchKernelGGL((k_gram_stripes<VT>), dim3((unsigned)(n_w * g.n_z)), dim3(kGramWaves * kWave), g.lds_bytes, st,
                           rm.ptr, (const GramPk<VT>*)rm.pk, boff, rec_base, recs, g.n_rblk, g.rblk, rm.k, g.sr_shift, g.n_wg,
                           g.n_stripes, g.n_chunk, w0, n_w, Gp,
                           (const uint32_t*)(rec_base && !force_f64 ? reinterpret_cast<uint32_t*>(rec_base + g.n_rblk + 2) : nullptr));
    };
    // Sharded rows: owner w holds the stripes w and n_stripes - 1 - w, so the owners [0, h) hold the rows [0, h SR) and
    // [k - h SR, k) of the triangle — two contiguous ranges of the packed array — and the others the rows between.  Two
    // launches; the first one's ranges go round the ranks (RCCL, communication stream) under the second launch — which runs
    // on the CU-masked stream, so that the collective's workgroups have CUs of their own —, the middle range after it: half
    // of the 16 MB exchange is hidden.  (One launch on a single rank: the owners of a chunk share what they pull into L2,
    // and halving them costs more than nothing.)
    if (split) {
        SRX_TRY(ensure_comm_streams(ctx));
        const int k = rm.k;
        int r_lo, r_hi;
        gram_exchange_rows(g, r_lo, r_hi);                                       // rows [0, r_lo) + [r_hi, k): the first launch
        auto off = [&](int row) { return packed_row_offset(row, k); };
        launch(0, h, ctx->stream);
        SRX_HIP(ctx, hipGetLastError());
        SRX_HIP(ctx, hipEventRecord(ctx->comm_fork, ctx->stream));
        SRX_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->comm_fork, 0));
        SRX_TRY(allreduce_f64_on(ctx, Gp, off(r_lo), ctx->comm_stream));
        SRX_TRY(allreduce_f64_on(ctx, Gp + off(r_hi), off(k) - off(r_hi), ctx->comm_stream));
        // second half of the owners on the masked stream, joined back into the context's stream
        SRX_HIP(ctx, hipEventRecord(ctx->gram_fork, ctx->stream));
        SRX_HIP(ctx, hipStreamWaitEvent(ctx->gram_stream, ctx->gram_fork, 0));
        launch(h, g.n_wg - h, ctx->gram_stream);
        SRX_HIP(ctx, hipGetLastError());
        SRX_HIP(ctx, hipEventRecord(ctx->gram_join, ctx->gram_stream));
        SRX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->gram_join, 0));
        // the middle rows: on the communication stream too (one stream for all of the communicator's collectives in
        // flight), after the second launch
        SRX_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->gram_join, 0));
        SRX_TRY(allreduce_f64_on(ctx, Gp + off(r_lo), off(r_hi) - off(r_lo), ctx->comm_stream));
        SRX_HIP(ctx, hipEventRecord(ctx->comm_join, ctx->comm_stream));
        SRX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->comm_join, 0));
        *reduce = true;
        ctx->gram_splits++;
        return SRX_OK;
    }
    launch(0, g.n_wg, ctx->stream);
    SRX_HIP(ctx, hipGetLastError());
    return SRX_OK;
}

template int32_t launch_gram<float>(srx_ctx*, const RowMajor&, double*, bool*, bool);
template int32_t launch_gram<double>(srx_ctx*, const RowMajor&, double*, bool*, bool);

}  // namespace srx

extern "C" int32_t srx_gram_mode_info(srx_ctx* ctx, int32_t* mode_out) {
    using namespace srx;
    if (!ctx || !mode_out) return fail(ctx, SRX_E_ARG, "srx_gram_mode_info: null argument");
    *mode_out = 0;
    if (ctx->gram_mode_state == 0) return SRX_OK;                  // no launch yet, or the last one had no rows on this rank
    if (ctx->gram_mode_state == 1) {                               // SRX_GRAM_F64_ATOMICS
        *mode_out = 1;
        return SRX_OK;
    }
    uint32_t w[3] = {0, 0, 0};
    SRX_TRY(d2h(ctx, w, ctx->d_gram_mode, sizeof w));
    int kq;
    *mode_out = (ctx->gram_mode_f32 ? gram_fixed_point_mode(w[0], w[1], w[2], kq) : gram_fixed_point_mode64(w[0], w[1], w[2], kq)) ? 2 : 1;
    return SRX_OK;
}

extern "C" int32_t srx_gram_exchange_ranges(uint64_t k, uint64_t* offsets_out) {
    using namespace srx;
    if (!offsets_out || k < 1 || k > 16384) return fail(nullptr, SRX_E_ARG, "srx_gram_exchange_ranges: bad arguments");
    GramPlan g;
    gram_stripes_of((int)k, g);
    int r_lo, r_hi;
    gram_exchange_rows(g, r_lo, r_hi);
    offsets_out[0] = 0;
    offsets_out[1] = packed_row_offset(r_lo, (int)k);
    offsets_out[2] = packed_row_offset(r_hi, (int)k);
    offsets_out[3] = packed_row_offset((int)k, (int)k);
    return SRX_OK;
}
