// spmm.inl — included by pca_solve.hip inside namespace srx (one translation unit: the kernels share its helpers and constants).
// The sparse products with a k x 64 panel: forward (tile-major and row-major record forms) and transposed.

// ---- forward SpMM: Y = A P - 1 cvec^T ----------------------------------------------------------
// Workgroup = 512 threads = 32 groups of 16 lanes; a group owns 16 consecutive cells and lane
// q of it owns panel columns 4q..4q+3 of all 16 output rows (64 accumulator registers).  The
// workgroup walks the gene tiles; per tile it stages the 256 x 64 panel tile in LDS (64 KiB
// as f32), then every group reads the <= 16 (index, value) pairs of each of its rows' tile
// segment with one coalesced load and every lane visits them in DPP-rotated order (lane q
// takes pair (q+s)%16 at step s): one conflict-free ds_read_b128 of the panel row + 4 FMAs
// per pair, no broadcast and no atomics.  Empty slots carry value 0 and index 0.
constexpr int kFwdThreads = 512;
template <typename PT> struct FwdCfg;
template <> struct FwdCfg<float> { static constexpr int kRows = 8, kWavesPerSimd = 4, kStage = 8; };   // 2 workgroups / CU
template <> struct FwdCfg<double> { static constexpr int kRows = 8, kWavesPerSimd = 2, kStage = 4; };   // 1 workgroup / CU

template <typename PT, int S>
struct FwdRot {
    // Step S uses the pair currently in (ci, cv), then rotates both by ONE lane for the next step.
    // (Rotating the original pair by S at every step gives the scheduler 30 independent DPP moves
    // per row to hoist — it did, and spilled; the chain keeps one live copy.)
    static __device__ __forceinline__ void run(int ci, PT cv, const PT* __restrict__ panel_q, PT (&a)[4]) {
        Vec4<PT> p;
        p.load(panel_q + ci);
        a[0] += cv * p[0];
        a[1] += cv * p[1];
        a[2] += cv * p[2];
        a[3] += cv * p[3];
        // at most 4 panel reads (16 VGPRs) in flight: without the fence the scheduler hoists the
        // ds_read_b128 of all 16 steps (64 VGPRs per row) and spills under the 128-VGPR budget
        if constexpr ((S & 3) == 3) asm volatile("" ::: "memory");
        if constexpr (S + 1 < 16) FwdRot<PT, S + 1>::run(ror16<1>(ci), ror16<1>(cv), panel_q, a);
    }
};

// One batch of kStage rows of a group: issue their (index, value) chunk loads together, then
// run the 16 rotation steps of each.  H is a compile-time row offset so that the accumulator
// array is only ever indexed statically (it must stay in registers).  Rows of a group are
// consecutive, so row r's segment ends where row r+1's starts: `la` (lane q: start of row q,
// relative to the group's first entry) and `le` (end of the last row) describe all of them.
// Index loads are unconditional (the arrays are padded by 64 entries; a stray index is a valid
// local column) and only the VALUE is masked to 0 — no divergent branches around the loads.
template <typename VT, typename PT, int kRows, int kStage, int H>
__device__ __forceinline__ void fwd_stage(const GramPk<VT>* __restrict__ gpk, int la, int le, int c, int q,
                                          const PT* __restrict__ panel_q, PT (&acc)[kRows][4]) {
    if constexpr (H < kRows) {
        int ci[kStage];
        PT cvv[kStage];
#pragma unroll
        for (int r = 0; r < kStage; ++r) {
            const int lo = __shfl(la, H + r, 16) + c;
            const int hi = (H + r + 1 < 16) ? __shfl(la, (H + r + 1) & 15, 16) : le;
            const int p = lo + q;
            const GramPk<VT> e = gpk[p];
            ci[r] = e.j * L;
            cvv[r] = p < hi ? (PT)e.v : PT(0);
        }
#pragma unroll
        for (int r = 0; r < kStage; ++r) FwdRot<PT, 0>::run(ci[r], cvv[r], panel_q, acc[H + r]);
        fwd_stage<VT, PT, kRows, kStage, H + kStage>(gpk, la, le, c, q, panel_q, acc);
    }
}

// Entries 16.. of row H (and, recursively, of the rows after it) for the groups that have them.
template <typename VT, typename PT, int kRows, int H>
__device__ __forceinline__ void fwd_overflow(const GramPk<VT>* __restrict__ gpk, int la, int le, int q,
                                             const PT* __restrict__ panel_q, PT (&acc)[kRows][4]) {
    if constexpr (H < kRows) {
        const int lo = __shfl(la, H, 16);
        const int hi = (H + 1 < 16) ? __shfl(la, (H + 1) & 15, 16) : le;
        for (int c = 16; __any(hi - lo > c); c += 16) {
            const int p = lo + c + q;
            const GramPk<VT> e = gpk[p];
            FwdRot<PT, 0>::run(e.j * L, p < hi ? (PT)e.v : PT(0), panel_q, acc[H]);
        }
        fwd_overflow<VT, PT, kRows, H + 1>(gpk, la, le, q, panel_q, acc);
    }
}

template <typename VT, typename PT>
__global__ __launch_bounds__(kFwdThreads, FwdCfg<PT>::kWavesPerSimd) void k_spmm_fwd(
    const int64_t* __restrict__ tptr, const GramPk<VT>* __restrict__ tpk, uint64_t n_rows,
    int nt, int k, const PT* __restrict__ P, const PT* __restrict__ cvec, PT* __restrict__ Y,
    double* __restrict__ scores /* nullable: n_rows x ld row-major f64 (first n_pc panel columns), written INSTEAD of Y */,
    int n_pc, int ld) {
    constexpr int kRows = FwdCfg<PT>::kRows;            // rows per 16-lane group
    constexpr int kRowsPerWg = (kFwdThreads / 16) * kRows;
    constexpr int kStage = FwdCfg<PT>::kStage;          // rows whose (index, value) chunks are in flight together
    extern __shared__ double lds_raw[];
    PT* panel = reinterpret_cast<PT*>(lds_raw);
    const int q = threadIdx.x & 15;
    const int group = threadIdx.x >> 4;
    const PT* panel_q = panel + 4 * q;
    Vec4<PT> cv4;
    cv4.load(cvec + 4 * q);
    const uint64_t n_blocks = (n_rows + kRowsPerWg - 1) / kRowsPerWg;
    for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const uint64_t i0 = blk * kRowsPerWg + (uint64_t)group * kRows;
        PT acc[kRows][4];
#pragma unroll
        for (int r = 0; r < kRows; ++r) acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = PT(0);
        for (int t = 0; t < nt; ++t) {
            __syncthreads();                         // everyone is done with the previous tile
            for (int e = threadIdx.x * 4; e < KT * L; e += kFwdThreads * 4) {
                Vec4<PT> v;
                if (t * KT + e / L < k) v.load(P + (size_t)t * KT * L + e);
                else v[0] = v[1] = v[2] = v[3] = PT(0);
                v.store(panel + e);
            }
            __syncthreads();
            // lane q: start of row i0+q in this tile (rows past the end collapse to empty segments)
            const int64_t* tp = tptr + (uint64_t)t * n_rows;
            const uint64_t rq = i0 + q < n_rows ? i0 + q : n_rows;
            const uint64_t rend = i0 + kRows < n_rows ? i0 + kRows : n_rows;
            const int64_t pa = tp[rq];
            const int64_t p0 = __shfl(pa, 0, 16);
            const int la = (int)(pa - p0);
            const int le = (int)(tp[rend] - p0);
            // NB: the shuffle must run with every lane active (a lane-dependent ?: would mask lane 15
            // out of the ds_bpermute and lane 14 would read 0 from it)
            const int la_next = __shfl(la, (q + 1) & 15, 16);
            const int nxt = (q + 1 < 16) ? la_next : le;
            const int len = q < kRows ? (q + 1 < kRows ? nxt : le) - la : 0;
            const GramPk<VT>* gpk = tpk + p0;
            fwd_stage<VT, PT, kRows, kStage, 0>(gpk, la, le, 0, q, panel_q, acc);
            // segments longer than 16 entries are rare (~1 % of rows at m/k*256 = 9): finish them row by
            // row instead of sending the whole wave through another 16-row pass
            if (__any(len > 16)) fwd_overflow<VT, PT, kRows, 0>(gpk, la, le, q, panel_q, acc);
        }
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const uint64_t row = i0 + r;
            if (row < n_rows) {
                Vec4<PT> o;
                o[0] = acc[r][0] - cv4[0];
                o[1] = acc[r][1] - cv4[1];
                o[2] = acc[r][2] - cv4[2];
                o[3] = acc[r][3] - cv4[3];
                if (scores) {            // the transform pass: obsm["X_pca"] layout directly (dim_red/mod.rs:105-106)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * q + j < n_pc) scores[row * (uint64_t)ld + 4 * q + j] = (double)o[j];
                } else {
                    o.store(Y + row * L + 4 * q);
                }
            }
        }
    }
}

// ---- forward SpMM from the ROW-MAJOR records: Y[:, slice] = A P[:, slice] - 1 c^T (round 2) -------------------------
// The tile-major kernel above cuts the gene axis into 256-column tiles because a 64-column panel does not fit the LDS
// (2000 x 64 x 4 B = 512 KB); a row then falls into ~8 segments of ~9 entries, visited in 16-slot chunks: 56 % of the
// rotation steps carry a zero.  Here the PANEL COLUMNS are cut instead: a workgroup holds C = 4 Q columns of ALL k genes
// (k x C x sizeof(PT): 128 KB at k = 2000 with C = 16 floats / 8 doubles) and walks whole rows; the n_pc / C column
// slices of the same rows run on neighbouring workgroups (blockIdx = row group x n_slices + slice), so the matrix comes
// out of L2 for all but the first of them.  A row is taken by Q adjacent lanes (lane q owns columns 4q .. 4q+3 of the
// slice): they load Q consecutive records per step and broadcast them to each other in order (DPP quad_perm), one
// ds_read_b128 of the panel row + 4 FMAs per record and lane — chunks of Q instead of 16: no padding worth mentioning.
constexpr int kFwdRowsThreads = 1024;

template <int Q>
__device__ __forceinline__ int quad_bcast(int x, int s) {
    if constexpr (Q == 1) {
        return x;                                               // one lane per row: nothing to broadcast
    } else if constexpr (Q == 4) {
        switch (s) {                                            // v_mov_b32_dpp quad_perm:[s,s,s,s]
            case 0: return __builtin_amdgcn_update_dpp(0, x, 0x00, 0xf, 0xf, true);   // (bound_ctrl: no "old" value to set up)
            case 1: return __builtin_amdgcn_update_dpp(0, x, 0x55, 0xf, 0xf, true);
            case 2: return __builtin_amdgcn_update_dpp(0, x, 0xaa, 0xf, 0xf, true);
            default: return __builtin_amdgcn_update_dpp(0, x, 0xff, 0xf, 0xf, true);
        }
    } else {                                                    // pairs: lanes (2i, 2i+1) of every quad
        return s == 0 ? __builtin_amdgcn_update_dpp(0, x, 0xa0, 0xf, 0xf, true)       // [0,0,2,2]
                      : __builtin_amdgcn_update_dpp(0, x, 0xf5, 0xf, 0xf, true);      // [1,1,3,3]
    }
}
template <int Q>
__device__ __forceinline__ float quad_bcast_v(float x, int s) {
    return __builtin_bit_cast(float, quad_bcast<Q>(__builtin_bit_cast(int, x), s));
}
template <int Q>
__device__ __forceinline__ double quad_bcast_v(double x, int s) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = quad_bcast<Q>((int)(b & 0xffffffffll), s), hi = quad_bcast<Q>((int)(b >> 32), s);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Rows ordered by their number of kept entries (counting sort on the length, `perm`): the Q-lane groups of a wave then
// hold rows of (nearly) equal length and the record loop is wave-uniform — with rows in natural order a wave runs as long as
// the longest of its 16 rows (+30 %), and walking several rows per group as one stream instead puts a row-boundary test
// into every step of 16 independent streams (that version: 1.44 ms against the tile-major kernel's 1.22).
constexpr int kLenBins = 512;
constexpr int kLenRowsPerWg = 4096;
__global__ __launch_bounds__(256) void k_len_hist(const int64_t* __restrict__ rm_ptr, uint64_t n_rows,
                                                  uint32_t* __restrict__ hist /* kLenBins, zeroed */) {
    __shared__ uint32_t s[kLenBins];
    for (int e = threadIdx.x; e < kLenBins; e += blockDim.x) s[e] = 0u;
    __syncthreads();
    const uint64_t r0 = (uint64_t)blockIdx.x * kLenRowsPerWg, r1 = r0 + kLenRowsPerWg < n_rows ? r0 + kLenRowsPerWg : n_rows;
    for (uint64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const int64_t n = rm_ptr[r + 1] - rm_ptr[r];
        atomicAdd(&s[n < kLenBins - 1 ? (int)n : kLenBins - 1], 1u);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kLenBins; e += blockDim.x)
        if (s[e]) atomicAdd(&hist[e], s[e]);
}
__global__ void k_len_scan(uint32_t* __restrict__ hist /* kLenBins counts -> exclusive offsets */) {
    __shared__ uint32_t s[kLenBins];
    const int t = threadIdx.x;
    s[t] = hist[t];
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (int i = 0; i < kLenBins; ++i) { const uint32_t c = s[i]; s[i] = run; run += c; }
    }
    __syncthreads();
    hist[t] = s[t];
}
// a workgroup reserves, per length, a run for its rows with ONE global atomic and fills it through LDS cursors
__global__ __launch_bounds__(256) void k_len_scatter(const int64_t* __restrict__ rm_ptr, uint64_t n_rows,
                                                     uint32_t* __restrict__ cursor, uint32_t* __restrict__ perm) {
    __shared__ uint32_t s[kLenBins];
    for (int e = threadIdx.x; e < kLenBins; e += blockDim.x) s[e] = 0u;
    __syncthreads();
    const uint64_t r0 = (uint64_t)blockIdx.x * kLenRowsPerWg, r1 = r0 + kLenRowsPerWg < n_rows ? r0 + kLenRowsPerWg : n_rows;
    for (uint64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const int64_t n = rm_ptr[r + 1] - rm_ptr[r];
        atomicAdd(&s[n < kLenBins - 1 ? (int)n : kLenBins - 1], 1u);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kLenBins; e += blockDim.x) {
        const uint32_t c = s[e];
        s[e] = c ? atomicAdd(&cursor[e], c) : 0u;
    }
    __syncthreads();
    for (uint64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const int64_t n = rm_ptr[r + 1] - rm_ptr[r];
        perm[atomicAdd(&s[n < kLenBins - 1 ? (int)n : kLenBins - 1], 1u)] = (uint32_t)r;
    }
}

// RANGE: the panel slice of ALL k genes does not fit the LDS (f64 panels beyond 5118 genes): the launch covers the genes
// [k_lo, k_hi) only — entries outside contribute nothing — and, from the second range on (`accumulate`), adds to what the
// earlier ranges left in the output.
// CL: panel columns per lane — 4, or 5 (the lane's four + one of the slice's last Q columns): a slice of 5 Q columns of all k
// genes is the widest the LDS takes at k = 2000 (160 000 B), and n_pc = 50 then needs 3 (f32; 5 with f64 panels) passes over
// the matrix instead of 4 (7) — the kernel is bound by its reads (see the note in the body), not by what it does with them.
template <typename VT, typename PT, int Q, bool RANGE = false, int CL = 4>
__global__ __launch_bounds__(kFwdRowsThreads) void k_spmm_rows(
    const int64_t* __restrict__ rm_ptr, const GramPk<VT>* rm /* NOT __restrict__: see the barrier behind the chunk loads */,
    const uint32_t* __restrict__ perm /* nullable */,
    uint64_t n_rows, int k, const PT* __restrict__ P /* k x 64 */, const PT* __restrict__ cvec /* 64 */,
    int n_cols /* panel columns wanted */, double* __restrict__ scores /* n_rows x ld f64 (nullable) */,
    PT* __restrict__ Y /* n_rows x 64 (nullable) */, int ld, int ldp /* elements between two genes of the slice in LDS */,
    int k_lo = 0, int k_hi = 0, int accumulate = 0, int col_base = 0 /* first panel column of this launch */) {
    static_assert(CL == 4 || CL == 5, "columns per lane");
    constexpr int C = CL * Q;
    if constexpr (!RANGE) {
        k_lo = 0;
        k_hi = k;
    }
    extern __shared__ double lds_raw[];
    PT* panel = reinterpret_cast<PT*>(lds_raw);                 // k x C
    const int n_slices = (n_cols - col_base + C - 1) / C;
    // the column slices of one row range sit on the SAME XCD (consecutive workgroup ids go round the 8 XCDs): they walk the
    // same rows at the same pace, so the entries come out of that XCD's L2 for all but the first of them
    const uint64_t n_wg = gridDim.x / n_slices;
    int slice;
    uint64_t wg;
    if (n_wg % 8 == 0) {
        const uint64_t t = blockIdx.x / 8;
        slice = (int)(t % n_slices);
        wg = (t / n_slices) * 8 + blockIdx.x % 8;
    } else {
        slice = blockIdx.x % n_slices;
        wg = blockIdx.x / n_slices;
    }
    // a gene of the slice in LDS: its C columns in panel order
    if constexpr (C % 4 == 0) {
        for (int e = threadIdx.x; e < (k_hi - k_lo) * (C / 4); e += kFwdRowsThreads) {
            const int j = e / (C / 4), cq = e % (C / 4);
            Vec4<PT> v;
            v.load(P + (size_t)(k_lo + j) * L + col_base + slice * C + cq * 4);
            v.store(panel + (size_t)j * ldp + cq * 4);
        }
    } else {
        for (int e = threadIdx.x; e < (k_hi - k_lo) * C; e += kFwdRowsThreads) {
            const int j = e / C, cq = e % C;
            panel[(size_t)j * ldp + cq] = P[(size_t)(k_lo + j) * L + col_base + slice * C + cq];
        }
    }
    __syncthreads();
    const int ql = threadIdx.x % Q;                             // lane within the row's lane group
    constexpr uint64_t kGroups = kFwdRowsThreads / Q;
    const int col0 = col_base + slice * C + ql * 4;             // the lane's four columns ...
    const int colx = col_base + slice * C + 4 * Q + ql;         // ... and, CL = 5, its one of the slice's last Q
    Vec4<PT> cv4;
    cv4.load(cvec + col0);
    PT cvx = PT(0);
    if constexpr (CL == 5) cvx = cvec[colx];
    constexpr int kGeneBytes = C * (int)sizeof(PT);             // one gene of the slice in LDS
    const char* panel_q = reinterpret_cast<const char*>(panel) + ql * 4 * (int)sizeof(PT);
    const char* panel_x = reinterpret_cast<const char*>(panel) + (4 * Q + ql) * (int)sizeof(PT);
    // sorted position i -> row perm[i]; a wave's groups take consecutive positions (equal lengths), the workgroups
    // interleave so that the long rows at the end are spread over all of them.
    // A chunk = 4 Q consecutive records of a row (lane ql holds records 4 ql .. 4 ql + 3: one 32- / 64-byte load per lane); a
    // BATCH = kSub chunks.  The lane group's rows are one stream of batches worked through with two register sets: the batch
    // after this one — the row's next, or the first of the next row, whose pointers came a row ahead — is in flight while this
    // one is multiplied, and a row's output stores are issued behind the loads of the next row's first batch (vmcnt counts in
    // order: a load waited for behind a store waits for the store's acknowledgement too).
    // What round 2's loop did instead, timed inside the kernel (-DSPMM_TIMING, ns per 16-row step of a wave, 7.1-10.7 us in
    // all): it had "load the next chunk, work on this one, cur = nxt", which the compiler turned into "load this chunk,
    // wait, work on it" (the next load equals the following iteration's) — an exposed ~0.85 us round trip per chunk; its 16
    // panel reads per chunk were re-interleaved with the multiply-adds two reads deep by the scheduler — 16 LDS round
    // trips per chunk, 1.7 us per row; the copy of the next row's pointers at the top of the body waited for the loads
    // just issued; four 8-byte stores per lane behind exec branches.
    constexpr int kSub = sizeof(VT) == 4 && CL == 4 ? 2 : 1;
    constexpr int kBatch = kSub * 4 * Q;                        // records of a batch
    const uint64_t stride = n_wg * kGroups, i_first = wg * kGroups + threadIdx.x / Q, last_row = n_rows ? n_rows - 1 : 0;
    if (i_first >= n_rows) return;
    auto row_at = [&](uint64_t i) -> uint64_t {
        const uint64_t c = i < n_rows ? i : last_row;
        return perm ? (uint64_t)perm[c] : c;
    };
    struct Batch { GramPk<VT> r[kSub][4]; };
    auto load_batch = [&](Batch& b, const GramPk<VT>* base /* the row's records */, int st, int n) {
#pragma unroll
        for (int c = 0; c < kSub; ++c) {
            // (a chunk past the row's end is not fetched: its lanes read the row's first chunk again)
            const GramPk<VT>* at = st + c * 4 * Q < n ? base + st + c * 4 * Q : base;
#pragma unroll
            for (int u = 0; u < 4; ++u) b.r[c][u] = at[ql * 4 + u];    // (the array is padded by a wave of records)
        }
    };
    uint64_t i = i_first;
    // position -> row -> row pointers -> records is three dependent loads: the pointers of the next row are resident, those of
    // the row after it and the row of the position behind that one were asked for a row ago (a step selects between "this
    // row's next batch" and "the next row's first" — it needs the next row's pointers at once)
    uint64_t row_c = row_at(i), row_n = row_at(i + stride), row_nn = row_at(i + 2 * stride), row_n3 = row_at(i + 3 * stride);
    int64_t lo = rm_ptr[row_c];
    int n = (int)(rm_ptr[row_c + 1] - lo);
    int64_t lo_n = rm_ptr[row_n], hi_n = rm_ptr[row_n + 1];
    int64_t lo_nn = rm_ptr[row_nn], hi_nn = rm_ptr[row_nn + 1];
    const GramPk<VT>* rr = rm + lo;
    int st = 0;
    PT a0 = PT(0), a1 = PT(0), a2 = PT(0), a3 = PT(0), ax = PT(0);
    bool done = false;
    struct Out { uint64_t row; PT o[4]; PT x; };
    auto store_row = [&](const Out& q) {
        const PT o0 = q.o[0], o1 = q.o[1], o2 = q.o[2], o3 = q.o[3], ox = q.x;
        const uint64_t row_o = q.row;
        if (scores) {
            if constexpr (CL == 5) {
                if (colx < n_cols) {
                    double* dx = scores + row_o * (uint64_t)ld + colx;
                    *dx = RANGE && accumulate ? *dx + (double)ox : (double)ox;
                }
            }
            double* dst = scores + row_o * (uint64_t)ld + col0;
            if (!(RANGE && accumulate) && col0 + 3 < n_cols && ld % 2 == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0) {
                // four doubles as two 16-byte stores.  (Non-temporal stores: 0.79 against 0.68 ms — the pieces of a row written
                // by the slices' workgroups meet in L2.  Lanes owning the column pairs {2q, 2q + 1} and {2Q + 2q, ..} so that a
                // store instruction covers 64 contiguous bytes per row, and record loads laid out the same way: no change.)
                *reinterpret_cast<double2*>(dst) = double2{(double)o0, (double)o1};
                *reinterpret_cast<double2*>(dst + 2) = double2{(double)o2, (double)o3};
            } else if (RANGE && accumulate) {
                if (col0 + 0 < n_cols) dst[0] += (double)o0;
                if (col0 + 1 < n_cols) dst[1] += (double)o1;
                if (col0 + 2 < n_cols) dst[2] += (double)o2;
                if (col0 + 3 < n_cols) dst[3] += (double)o3;
            } else {
                if (col0 + 0 < n_cols) dst[0] = (double)o0;
                if (col0 + 1 < n_cols) dst[1] = (double)o1;
                if (col0 + 2 < n_cols) dst[2] = (double)o2;
                if (col0 + 3 < n_cols) dst[3] = (double)o3;
            }
        } else {
            Vec4<PT> o;
            if (RANGE && accumulate) {
                o.load(Y + row_o * L + col0);
                o[0] += o0; o[1] += o1; o[2] += o2; o[3] += o3;
            } else {
                o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
            }
            o.store(Y + row_o * L + col0);
            if constexpr (CL == 5) {
                PT* yx = Y + row_o * L + colx;
                *yx = RANGE && accumulate ? *yx + ox : ox;
            }
        }
    };
    // one batch: fetch the following one into `nxt`, multiply `cur`, finish the row if this was its last batch
    auto step = [&](const Batch& cur, Batch& nxt) {
        const bool last = st + kBatch >= n;
        load_batch(nxt, last ? rm + lo_n : rr, last ? 0 : st + kBatch, last ? (int)(hi_n - lo_n) : n);
        // the loads stay HERE (`rm` is not `__restrict__`: a load nothing can alias may be moved across this barrier, and the
        // compiler then sinks it to its first use — behind the multiplication)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < kSub; ++c) {
            const int s0 = st + c * 4 * Q;
            if (s0 >= n && c > 0) break;
            // the lane's own four records: byte offset of the gene in the slice, value zeroed past the row's end (the
            // column is then a valid one of a later row, or of the zeroed tail) or outside the launch's gene range
            int off[4];
            PT val[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int j = cur.r[c][u].j;
                bool ok = s0 + ql * 4 + u < n;
                if constexpr (RANGE) {
                    const bool in = j >= k_lo && j < k_hi;
                    ok = ok && in;
                    j = in ? j - k_lo : 0;
                }
                off[u] = j * kGeneBytes;
                val[u] = ok ? (PT)cur.r[c][u].v : PT(0);
            }
            // kDeep panel reads first (their addresses only need the broadcast offsets), then their multiply-adds; the barrier
            // keeps the scheduler from re-interleaving them two deep to save registers
            constexpr int kDeep = sizeof(PT) == 4 ? (CL == 4 ? 4 * Q : 2 * Q) : (Q >= 2 ? 2 * Q : 4 * Q);
#pragma unroll
            for (int h = 0; h < 4 * Q; h += kDeep) {
                Vec4<PT> pv[kDeep];
                PT px[kDeep];
#pragma unroll
                for (int s_ = 0; s_ < kDeep; ++s_) {
                    const int o = quad_bcast<Q>(off[(h + s_) & 3], (h + s_) >> 2);
                    pv[s_].load(reinterpret_cast<const PT*>(panel_q + o));
                    if constexpr (CL == 5) px[s_] = *reinterpret_cast<const PT*>(panel_x + o);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s_ = 0; s_ < kDeep; ++s_) {
                    const PT v = quad_bcast_v<Q>(val[(h + s_) & 3], (h + s_) >> 2);
                    a0 += v * pv[s_][0];
                    a1 += v * pv[s_][1];
                    a2 += v * pv[s_][2];
                    a3 += v * pv[s_][3];
                    if constexpr (CL == 5) ax += v * px[s_];
                }
            }
        }
        if (!last) {
            st += kBatch;
            return;
        }
        // (a register queue that sent the results of 2 / 4 rows out together changed nothing: it is not the stores' latency)
        i += stride;
        done = i >= n_rows;
        {
            Out q;
            q.row = row_c;
            q.o[0] = a0 - cv4[0]; q.o[1] = a1 - cv4[1]; q.o[2] = a2 - cv4[2]; q.o[3] = a3 - cv4[3];
            q.x = ax - cvx;
            if (RANGE && accumulate) { q.o[0] = a0; q.o[1] = a1; q.o[2] = a2; q.o[3] = a3; q.x = ax; }   // (the centring term went in with the first range)
            store_row(q);
        }
        // next row
        row_c = row_n;
        lo = lo_n;
        n = (int)(hi_n - lo_n);
        rr = rm + lo;
        st = 0;
        a0 = a1 = a2 = a3 = ax = PT(0);
        row_n = row_nn;
        lo_n = lo_nn;
        hi_n = hi_nn;
        row_nn = row_n3;
        lo_nn = rm_ptr[row_nn];
        hi_nn = rm_ptr[row_nn + 1];
        row_n3 = row_at(i + 3 * stride);
    };
    Batch A, B;
    load_batch(A, rr, 0, n);
    for (;;) {
        step(A, B);
        if (done) break;
        step(B, A);
        if (done) break;
    }
}

// ---- forward SpMM over gene RANGES of the whole 64-column panel, one QUAD per row (round 6) -------------------------
// k_spmm_rows above cuts the panel's COLUMNS (16 per workgroup): a gene of the slice is 64 bytes, so the four rows a
// ds_read_b128 serves per LDS cycle (its fixed lane groups of 16) hit four 16-bank windows chosen by gene mod 4 — 2.1 cycles per
// read on random genes (half of the LDS pipe's time is bank conflicts, profiles/r05_pmc_spmm.md) — and the matrix is walked once
// per slice (4 walks for n_pc = 50).  Here the panel's GENES are cut instead: the LDS holds all 64 columns of G = 512 genes
// (256 bytes per gene, 128 KB) and the workgroup walks the ranges in phases over a block of 256 S rows whose accumulators stay
// in registers.  Entries of a row are column-sorted, so phase r consumes the run [cursor, first j >= hi) and leaves the cursor
// for phase r + 1: the matrix is read ONCE.
// A row is taken by the four lanes of a quad: lane w owns the 16 columns 16 w .. 16 w + 15 as four 16-byte pieces, and reads
// piece (i + quad) mod 4 with its i-th ds_read_b128 — the hardware serves a b128 read in fixed groups of 16 lanes
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, .. whose four quads have four different (quad mod 4): their 16 reads cover the 64
// banks once whatever the genes.  CONFLICT-FREE (SQ_LDS_BANK_CONFLICT = 0, profiles/r06_pmc_spmm.md).
// A wave holds 16 rows at a time, in lockstep: rows come in length order (`perm`), 16 neighbours per wave slot, slots dealt
// round-robin over all waves of the grid.  A lane loads 8 consecutive records of its row's run (a quad: a chunk of 32) and
// entry t of the chunk reaches the quad's lanes by a quad_perm DPP broadcast from lane t / 8.
// (First form of the round, one 16-LANE group per row with the chunk replicated in its four quads: conflict-free too, but the
//  per-record preparation and the record loads were done four times over — 2.5e8 VALU instructions against the column-slice
//  kernel's 1.7e8 — 0.82 ms against 0.62: profiles/r06_knockouts.md.)
template <typename PT> struct RgCfg;
template <> struct RgCfg<float> { static constexpr int kGenes = 512; };
template <> struct RgCfg<double> { static constexpr int kGenes = 256; };

template <typename VT, typename PT, int S /* wave slots (rows per quad) per block, <= 4 */, int kRgThreads /* 512: 256 registers a wave */,
          int kDeep /* entries whose panel reads are in flight together */>
__global__ __launch_bounds__(kRgThreads) void k_spmm_ranges(
    const int64_t* __restrict__ rm_ptr, const GramPk<VT>* rm /* NOT __restrict__: the prefetches must stay where they are issued */,
    const uint32_t* __restrict__ perm /* nullable */, uint64_t n_rows, int k, const PT* __restrict__ P /* k x 64 */,
    const PT* __restrict__ cvec /* 64 */, int n_cols, double* __restrict__ scores /* n_rows x ld f64 (nullable) */,
    PT* __restrict__ Y /* n_rows x 64 (nullable) */, int ld) {
    static_assert(S >= 1 && S <= 4, "a lane of the quad keeps one slot's cursor");
    static_assert(kDeep >= 1 && kDeep <= 8 && 8 % kDeep == 0, "groups of entries inside a lane's 8 records");
    constexpr int G = RgCfg<PT>::kGenes;
    constexpr int RPL = sizeof(VT) == 4 ? 8 : 4;                // records per lane and chunk
    constexpr int kChunk = 4 * RPL;
    constexpr int kRowBytes = L * (int)sizeof(PT);
    constexpr int kWaves = kRgThreads / 64;
    extern __shared__ double lds_raw[];
    char* panel = reinterpret_cast<char*>(lds_raw);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar: slot arithmetic in SGPRs)
    const int quad = lane >> 2, w4 = lane & 3;
    const uint64_t n_waves = (uint64_t)gridDim.x * kWaves;
    const uint64_t W = (uint64_t)blockIdx.x * kWaves + wave;
    const uint64_t n_slots = (n_rows + 15) / 16;
    const uint64_t per_wave = (n_slots + n_waves - 1) / n_waves;
    const int n_blocks = (int)((per_wave + S - 1) / S);         // the same for every wave: the phases are barriers
    const int n_ranges = (k + G - 1) / G;
    // piece i of this lane: columns 16 w4 + 4 ((i + quad) & 3) .. + 3 (pcb: its byte offset in a gene's panel row)
    int pc[4], pcb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pc[i] = 16 * w4 + 4 * ((i + quad) & 3);
        pcb[i] = pc[i] * (int)sizeof(PT);
    }
    struct Chunk { GramPk<VT> r[RPL]; };
    auto load_chunk = [&](Chunk& c, uint32_t at) {
        const GramPk<VT>* p = rm + at + RPL * w4;               // (the array is padded by a wave of records)
#pragma unroll
        for (int u = 0; u < RPL; ++u) c.r[u] = p[u];
    };
    auto row_of = [&](int b, int s_) -> uint64_t {              // the row of this lane's quad in slot s_ of block b, or n_rows
        const uint64_t pos = 16 * (((uint64_t)b * S + s_) * n_waves + W) + quad;
        if (pos >= n_rows) return n_rows;
        return perm ? (uint64_t)perm[pos] : pos;
    };
    for (int b = 0; b < n_blocks; ++b) {
        // lane w of a quad keeps the cursor and the end of the quad's row in slot w
        uint32_t curv = 0u, endv = 0u;
        if (w4 < S) {
            const uint64_t row = row_of(b, w4);
            if (row < n_rows) {
                curv = (uint32_t)rm_ptr[row];
                endv = (uint32_t)rm_ptr[row + 1];
            }
        }
        PT acc[S][4][4];
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[s_][i][0] = acc[s_][i][1] = acc[s_][i][2] = acc[s_][i][3] = PT(0);
        for (int r = 0; r < n_ranges; ++r) {
            const int lo = r * G, hi = lo + G < k ? lo + G : k;
            __syncthreads();                                    // everyone is done with the previous range
            for (int e = threadIdx.x; e < G * (L / 4); e += kRgThreads) {
                const int jl = e / (L / 4), c4 = e % (L / 4);
                Vec4<PT> v;
                if (lo + jl < k) v.load(P + (size_t)(lo + jl) * L + c4 * 4);
                else v[0] = v[1] = v[2] = v[3] = PT(0);         // (a masked entry multiplies whatever gene its low bits name by 0)
                v.store(reinterpret_cast<PT*>(panel + (size_t)jl * kRowBytes) + c4 * 4);
            }
            __syncthreads();
            // one chunk: count the entries of this range (a prefix: columns ascend), multiply them, move the cursor
            auto process = [&](Chunk& ch, uint32_t& c, uint32_t e, PT (&a)[4][4]) -> bool {
                const int rem = (int)(e - c) - RPL * w4;          // entries of the row left from this lane's first record on
                int t8[RPL];
                PT val[RPL];
                int cnt = 0;
#pragma unroll
                for (int u = 0; u < RPL; ++u) {
                    const int j = ch.r[u].j;
                    const bool ok = u < rem && j < hi;
                    cnt += ok ? 1 : 0;
                    val[u] = ok ? (PT)ch.r[u].v : PT(0);
                    t8[u] = (j & (G - 1)) * kRowBytes;
                }
                cnt += __builtin_amdgcn_update_dpp(0, cnt, 0xb1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
                cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x4e, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (!__any(cnt > RPL * w)) break;
                    // kDeep entries at a time: their 4 kDeep panel reads, then their multiply-adds; the barrier keeps the
                    // scheduler from hoisting the reads of every entry of the chunk
#pragma unroll
                    for (int u0 = 0; u0 < RPL; u0 += kDeep) {
                        if (u0 > 0 && !__any(cnt > RPL * w + u0)) break;
                        Vec4<PT> pv[kDeep][4];
#pragma unroll
                        for (int d = 0; d < kDeep; ++d) {
                            const int tb = quad_bcast<4>(t8[u0 + d], w);
#pragma unroll
                            for (int i = 0; i < 4; ++i) pv[d][i].load(reinterpret_cast<const PT*>(panel + tb + pcb[i]));
                        }
#pragma unroll
                        for (int d = 0; d < kDeep; ++d) {
                            const PT v = quad_bcast_v<4>(val[u0 + d], w);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                a[i][0] += v * pv[d][i][0];
                                a[i][1] += v * pv[d][i][1];
                                a[i][2] += v * pv[d][i][2];
                                a[i][3] += v * pv[d][i][3];
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                c += (uint32_t)cnt;
                return __any(cnt == kChunk) != 0;
            };
            Chunk A, B;
            load_chunk(A, (uint32_t)quad_bcast<4>((int)curv, 0));
            static_for<S>([&](auto tag) {
                constexpr int s_ = decltype(tag)::value;
                uint32_t c = (uint32_t)quad_bcast<4>((int)curv, s_);
                const uint32_t e = (uint32_t)quad_bcast<4>((int)endv, s_);
                // the next slot's first chunk is in flight while this one is multiplied (its cursor is the previous phase's)
                if constexpr (s_ + 1 < S) load_chunk(B, (uint32_t)quad_bcast<4>((int)curv, s_ + 1));
                asm volatile("" ::: "memory");
                // (nothing of this slot's arithmetic above the loads: a compare of chunk A hoisted there makes the wave wait for A
                //  before it has issued B)
                __builtin_amdgcn_sched_barrier(0);
                bool more = process(A, c, e, acc[s_]);
                while (more) {                                  // a run longer than a chunk: rare (mean 18 entries of 32)
                    load_chunk(A, c);
                    more = process(A, c, e, acc[s_]);
                }
                curv = w4 == s_ ? c : curv;
                if constexpr (s_ + 1 < S) A = B;
            });
        }
        // the block's rows: centring term, then the scores (f64, obsm["X_pca"] layout) or the panel product
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            const uint64_t row = row_of(b, s_);
            if (row >= n_rows) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col0 = pc[i];
                Vec4<PT> cv4;
                cv4.load(cvec + col0);
                const PT o0 = acc[s_][i][0] - cv4[0], o1 = acc[s_][i][1] - cv4[1], o2 = acc[s_][i][2] - cv4[2], o3 = acc[s_][i][3] - cv4[3];
                if (scores) {
                    double* dst = scores + row * (uint64_t)ld + col0;
                    if (col0 + 3 < n_cols && ld % 2 == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0) {
                        *reinterpret_cast<double2*>(dst) = double2{(double)o0, (double)o1};
                        *reinterpret_cast<double2*>(dst + 2) = double2{(double)o2, (double)o3};
                    } else {
                        if (col0 + 0 < n_cols) dst[0] = (double)o0;
                        if (col0 + 1 < n_cols) dst[1] = (double)o1;
                        if (col0 + 2 < n_cols) dst[2] = (double)o2;
                        if (col0 + 3 < n_cols) dst[3] = (double)o3;
                    }
                } else {
                    Vec4<PT> o;
                    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
                    o.store(Y + row * L + col0);
                }
            }
        }
    }
}

// ---- transposed SpMM: T = A^T Y (k x l), s = 1^T Y ---------------------------------------------
// Workgroup = (gene tile, row block), 1024 threads; the tile's 256 x 64 accumulators live in
// LDS (128 KiB as f64).  A wave takes 16 consecutive cells at a time: their tile segments
// are ONE contiguous range of the tile-major arrays (coalesced 64-wide loads), lane c holds
// y[r][c] of the 16 rows in registers, and each non-zero is broadcast with v_readlane and
// scattered with one 64-lane LDS atomic add on 64 consecutive words (conflict-free).  The
// per-row-block partials are summed in fixed order by k_t_reduce.
constexpr int kTBatch = 16;

template <typename VT, typename YT, typename AT>
__global__ __launch_bounds__(kTThreads) void k_spmm_t(const int64_t* __restrict__ tptr,
                                                      const GramPk<VT>* __restrict__ tpk, uint64_t n_rows, int k, int nt,
                                                      uint64_t rows_per_block, const YT* __restrict__ Y,
                                                      AT* __restrict__ part /* [rb][k][L] */,
                                                      double* __restrict__ part_s /* [rb][L] */) {
    extern __shared__ double lds_raw[];
    AT* acc = reinterpret_cast<AT*>(lds_raw);
    for (int e = threadIdx.x; e < KT * L; e += kTThreads) acc[e] = AT(0);
    __syncthreads();
    const int tile = blockIdx.x % nt;
    const uint64_t rb = blockIdx.x / nt;
    const uint64_t r0 = rb * rows_per_block;
    const uint64_t r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
    const int lane = lane_id();
    const int wave = threadIdx.x / kWave;
    constexpr int kWaves = kTThreads / kWave;
    const int64_t* tp = tptr + (uint64_t)tile * n_rows;
    double ysum = 0.0;
    for (uint64_t rr = r0 + (uint64_t)wave * kTBatch; rr < r1; rr += (uint64_t)kWaves * kTBatch) {
        const int nb = (int)(r1 - rr < (uint64_t)kTBatch ? r1 - rr : (uint64_t)kTBatch);
        const int64_t myp = lane <= nb ? tp[rr + lane] : 0;
        AT y[kTBatch];
#pragma unroll
        for (int r = 0; r < kTBatch; ++r) y[r] = r < nb ? (AT)Y[(rr + r) * L + lane] : AT(0);
        if (tile == 0) {
#pragma unroll
            for (int r = 0; r < kTBatch; ++r) ysum += (double)y[r];
        }
        int64_t p[kTBatch + 1];
#pragma unroll
        for (int r = 0; r <= kTBatch; ++r) p[r] = readlane64(myp, r < nb ? r : nb);
        const int64_t pend = p[kTBatch];
        for (int64_t cb = p[0]; cb < pend; cb += kWave) {
            const int64_t pq = cb + lane;
            const GramPk<VT> e = tpk[pq < pend ? pq : p[0]];
            const int32_t ci = pq < pend ? e.j * L : 0;
            const VT cvv = pq < pend ? e.v : VT(0);
#pragma unroll
            for (int r = 0; r < kTBatch; ++r) {
                const int64_t a = p[r] > cb ? p[r] : cb;
                const int64_t b = p[r + 1] < cb + kWave ? p[r + 1] : cb + kWave;
                const int lo = (int)(a - cb), hi = (int)(b - cb);
                for (int s = lo; s < hi; ++s) {
                    const int j = __builtin_amdgcn_readlane(ci, s);
                    const AT v = (AT)readlane_v(cvv, s);
                    __hip_atomic_fetch_add(&acc[j + lane], v * y[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < KT * L; e += kTThreads) {
        int j = tile * KT + e / L;
        if (j < k) part[(rb * (uint64_t)k + j) * L + (e % L)] = acc[e];
    }
    if (tile == 0) {
        __shared__ double s_y[kWaves][L];
        s_y[wave][lane] = ysum;
        __syncthreads();
        if (threadIdx.x < L) {
            double t = 0.0;
            for (int w = 0; w < kWaves; ++w) t += s_y[w][threadIdx.x];
            part_s[rb * L + threadIdx.x] = t;
        }
    }
}


// T[k*L .. k*L+L) receives s.  Fixed summation order over the row blocks.
template <typename AT>
__global__ void k_t_reduce(const AT* __restrict__ part, const double* __restrict__ part_s, int k, uint64_t n_rb,
                           double* __restrict__ T) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t kl = (uint64_t)k * L;
    if (e < kl) {
        double s = 0.0;
        for (uint64_t b = 0; b < n_rb; ++b) s += (double)part[b * kl + e];
        T[e] = s;
    } else if (e < kl + L) {
        double s = 0.0;
        for (uint64_t b = 0; b < n_rb; ++b) s += part_s[b * L + (e - kl)];
        T[e] = s;
    }
}
