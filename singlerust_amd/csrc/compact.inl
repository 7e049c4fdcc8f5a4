// compact.inl — included by pca_form.hip inside namespace srx (the kernels share pca_internal.hpp's helpers and constants).
// HVG compaction of the CSR matrix to the selected features: counts, scans, fill passes (row-major records, tile-major views), the selection table in LDS, the owner-record format of the Gram kernel.

// ---- HVG compaction --------------------------------------------------------------------------
// remap[g] = position of gene g among the selected genes in ascending gene order, or -1.
__global__ __launch_bounds__(256) void k_compact_count(const int64_t* __restrict__ indptr,
                                                       const int32_t* __restrict__ idx,
                                                       const int32_t* __restrict__ remap, uint64_t n_rows,
                                                       int64_t* __restrict__ counts) {
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t lo = indptr[r], hi = indptr[r + 1];
        int c = 0;
        for (int64_t p = lo + lane; p < hi; p += kWave) c += remap[idx[p]] >= 0;
        c = wave_sum(c);
        if (lane == 0) counts[r] = c;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_compact_fill(const int64_t* __restrict__ indptr,
                                                      const int32_t* __restrict__ idx, const T* __restrict__ vals,
                                                      const int32_t* __restrict__ remap, uint64_t n_rows,
                                                      const int64_t* __restrict__ out_ptr,
                                                      int32_t* __restrict__ out_idx, T* __restrict__ out_vals) {
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t lo = indptr[r], hi = indptr[r + 1];
        int64_t o = out_ptr[r];
        for (int64_t base = lo; base < hi; base += kWave) {
            int64_t p = base + lane;
            int32_t c = p < hi ? remap[idx[p]] : -1;
            unsigned long long mask = __ballot(c >= 0);
            if (c >= 0) {
                int pos = __popcll(mask & ((1ull << lane) - 1ull));
                out_idx[o + pos] = c;
                out_vals[o + pos] = vals[p];
            }
            o += __popcll(mask);
        }
    }
}

// ---- exclusive scan of int64 counts (3 phases, 4096 elements per block) ------------------------
constexpr int kScanItems = 4;
constexpr int kScanBlock = 1024;
__global__ __launch_bounds__(kScanBlock) void k_scan_block_sums(const int64_t* __restrict__ in, uint64_t n,
                                                                int64_t* __restrict__ block_sums) {
    __shared__ int64_t s_w[kScanBlock / kWave];
    uint64_t base = (uint64_t)blockIdx.x * kScanBlock * kScanItems;
    int64_t s = 0;
    for (int t = 0; t < kScanItems; ++t) {
        uint64_t i = base + (uint64_t)threadIdx.x * kScanItems + t;
        if (i < n) s += in[i];
    }
    s = wave_sum(s);
    if (lane_id() == 0) s_w[threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t tot = 0;
        for (int w = 0; w < kScanBlock / kWave; ++w) tot += s_w[w];
        block_sums[blockIdx.x] = tot;
    }
}
// Exclusive scan of the block sums in place by ONE workgroup (nb is n/4096: a few thousand).
__global__ __launch_bounds__(kScanBlock) void k_scan_serial(int64_t* __restrict__ block_sums, uint64_t nb,
                                                            int64_t* __restrict__ total) {
    __shared__ int64_t s_w[kScanBlock / kWave];
    const uint64_t per = (nb + kScanBlock - 1) / kScanBlock;
    const uint64_t b0 = (uint64_t)threadIdx.x * per;
    const uint64_t b1 = b0 + per < nb ? b0 + per : nb;
    int64_t s = 0;
    for (uint64_t b = b0; b < b1; ++b) s += block_sums[b];
    int64_t inc = s;
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        int64_t o = __shfl_up(inc, off, kWave);
        if (lane >= off) inc += o;
    }
    if (lane == kWave - 1) s_w[threadIdx.x / kWave] = inc;
    __syncthreads();
    int64_t wave_off = 0, all = 0;
    for (int w = 0; w < kScanBlock / kWave; ++w) {
        if (w < (int)(threadIdx.x / kWave)) wave_off += s_w[w];
        all += s_w[w];
    }
    int64_t acc = wave_off + inc - s;
    for (uint64_t b = b0; b < b1; ++b) {
        int64_t v = block_sums[b];
        block_sums[b] = acc;
        acc += v;
    }
    if (threadIdx.x == 0) *total = all;
}
__global__ __launch_bounds__(kScanBlock) void k_scan_apply(const int64_t* __restrict__ in, uint64_t n,
                                                           const int64_t* __restrict__ block_offs,
                                                           const int64_t* __restrict__ total,
                                                           int64_t* __restrict__ out /* n + 1 */) {
    __shared__ int64_t s_w[kScanBlock / kWave];
    uint64_t base = (uint64_t)blockIdx.x * kScanBlock * kScanItems;
    int64_t v[kScanItems];
    int64_t s = 0;
    for (int t = 0; t < kScanItems; ++t) {
        uint64_t i = base + (uint64_t)threadIdx.x * kScanItems + t;
        v[t] = i < n ? in[i] : 0;
        s += v[t];
    }
    // inclusive scan of the per-thread sums across the wave, then across waves
    int64_t inc = s;
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        int64_t o = __shfl_up(inc, off, kWave);
        if (lane >= off) inc += o;
    }
    if (lane == kWave - 1) s_w[threadIdx.x / kWave] = inc;
    __syncthreads();
    int64_t wave_off = 0;
    for (int w = 0; w < (int)(threadIdx.x / kWave); ++w) wave_off += s_w[w];
    int64_t excl = block_offs[blockIdx.x] + wave_off + inc - s;
    for (int t = 0; t < kScanItems; ++t) {
        uint64_t i = base + (uint64_t)threadIdx.x * kScanItems + t;
        if (i < n) out[i] = excl;
        excl += v[t];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;
}

// ---- tile-major layout of the compacted matrix --------------------------------------------------
// The compacted N x k matrix is stored as n_t = ceil(k / kt) sub-matrices, one per GENE TILE of
// kt compacted columns, back to back: sub-matrix t holds, row by row, the entries of every
// cell that fall in columns [kt t, kt t + kt), with LOCAL column indices and row pointers
// tptr[t*N + i].  A workgroup that owns (tile, row range) therefore streams ONE contiguous
// index/value range, fully coalesced, instead of ~9-entry pieces of 1.3M rows.
//   kt = 256 (KT)  SpMM kernels: 256 x 64 panel entries are what LDS holds — 64 KiB as f32
//                  (forward panel tile, two workgroups per CU), 128 KiB as f64 (transposed
//                  accumulators, one workgroup per CU);
//   kt = 128 (KG)  Gram kernel: a 128 x 128 f64 tile of A^T A is 128 KiB.
__global__ void k_seglen(const int64_t* __restrict__ indptr, const int64_t* __restrict__ tp, uint64_t n_rows, int nt,
                         int64_t* __restrict__ seglen) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t total = (uint64_t)nt * n_rows;
    for (; e < total; e += stride) {
        uint64_t t = e / n_rows, i = e % n_rows;
        int64_t lo = t == 0 ? indptr[i] : tp[(t - 1) * n_rows + i];
        int64_t hi = t == (uint64_t)nt - 1 ? indptr[i + 1] : tp[t * n_rows + i];
        seglen[e] = hi - lo;
    }
}

// One entry of a tile-major layout: local column and value side by side, so that every consumer (Gram kernel,
// forward / transposed SpMM) fetches an entry with ONE 8-byte (f32 storage) or 16-byte (f64) load and the
// compaction writes it with one store.
template <typename T>
__global__ __launch_bounds__(256) void k_retile(const int64_t* __restrict__ indptr, const int64_t* __restrict__ tp,
                                                const int32_t* __restrict__ idx, const T* __restrict__ vals,
                                                uint64_t n_rows, int nt, int kt, const int64_t* __restrict__ tptr,
                                                GramPk<T>* __restrict__ tpk) {
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t lo = indptr[r], hi = indptr[r + 1];
        for (int64_t p = lo + lane; p < hi; p += kWave) {
            int32_t c = idx[p];
            int t = c / kt;
            int64_t seg_lo = t == 0 ? lo : tp[(uint64_t)(t - 1) * n_rows + r];
            int64_t dst = tptr[(uint64_t)t * n_rows + r] + (p - seg_lo);
            GramPk<T> e{};
            e.j = c - t * kt;
            e.v = vals[p];
            tpk[dst] = e;
        }
    }
}

// ---- fused HVG compaction straight into the tile-major layouts (<= 64 tiles of 128) ----------------
// Pass 1 (k_tcount): per cell, the number of kept entries in each 128-column tile (and, summed in
// pairs, in each 256-column tile).  Kept entries of a row are sorted by compacted column, so the
// tiles present in a 64-entry chunk are found with a short ballot-match loop; lane t of the wave
// is the counter of tile t.  Pass 2 (k_tfill) re-reads the row and scatters every kept entry to
//   tptr[tile*N + i] + (rank of the entry among the row's kept entries - kept entries in earlier tiles)
// in BOTH layouts.  The scans of the counts in between give tptr.
// Membership + compacted column of a gene WITHOUT a G-entry remap table in L2 (a 4-byte gather per
// non-zero drags a 64-byte line each: 70 GB of L2 traffic at c3): the selection is a bitmask
// (G/32 words) plus the number of selected genes before each word, both staged in LDS (7 KB at
// G = 28k); column = prefix[w] + popcount(bits[w] below the gene's bit).
// ---- owner buckets of the Gram kernel (k_gram_stripes, below) -----------------------------------------
struct SelLds {
    const uint32_t* bits;
    const uint32_t* prefix;
    __device__ __forceinline__ int column(int32_t gene) const {
        const uint32_t w = bits[gene >> 5];
        const uint32_t bit = 1u << (gene & 31);
        return (w & bit) ? (int)(prefix[gene >> 5] + __popc(w & (bit - 1u))) : -1;
    }
};
__device__ __forceinline__ SelLds stage_selection(const uint32_t* __restrict__ g_bits,
                                                  const uint32_t* __restrict__ g_prefix, int n_words, uint32_t* lds) {
    for (int e = threadIdx.x; e < n_words; e += blockDim.x) {
        lds[e] = g_bits[e];
        lds[n_words + e] = g_prefix[e];
    }
    __syncthreads();
    return SelLds{lds, lds + n_words};
}

constexpr int kFillUnroll = 4;       // 64-entry chunks of a row in flight per wave in k_tfill
constexpr int kCompactRows = 8;      // consecutive rows per wave visit (one 64-byte line of 8-byte per-row counters)
// rows r0*8 .. r0*8+7 of a wave's block, then the block n_waves further on
__device__ __forceinline__ uint64_t next_compact_row(uint64_t r, uint64_t n_waves) {
    return ((r + 1) % kCompactRows) ? r + 1 : r + 1 + (n_waves - 1) * kCompactRows;
}

// Kept entries per row, nothing else (the row-major layout's row lengths; the tile counters of k_tcount are only wanted by
// the matrix-free solver's 256-tiled view): the bit of the gene in the selection mask is the whole test — one LDS read per
// entry, no prefix lookup, no LDS atomic.  The mask holds at most k bits (k_sel_finish / the host route make sure).
template <typename I>
__global__ __launch_bounds__(256) void k_rowcount(const int64_t* __restrict__ indptr, const I* __restrict__ idx,
                                                  const uint32_t* __restrict__ g_bits, int n_words, uint64_t n_rows,
                                                  int64_t* __restrict__ cntrow) {
    extern __shared__ double lds_raw[];
    uint32_t* bits = reinterpret_cast<uint32_t*>(lds_raw);
    for (int e = threadIdx.x; e < n_words; e += blockDim.x) bits[e] = g_bits[e];
    __syncthreads();
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    constexpr int kCountUnroll = 16;          // 1024 entries in flight: a ~840-entry row is one round trip
    for (uint64_t r0 = wave * kCompactRows; r0 < n_rows; r0 += n_waves * kCompactRows) {
        const int nr = (int)(n_rows - r0 < (uint64_t)kCompactRows ? n_rows - r0 : kCompactRows);
        uint32_t mine = 0;                    // lane i: row r0 + i
        for (int i = 0; i < nr; ++i) {
            const int64_t lo = indptr[r0 + i], hi = indptr[r0 + i + 1];
            uint32_t c = 0;
            for (int64_t base = lo; base < hi; base += kCountUnroll * kWave) {
                int32_t g[kCountUnroll];
#pragma unroll
                for (int u = 0; u < kCountUnroll; ++u) {
                    const int64_t p = base + u * kWave + lane;
                    g[u] = p < hi ? (int32_t)idx[p] : -1;
                }
#pragma unroll
                for (int u = 0; u < kCountUnroll; ++u)
                    if (g[u] >= 0) c += (bits[g[u] >> 5] >> (g[u] & 31)) & 1u;
            }
            c = wave_sum(c);
            if (lane == i) mine = c;
        }
        if (lane < nr) cntrow[r0 + lane] = (int64_t)mine;     // 8 counters = one 64-byte line
    }
}

// Two-pass compaction, second form (round 3): the count pass also LEAVES A LIST of what it found — per kept entry one 32-bit
// word (position in the row << 16 | compacted column) at kept[indptr[r] + rank], i.e. at the start of the row's own span of a
// scratch array as long as the matrix — so that the fill pass never walks the column indices again: per row it reads its
// ~72 words (one coalesced load), gathers the ~72 values and stores the entries.  Needs n_cols <= 65536 (16-bit positions and
// columns) and the row-major layout alone (no 256-tiled view).
// Round 5: the same list, 0.82 -> 0.67 ms.  The selection is expanded into a 16-bit table in LDS — gene -> compacted column,
// 0xffff = dropped, one entry more for the lanes behind a row's end — so that the test of an entry is ONE 2-byte LDS read at the
// gene's own address (round 4: the mask word and the prefix word, two reads: the LDS pipe was its bound); a lane's rank comes
// from the two mbcnt instructions with the row's running count as their addend, the list word from one OR with a scalar, the
// store address from the row's base in scalar registers and the rank alone — 11 instructions per 64-entry slot, written out
// (inline assembly: no branch inside a slot).  A wave walks its rows as a sequence of BATCHES (up to 1024 entries of one row) with
// two register sets: the 16 loads of the next batch — always 16, so that the wait before a batch's work is a fixed count — are
// in flight while this one is worked on.  What is left (profiles/r05_knockouts.md): without its loads the pass takes 0.68 ms,
// without its stores 0.64, the scalar skeleton around the slots alone 0.4 — it runs at the rate its waves issue instructions,
// and a second resident workgroup per CU only takes the issue slots the first one leaves (358 / 584 us for the two).
constexpr int kCountThreads = 1024;
// a value every lane holds alike, told to the compiler (scalar registers, scalar branches)
__device__ __forceinline__ int64_t uniform64(int64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__global__ __launch_bounds__(kCountThreads) void k_rowcount_list(const int64_t* __restrict__ indptr, const uint16_t* __restrict__ idx,
                                                                 const uint32_t* __restrict__ g_bits, const uint32_t* __restrict__ g_prefix,
                                                                 int n_words, uint64_t n_rows, uint64_t nnz, int k,
                                                                 int64_t* __restrict__ cntrow, uint32_t* __restrict__ kept) {
    extern __shared__ double lds_raw[];
    uint16_t* col = reinterpret_cast<uint16_t*>(lds_raw);          // 32 n_words + 1 entries
    const int n_genes = n_words * 32;
    for (int g = threadIdx.x; g <= n_genes; g += blockDim.x) {
        uint32_t c = 0xffffu;
        if (g < n_genes) {
            const uint32_t w = g_bits[g >> 5], bit = 1u << (g & 31);
            if (w & bit) {
                c = g_prefix[g >> 5] + (uint32_t)__popc(w & (bit - 1u));
                if (c >= (uint32_t)k) c = 0xffffu;       // only a broken selection (NaN variances) has such columns: dropped
            }
        }
        col[g] = (uint16_t)c;
    }
    __syncthreads();
    const uint32_t behind = (uint32_t)n_genes;                     // the table's last entry: dropped
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    const uint32_t lane16 = (uint32_t)lane << 16;
    constexpr int kU = 16;                    // 1024 entries per batch: a ~840-entry row is one
    constexpr uint32_t kBatch = kU * kWave;
    struct Cur {                              // a batch: entries [b, b + 1024) of row r (all wave-uniform)
        uint64_t r;
        int64_t lo;
        uint32_t n, b;
        int64_t plo, phi;                     // the row pointers of the row BEHIND r in this wave's order, asked for when r was opened:
    };                                        // a row's pointers arrive a batch before they are wanted (a scalar load per row in
                                              // the wave's way cost 630 of a batch's 4500 cycles)
    auto ask_next = [&](Cur& c) {
        uint64_t rn = next_compact_row(c.r, n_waves);
        if (rn >= n_rows) rn = n_rows - 1;     // (not followed: any valid row)
        rn = (uint64_t)uniform64((int64_t)rn);
        c.plo = indptr[rn];
        c.phi = indptr[rn + 1];
    };
    auto open_row = [&](Cur& c, uint64_t r) {  // the first row of the wave
        r = (uint64_t)uniform64((int64_t)r);
        c.r = r;
        c.lo = uniform64(indptr[r]);
        c.n = (uint32_t)(uniform64(indptr[r + 1]) - c.lo);
        c.b = 0;
        ask_next(c);
    };
    auto open_next = [&](Cur& c, uint64_t r) { // r = the row behind c.r: its pointers are there
        c.r = (uint64_t)uniform64((int64_t)r);
        c.lo = uniform64(c.plo);
        c.n = (uint32_t)(uniform64(c.phi) - c.lo);
        c.b = 0;
        ask_next(c);
    };
    // the batch behind c, in this wave's order of rows: 1 = there it is, 0 = no row is left, 2 = c is now at a row whose batches
    // may run past the array's end (one of its last rows: the plain loop at the end takes over)
    auto advance = [&](Cur& c) -> int {
        if (c.b + kBatch < c.n) {
            c.b += kBatch;
            return 1;
        }
        const uint64_t r = (uint64_t)uniform64((int64_t)next_compact_row(c.r, n_waves));
        if (r >= n_rows) return 0;
        open_next(c, r);
        return ((uint64_t)c.lo + c.n + kBatch <= nnz) ? 1 : 2;
    };
    // the 16 loads of a batch, unpredicated (what lies behind the row's end is the next rows' entries: replaced by `behind`
    // before use; the rows whose batches could run past the ARRAY's end are left to the plain loop at the end).  Inline
    // assembly: the compiler waits for ALL outstanding loads before the first use of any (it cannot tell the two register sets
    // apart across the loop); the wait is written out below, as a count.
    static_assert(kU == 16, "16 loads of 2-byte indices");
    auto issue = [&](uint32_t (&g)[kU], const Cur& c) {
        const unsigned long long base = (unsigned long long)uniform64((int64_t)(idx + c.lo + c.b));      // (a scalar register pair)
        const uint32_t off = (uint32_t)lane * 2u;
        asm volatile(
            "global_load_ushort %0, %16, %17\n\tglobal_load_ushort %1, %16, %17 offset:128\n\t"
            "global_load_ushort %2, %16, %17 offset:256\n\tglobal_load_ushort %3, %16, %17 offset:384\n\t"
            "global_load_ushort %4, %16, %17 offset:512\n\tglobal_load_ushort %5, %16, %17 offset:640\n\t"
            "global_load_ushort %6, %16, %17 offset:768\n\tglobal_load_ushort %7, %16, %17 offset:896\n\t"
            "global_load_ushort %8, %16, %17 offset:1024\n\tglobal_load_ushort %9, %16, %17 offset:1152\n\t"
            "global_load_ushort %10, %16, %17 offset:1280\n\tglobal_load_ushort %11, %16, %17 offset:1408\n\t"
            "global_load_ushort %12, %16, %17 offset:1536\n\tglobal_load_ushort %13, %16, %17 offset:1664\n\t"
            "global_load_ushort %14, %16, %17 offset:1792\n\tglobal_load_ushort %15, %16, %17 offset:1920"
            : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7]),
              "=&v"(g[8]), "=&v"(g[9]), "=&v"(g[10]), "=&v"(g[11]), "=&v"(g[12]), "=&v"(g[13]), "=&v"(g[14]), "=&v"(g[15])
            : "v"(off), "s"(base)
            : "memory");
    };
    // the same batch by plain loads clamped to the array's last entry (the compiler's own waits): the array's last rows
    auto issue_plain = [&](uint32_t (&g)[kU], const Cur& c) {
        const uint32_t last = (uint32_t)(nnz - 1 - ((uint64_t)c.lo + c.b));      // (nnz > lo + b: the row has entries there)
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t o = (uint32_t)(u * kWave + lane);
            g[u] = (uint32_t)idx[c.lo + c.b + (o < last ? o : last)];
        }
    };
    // Everything but the 16 most recent vector-memory operations — the loads of the batch behind this one — has completed: this
    // batch's registers hold its loads.  (Stores count in vmcnt like loads: the wait also covers the stores of the batch worked
    // on before.  Leaving those outstanding as well — a wait count of 16 + that batch's slots — was measured: 0.48 ms SLOWER.)
    auto arrived = [&](uint32_t (&g)[kU]) {
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        // (the registers are handed to the work below by these two statements only: tests/test_abi_cpu.py checks the listing
        //  for reads of a set between its loads and here)
        asm volatile("" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]));
        asm volatile("" : "+v"(g[8]), "+v"(g[9]), "+v"(g[10]), "+v"(g[11]), "+v"(g[12]), "+v"(g[13]), "+v"(g[14]), "+v"(g[15]));
    };
    uint32_t rank0 = 0, mine = 0;             // kept entries of the row before this slot (wave-uniform); lane i: count of row (r & ~7) + i
    auto work = [&](uint32_t (&g)[kU], const Cur& c) {
        const int rem = __builtin_amdgcn_readfirstlane((int)(c.n - c.b));      // (wave-uniform: the tests on it are scalar branches)
        if (c.b == 0) rank0 = 0;
        char* krow = reinterpret_cast<char*>(kept + c.lo);
        const int left = rem - lane;           // > 64 u: this lane's entry of slot u exists
        // (groups of four slots: one that lies inside the row reads the table at its genes as they are; only the group holding the
        //  row's end sends the lanes behind it to the table's last entry; a group behind the end is not looked at)
#pragma unroll
        for (int q = 0; q < kU / 4; ++q) {
            if (rem <= q * 4 * kWave) break;
            if (rem < (q + 1) * 4 * kWave) {
#pragma unroll
                for (int j = 0; j < 4; ++j) g[4 * q + j] = col[left > (4 * q + j) * kWave ? g[4 * q + j] : behind];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) g[4 * q + j] = col[g[4 * q + j]];       // (the reads of all groups in flight together)
            }
        }
        // Four slots per statement, no branch inside (the listing the compiler makes of the same steps spends 2.6 branches per slot
        // — around the store, at the row's end —, and the pass waits on instruction issue, not on memory): the kept lanes become
        // the exec mask, their ranks come from the mbcnt pair over the running count, the list word from one OR with the slot's
        // scalar position, one store, and the count moves on by the mask's population.  A slot behind the row's end holds 0xffff
        // everywhere: an empty mask.
        uint32_t vr = rank0;                   // (the running count in a vector register: mbcnt's addend)
        const unsigned long long kr = (unsigned long long)krow;
#define SRX_COUNT_SLOT(C)                                     \
    "v_cmp_ne_u32_e32 vcc, 0xffff, " C "\n\t"                 \
    "s_and_saveexec_b64 %[sv], vcc\n\t"                      \
    "v_mbcnt_lo_u32_b32 %[t], vcc_lo, %[vr]\n\t"             \
    "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"              \
    "v_or3_b32 %[w], " C ", %[l16], %[p]\n\t"                \
    "v_lshlrev_b32_e32 %[t], 2, %[t]\n\t"                    \
    "global_store_dword %[t], %[w], %[kr]\n\t"               \
    "s_mov_b64 exec, %[sv]\n\t"                              \
    "s_bcnt1_i32_b64 %[nk], vcc\n\t"                         \
    "v_add_u32_e32 %[vr], %[nk], %[vr]\n\t"                  \
    "s_add_i32 %[p], %[p], 0x400000\n\t"
#pragma unroll
        for (int q = 0; q < kU / 4; ++q) {
            if (rem <= q * 4 * kWave) break;   // (wave-uniform)
            unsigned long long sv;
            uint32_t nk, t, w, p16 = (c.b + (uint32_t)(q * 4 * kWave)) << 16;
            asm volatile(SRX_COUNT_SLOT("%[c0]") SRX_COUNT_SLOT("%[c1]") SRX_COUNT_SLOT("%[c2]") SRX_COUNT_SLOT("%[c3]")
                         : [vr] "+v"(vr), [p] "+s"(p16), [sv] "=&s"(sv), [nk] "=&s"(nk), [t] "=&v"(t), [w] "=&v"(w)
                         : [c0] "v"(g[4 * q]), [c1] "v"(g[4 * q + 1]), [c2] "v"(g[4 * q + 2]), [c3] "v"(g[4 * q + 3]), [l16] "v"(lane16),
                           [kr] "s"(kr)
                         : "vcc", "scc", "memory");
        }
#undef SRX_COUNT_SLOT
        rank0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)vr);
        if (c.b + kBatch >= c.n) {             // the row's last batch: its count; 8 counters = one 64-byte line
            const int i = __builtin_amdgcn_readfirstlane((int)(c.r % kCompactRows));
            if (lane == i) mine = rank0;
            if (i == kCompactRows - 1 || c.r + 1 == n_rows) {
                if (lane <= i) cntrow[c.r - (uint64_t)i + lane] = (int64_t)mine;
            }
        }
    };
    if (wave * kCompactRows >= n_rows) return;
    Cur a;
    open_row(a, wave * kCompactRows);
    uint32_t gA[kU], gB[kU];
    if ((uint64_t)a.lo + a.n + kBatch <= nnz) {
        // Two batches per turn, straight-line: the loads behind a batch are ALWAYS issued (with nothing behind it: the same batch
        // again, into the set that is not used any more), so that no branch separates a set's loads from its wait.
        issue(gA, a);
        int st;
        for (;;) {
            Cur b = a;
            st = __builtin_amdgcn_readfirstlane(advance(b));
            issue(gB, st == 1 ? b : a);
            arrived(gA);
            work(gA, a);
            a = b;
            if (st != 1) break;
            Cur c = b;
            st = __builtin_amdgcn_readfirstlane(advance(c));
            issue(gA, st == 1 ? c : b);
            arrived(gB);
            work(gB, b);
            a = c;
            if (st != 1) break;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the loads issued behind the last batch)
        if (st == 0) {
            return;
        }
    }
    // the array's last rows (a batch's 1024 entries may run past the array's end): one batch at a time, plain loads
    for (;;) {
        if (a.n != 0) issue_plain(gA, a);
        work(gA, a);
        if (a.b + kBatch < a.n) {
            a.b += kBatch;
            continue;
        }
        const uint64_t r = (uint64_t)uniform64((int64_t)next_compact_row(a.r, n_waves));
        if (r >= n_rows) break;
        open_next(a, r);
    }
}

// XF: `vals` are the raw values and the kept entries are stored as ln_1p(f64(v) * scale_row), rounded once (RowXf).
template <typename T, bool XF>
__global__ __launch_bounds__(256) void k_tfill_list(const int64_t* __restrict__ indptr, const T* __restrict__ vals,
                                                    const uint32_t* __restrict__ kept, uint64_t n_rows,
                                                    const int64_t* __restrict__ rm_ptr, const double* __restrict__ row_sum,
                                                    double target, GramPk<T>* __restrict__ rm) {
    __shared__ Log1pTabEntry s_tab[XF ? 128 : 1];
    if constexpr (XF) {
        stage_log1p_table(s_tab);
        __syncthreads();
    }
    // the 64 entries behind the last row are READ by the forward kernel (a row's last chunk runs past its end: value masked to
    // 0, column used as is): they must name a real column, or 0 x panel[garbage] is NaN (this used to be a memset launch)
    if (blockIdx.x == 0 && threadIdx.x < 64) rm[rm_ptr[n_rows] + threadIdx.x] = GramPk<T>{};
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    // a wave takes two rows at a time, half a wave each (a row keeps ~72 of its entries: two steps of 32); the rows of a pair
    // are neighbours, so their words, their values and their entries are neighbours too
    for (uint64_t r0 = wave * 2; r0 < n_rows; r0 += n_waves * 2) {
        const uint64_t r = r0 + (lane >> 5);
        const bool live = r < n_rows;
        const int64_t lo = live ? indptr[r] : 0;
        const int64_t o0 = live ? rm_ptr[r] : 0;
        const int n = live ? (int)(rm_ptr[r + 1] - o0) : 0;
        double scale = 1.0;
        if constexpr (XF) {
            const double sr = live ? row_sum[r] : 0.0;
            scale = sr == 0.0 ? 0.0 : target / sr;      // scale/mod.rs:9-15
        }
        const int n_max = __builtin_amdgcn_readfirstlane(max(__shfl(n, 0, kWave), __shfl(n, 32, kWave)));
        for (int t = lane & 31; t < n_max; t += 32) {
            if (t < n) {
                const uint32_t w = kept[lo + t];
                T v = vals[lo + (w >> 16)];
                if constexpr (XF) v = xf_stored(v, scale, s_tab);         // the value the write-back stores in X
                GramPk<T> e{};
                e.j = (int32_t)(w & 0xffffu);
                e.v = v;
                rm[o0 + t] = e;
            }
        }
    }
}

template <typename I>
__global__ __launch_bounds__(256) void k_tcount(const int64_t* __restrict__ indptr, const I* __restrict__ idx,
                                                const uint32_t* __restrict__ g_bits,
                                                const uint32_t* __restrict__ g_prefix, int n_words, uint64_t n_rows,
                                                int nt128, int nt256, int k, int64_t* __restrict__ cntrow,
                                                int64_t* __restrict__ cnt256) {
    extern __shared__ double lds_raw[];
    const SelLds sel = stage_selection(g_bits, g_prefix, n_words, reinterpret_cast<uint32_t*>(lds_raw));
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    // per-wave tile counters in LDS (after the selection table): a kept entry is one ds_add_u32 on its
    // tile's counter — ~5 active lanes per 64-entry chunk — instead of a ballot-match loop over the tiles
    // present in the chunk (the loop made this pass VALU-bound);
    // one 64-counter row per row of the wave's current block of kCompactRows rows
    uint32_t* tcnt = reinterpret_cast<uint32_t*>(lds_raw) + 2 * n_words + (threadIdx.x / kWave) * (kCompactRows * kWave);
#pragma unroll
    for (int i = 0; i < kCompactRows; ++i) tcnt[i * kWave + lane] = 0u;
    // A wave takes kCompactRows CONSECUTIVE rows at a time and writes their counters out together: lane (tile, row)
    // stores 8 bytes next to its 7 neighbours, i.e. one full 64-byte line per tile — stored row by row, the 24
    // strided 8-byte counters of a row cost ~44 bytes of HBM write each (1.39 GB written for 0.25 GB of counters).
    // Lane l of a chunk takes entry l (2-byte loads): consecutive entries of a row are ~1 bitmask word apart, so
    // the 64 lookups of a chunk fall into 64 different LDS banks — 8 consecutive entries per lane (16-byte loads)
    // were tried and cost an 8-way bank conflict per lookup.  kCountUnroll chunks (1024 entries) are issued
    // together: with 4 a ~840-entry row was 4 dependent round trips to HBM.
    constexpr int kCountUnroll = 16;
    for (uint64_t r0 = wave * kCompactRows; r0 < n_rows; r0 += n_waves * kCompactRows) {
        const int nr = (int)(n_rows - r0 < (uint64_t)kCompactRows ? n_rows - r0 : kCompactRows);
        for (int i = 0; i < nr; ++i) {
            const int64_t lo = indptr[r0 + i], hi = indptr[r0 + i + 1];
            uint32_t* row_cnt = tcnt + i * kWave;
            for (int64_t base = lo; base < hi; base += kCountUnroll * kWave) {
                int32_t g[kCountUnroll];
#pragma unroll
                for (int u = 0; u < kCountUnroll; ++u) {
                    const int64_t p = base + u * kWave + lane;
                    g[u] = p < hi ? (int32_t)idx[p] : -1;
                }
#pragma unroll
                for (int u = 0; u < kCountUnroll; ++u) {
                    int col = g[u] >= 0 ? sel.column(g[u]) : -1;             // -1 for dropped entries
                    if (col >= k) col = -1;          // only a broken selection (NaN variances) has such columns: dropped
                    if (col >= 0) __hip_atomic_fetch_add(&row_cnt[col >> 7], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // lane q -> (tile q / 8, row q % 8)
        if (lane < nr) {                                   // kept entries of row r0 + lane (the row-major layout's row length)
            uint32_t tot = 0;
            for (int t = 0; t < nt128; ++t) tot += tcnt[lane * kWave + t];
            cntrow[r0 + lane] = (int64_t)tot;
        }
        for (int q = lane; q < nt256 * kCompactRows; q += kWave) {
            const int t = q / kCompactRows, i = q % kCompactRows;
            if (i < nr) cnt256[(uint64_t)t * n_rows + r0 + i] = (int64_t)(tcnt[i * kWave + 2 * t] + tcnt[i * kWave + 2 * t + 1]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int i = 0; i < kCompactRows; ++i) tcnt[i * kWave + lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// XF: `vals` are the raw values and the kept entries are stored as ln_1p(f64(v) * scale_row), rounded once (RowXf).
template <typename T, typename I, bool XF>
__global__ __launch_bounds__(256) void k_tfill(const int64_t* __restrict__ indptr, const I* __restrict__ idx,
                                               const T* __restrict__ vals, const uint32_t* __restrict__ g_bits,
                                               const uint32_t* __restrict__ g_prefix, int n_words, uint64_t n_rows,
                                               int nt256, int k, const int64_t* __restrict__ cnt256,
                                               const int64_t* __restrict__ rm_ptr,
                                               const int64_t* __restrict__ tptr256, const double* __restrict__ row_sum,
                                               double target, GramPk<T>* __restrict__ rm, GramPk<T>* __restrict__ pk256) {
    extern __shared__ double lds_raw[];
    __shared__ Log1pTabEntry s_tab[XF ? 128 : 1];
    if constexpr (XF) stage_log1p_table(s_tab);
    const SelLds sel = stage_selection(g_bits, g_prefix, n_words, reinterpret_cast<uint32_t*>(lds_raw));
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    for (uint64_t r = (wave * kCompactRows); r < n_rows; r = next_compact_row(r, n_waves)) {
        const int64_t lo = indptr[r], hi = indptr[r + 1];
        // lane t: kept entries before 256-tile t in this row (exclusive prefix over the tile counters)
        const int c_t = lane < nt256 ? (int)cnt256[(uint64_t)lane * n_rows + r] : 0;
        int inc = c_t;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const int o = __shfl_up(inc, off, kWave);
            if (lane >= off) inc += o;
        }
        const int before256 = inc - c_t;
        // destination bias of each tile: tptr - (kept entries before the tile)
        const int64_t off256 = (lane < nt256 ? tptr256[(uint64_t)lane * n_rows + r] : 0) - before256;
        const int64_t row_base = rm_ptr[r];
        double scale = 1.0, row_table = 0.0;
        if constexpr (XF) {
            const double sr = row_sum[r];
            scale = sr == 0.0 ? 0.0 : target / sr;      // scale/mod.rs:9-15
        }
        (void)row_table;
        int rank0 = 0;                                  // kept entries of the row before this chunk
        // (tried and slower at c3: 16 chunks in flight, 2.42 ms — the ballots / shuffles of the masked-out tail
        //  chunks cost more than the loads gain; parking the kept entries in LDS and writing them out per row,
        //  2.59 ms — the value gather then waits for the whole row and the stage halves the occupancy; bucketing the
        //  entries by Gram owner here, one workgroup per row block: 2.65 ms against 1.3 + a separate 0.4 ms pass)
        for (int64_t base = lo; base < hi; base += kFillUnroll * kWave) {
            int32_t g[kFillUnroll];
#pragma unroll
            for (int u = 0; u < kFillUnroll; ++u) {
                const int64_t p = base + u * kWave + lane;
                g[u] = p < hi ? (int32_t)idx[p] : -1;
            }
#pragma unroll
            for (int u = 0; u < kFillUnroll; ++u) {
                const int64_t p = base + u * kWave + lane;
                int32_t c = g[u] >= 0 ? sel.column(g[u]) : -1;
                if (c >= k) c = -1;
                const unsigned long long mask = __ballot(c >= 0);
                const int cc = c >= 0 ? c : 0;
                const int64_t o256 = __shfl(off256, cc >> 8, kWave);   // shuffles run with all lanes active
                if (c >= 0) {
                    const int rank = rank0 + __popcll(mask & ((1ull << lane) - 1ull));
                    T v = vals[p];
                    if constexpr (XF) v = xf_stored(v, scale, s_tab);         // the value the write-back stores in X
                    GramPk<T> e{};
                    e.j = c;
                    e.v = v;
                    rm[row_base + rank] = e;
                    if (nt256 > 0) {
                        e.j = c & 255;
                        pk256[o256 + rank] = e;
                    }
                }
                rank0 += __popcll(mask);
            }
        }
    }
}
