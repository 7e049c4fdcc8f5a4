// genes.hip — per-gene (column-direction) passes over the row-major CSR, and the statistics
// entry points of the C ABI.
//
// Per-gene accumulation from a CSR is a scatter: global atomics (one per non-zero, 3.3e9
// for the 1.3M-cell config) are two orders of magnitude too slow, so the accumulators are
// privatised in LDS.  3 accumulators x G genes do not fit 160 KiB, hence GENE TILING: the
// gene axis is cut into tiles of <= 8000 genes (20 B of LDS each: u32 count + f64 sum + f64
// sum of squares); because column indices are sorted inside a row, a tile's entries are one
// contiguous segment of every row, found once per sparsity pattern by binary search
// (k_tile_ptr) and cached on the matrix.  A 1024-thread workgroup (one per CU: it owns the
// whole LDS) takes one (gene tile, row block), walks the row segments one wave per row with
// coalesced loads of indices/values, accumulates with LDS atomics, and flushes its tile to
// a per-block partial buffer; a second tiny kernel sums the partials in fixed order.
//
// Algorithmic bytes: nnz*(4 + s_v) + (N+1)*8 + G*24.
#include "common.hpp"
#include "log1p64.hpp"

#include <algorithm>
#include <cmath>

namespace srx {

constexpr int kMaxTileGenes = 10000;      // 10000 * 16 B (two 64-bit sums per gene) + the 2 KiB logarithm table <= 163840 B of LDS.
                                          // c3 (28k genes): 3 tiles of 9334 — a row's tile segment is ~280 entries instead of ~210 with
                                          // 4 tiles of 7000 (8000 * 20 B while the pass also counted): moments + store 3.09 -> see DESIGN;
                                          // the other way round, 5 / 6 tiles: 3.28 / 3.66 ms (round 4 A/B, SRX_EXP_TILE_GENES)
constexpr int kMomThreads = 1024;

// lower_bound of `bound` inside each row's sorted column list, for the interior tile cuts.
__global__ void k_tile_ptr(const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx,
                           uint64_t n_rows, int n_tiles, int tile_genes, int64_t* __restrict__ tp) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t total = (uint64_t)(n_tiles - 1) * n_rows;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        uint64_t t = i / n_rows + 1, r = i % n_rows;
        int32_t bound = (int32_t)(t * (uint64_t)tile_genes);
        int64_t lo = indptr[r], hi = indptr[r + 1];
        while (lo < hi) {
            int64_t mid = lo + ((hi - lo) >> 1);
            if (idx[mid] < bound) lo = mid + 1; else hi = mid;
        }
        tp[i] = lo;
    }
}

__device__ __forceinline__ void seg_bounds(const int64_t* __restrict__ indptr, const int64_t* __restrict__ tp,
                                           uint64_t n_rows, int n_tiles, int tile, uint64_t r,
                                           int64_t& lo, int64_t& hi) {
    lo = tile == 0 ? indptr[r] : tp[(uint64_t)(tile - 1) * n_rows + r];
    hi = tile == n_tiles - 1 ? indptr[r + 1] : tp[(uint64_t)tile * n_rows + r];
}

// The two value passes of the reference over the columns — scatter-add of x (csr.rs:94-100) and of x^2 (csr.rs:175-178) — in
// ONE walk; the third, the histogram of the indices (csr.rs:29-36), depends on the pattern only and is k_gene_count's.
// XF: the values are the RAW matrix and x = ln_1p(f64(v) * scale_row) is formed on the fly in f64 (RowXf, common.hpp):
// the moments of the normalised + log1p'd matrix to f64 accuracy whatever the storage type, before (and without) the
// in-place write-back.
// WB (with XF): the transformed value is also stored back in place, at the storage precision — every stored value belongs to
// exactly one (gene tile, row) segment, i.e. to one lane of one workgroup, so this pass IS the in-place normalise + log1p of
// the pipeline and nothing reads the raw matrix after it.
// (Round 3, measured and not kept: for COUNT DATA a per-segment value table — each lane makes ONE logarithm per group of row
//  segments, ln_1p(c * scale) for c = 1 .. 32, parked in the LDS the cached counts leave free; a value then costs an exactness
//  test and one ds_read_b64 instead of the ~30-instruction logarithm.  Bit-identical results, 16x fewer logarithms, and
//  3.64 ms against 3.18: the four dependent LDS reads of a chunk queue behind the other waves' atomics, and the arithmetic
//  they replace was what hid that latency.)
// BL (round 6): the accumulators in BLOCKS of 32 genes — 32 sums (256 bytes: every bank once), then the 32 sums of squares — instead
// of {sum, sum of squares} pairs side by side.  With the pairs a wave's 64 sum atomics only ever touch the banks 0, 1 mod 4 and its
// 64 square atomics the banks 2, 3 mod 4: half of the LDS's bank pairs per instruction (68 % of the LDS pipe's active cycles were
// bank conflicts, profiles/r05_pmc_gram.md).  One more VALU instruction per value for the address; the second atomic stays at an
// immediate offset (256 bytes).
template <typename T, typename I, bool XF, bool WB = false, bool BL = false>
__global__ __launch_bounds__(kMomThreads) void k_gene_moments(
    const int64_t* __restrict__ indptr, const int64_t* __restrict__ tp, const I* __restrict__ idx,
    std::conditional_t<WB, T, const T>* __restrict__ vals, uint64_t n_rows, uint64_t n_cols, int n_tiles, int tile_genes,
    uint64_t rows_per_block, const double* __restrict__ row_sum, double target, double fx_sum, double fx_sq,
    uint32_t* __restrict__ poison, double* __restrict__ part_sum, double* __restrict__ part_sq) {
    extern __shared__ double lds[];
    double* s_acc = lds;            // per gene {sum, sum of squares} side by side: one address computation per value, the second
                                    // atomic at the instruction's immediate offset
    // behind the accumulators (tile_genes * 16 B): the log1p table
    // (the table from global memory instead — 2 KB, L1-resident, one vector load per value — was tried: 5.7 ms against 3.5)
    const int tile_pad = BL ? (tile_genes + 31) & ~31 : tile_genes;
    constexpr int kSqOff = BL ? 32 : 1;                      // 8-byte words between a gene's sum and its sum of squares
    auto acc_slot = [](int g0) -> int { return BL ? ((g0 >> 5) << 6) + (g0 & 31) : 2 * g0; };
    const Log1pTabEntry* s_tab = reinterpret_cast<const Log1pTabEntry*>(reinterpret_cast<char*>(lds) + (size_t)tile_pad * 16);
    if constexpr (XF) stage_log1p_table(const_cast<Log1pTabEntry*>(s_tab));
    for (int g = threadIdx.x; g < 2 * tile_pad; g += kMomThreads) s_acc[g] = 0.0;
    __syncthreads();

    // The gene tiles of one row block run on the SAME XCD (consecutive workgroup ids go round the 8 XCDs): a row's tile segments
    // share their boundary cache lines — read by two tiles, and each tile's in-place stores cover only part of them — and
    // with the tiles on four different L2s every such line was fetched (and, written, merged) once per tile: 15.7 GB of HBM
    // traffic for 10.9 GB of data (profiles/r03_traffic_c3.json).
    int tile;
    uint64_t rb;
    {
        const uint64_t n_rb = gridDim.x / (unsigned)n_tiles;
        if (n_rb % 8 == 0) {
            const uint64_t slot = blockIdx.x / 8;
            tile = (int)(slot % (unsigned)n_tiles);
            rb = (slot / (unsigned)n_tiles) * 8 + blockIdx.x % 8;
        } else {
            tile = blockIdx.x % n_tiles;
            rb = blockIdx.x / n_tiles;
        }
    }
    const int32_t gbase = tile * tile_genes;
    const uint64_t r0 = rb * rows_per_block;
    const uint64_t r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);      // row bounds and scales are then scalars
    constexpr int kWaves = kMomThreads / kWave;

    // 4 consecutive entries per lane: one 16-byte (8-byte for 16-bit indices) load of the indices and one (f32) or two (f64) of
    // the values, from a 4-entry boundary; entries outside the segment are masked (the arrays are padded by 16 entries).
    // A row's segment in one of 3 gene tiles is ~280 entries, a wave step is 256 slots: one segment per step left the second
    // step of nearly every segment at 10 % fill behind a dependent load — and the pass is bound by the latency of its chains at
    // 4 waves per SIMD (60 VALU instructions per value, profiles/r04_pmc_gram.md).  So a wave works on BATCHES of kBR row segments laid end to end in one slot space
    // (slot = 4 entries of one segment; segment u owns the slots S[u] .. S[u + 1]): 4 segments = ~284 slots = 5 steps at 89 %
    // fill; a lane finds its segment with kBR - 1 compares against the (wave-uniform) prefix sums and selects the segment's
    // base, offset, length and scale.  kBS steps of loads are in flight all the time (a step of the current batch is evaluated, then
    // the same step of the next batch is fetched into the registers it freed); longer batches take their remaining steps with
    // plain loads.  (4 segments / 5 steps fill better, 89 %, and are slower: 2.88 against 2.82 ms.)
    constexpr int kBR = sizeof(T) == 4 ? 3 : 2;      // f64 values: half the batch (a chunk is 10 registers instead of 6)
    constexpr int kBS = sizeof(T) == 4 ? 4 : 3;
    struct Chunk {
        std::conditional_t<sizeof(I) == 4, int4, uint2> gi;       // the four column indices as loaded (16-bit ones stay packed)
        T v[4];
    };
    auto gene_of = [](const Chunk& c, int j) -> int {
        if constexpr (sizeof(I) == 4) return j == 0 ? c.gi.x : j == 1 ? c.gi.y : j == 2 ? c.gi.z : c.gi.w;
        else return (int)(((j < 2 ? c.gi.x : c.gi.y) >> (16 * (j & 1))) & 0xffffu);
    };
    auto load_chunk = [&](int64_t e0, Chunk& c) {
        if constexpr (sizeof(I) == 4) c.gi = *reinterpret_cast<const int4*>(idx + e0);
        else c.gi = *reinterpret_cast<const uint2*>(idx + e0);        // four 16-bit indices in one 8-byte load
        if constexpr (sizeof(T) == 4) {
            const float4 t4 = *reinterpret_cast<const float4*>(vals + e0);
            c.v[0] = t4.x; c.v[1] = t4.y; c.v[2] = t4.z; c.v[3] = t4.w;
        } else {
            const double2 a2 = *reinterpret_cast<const double2*>(vals + e0);
            const double2 b2 = *reinterpret_cast<const double2*>(vals + e0 + 2);
            c.v[0] = a2.x; c.v[1] = a2.y; c.v[2] = b2.x; c.v[3] = b2.y;
        }
    };
    // XF: the sums are kept in FIXED POINT (2^-sum_shift, 2^-sq_shift; 64-bit integer LDS atomics): integer addition is
    // associative, so the moments do not depend on the order the atomics land in — two genes with identical columns get
    // identical sums, as in the reference's sequential loops (HighlyVariable(n) breaks exact ties by gene index), and the
    // run is reproducible to the bit.  [Raw f32 values widened to f64 sum exactly anyway: the plain path keeps f64 adds.]
    // The magic constant's bit pattern is NOT subtracted per value: the accumulators sum bits(1.5 * 2^52 + x * 2^shift) =
    // bits(1.5 * 2^52) + round(x * 2^shift) in wrapping 64-bit arithmetic and k_moments_reduce takes count x bits(1.5 * 2^52)
    // off again (the per-gene counts are known).  Positions are tested as 32-bit offsets from the segment start.
    auto add_chunk = [&](int64_t e0, int rel /* position of the chunk's first entry in its segment: >= -3 */, unsigned len,
                         const Chunk& c, double scale) {
        // XF: the four logarithms first, as independent straight-line chains the compiler can interleave (the accumulators
        // leave room for 4 waves per SIMD: a wave has to bring its own instruction-level parallelism); the rare argument
        // classes are patched afterwards (entries outside the segment hold padding values: evaluated, never used)
        double y4[4] = {0.0, 0.0, 0.0, 0.0};
        if constexpr (XF) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double x = (double)c.v[j] * scale;
                if constexpr (sizeof(T) == 4) y4[j] = log1p_f64_moment_common(x, s_tab);
                else y4[j] = log1p_f64_fast(x, s_tab);
            }
            if constexpr (sizeof(T) == 4) {
                bool any_rare = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) any_rare |= log1p_f64_moment_is_rare((double)c.v[j] * scale);
                if (__builtin_expect(any_rare, 0)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const double x = (double)c.v[j] * scale;
                        if (log1p_f64_moment_is_rare(x)) y4[j] = log1p_f64_rare(x, s_tab);
                    }
                }
            }
        }
        if constexpr (XF && WB) {
            // the chunk's values go back as one 16-byte store (two for f64) when all four belong to the segment
            T o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (T)y4[j];
            if (rel >= 0 && (unsigned)(rel + 3) < len) {
                // non-temporal: nothing reads a stored value again in this pass (2.77 -> 2.73 ms; non-temporal LOADS cost 0.4 ms:
                // the boundary lines of a row's tile segments are shared by two workgroups through L2)
                if constexpr (sizeof(T) == 4) {
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(f4v{(float)o[0], (float)o[1], (float)o[2], (float)o[3]}, reinterpret_cast<f4v*>(vals + e0));
                } else {
                    typedef double d2v __attribute__((ext_vector_type(2)));
                    __builtin_nontemporal_store(d2v{(double)o[0], (double)o[1]}, reinterpret_cast<d2v*>(vals + e0));
                    __builtin_nontemporal_store(d2v{(double)o[2], (double)o[3]}, reinterpret_cast<d2v*>(vals + e0 + 2));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((unsigned)(rel + j) < len) vals[e0 + j] = o[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if ((unsigned)(rel + j) < len) {
                const int32_t g0 = gene_of(c, j) - gbase;
                double x0 = (double)c.v[j];
                if constexpr (XF) {
                    x0 = y4[j];
                    if (__builtin_expect(!(fabs(x0) < 64.0), 0)) {     // NaN, infinite, or outside the fixed-point range: the gene's moments are NaN
                        poison[(uint64_t)gbase + g0] = 1u;
                        continue;
                    }
                    // round-to-nearest integer of x * 2^shift through the 1.5 * 2^52 trick (|x * 2^shift| < 2^51)
                    const double kMagic = 6755399441055744.0;
                    const unsigned long long is = (unsigned long long)__double_as_longlong(__builtin_fma(x0, fx_sum, kMagic));
                    const unsigned long long iq = (unsigned long long)__double_as_longlong(__builtin_fma(x0 * x0, fx_sq, kMagic));
                    // (a poisoned value adds nothing while the reduction still takes its magic bits off: the gene's sums are
                    //  replaced by NaN anyway)
                    unsigned long long* a2 = reinterpret_cast<unsigned long long*>(s_acc) + acc_slot(g0);
                    __hip_atomic_fetch_add(a2, is, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(a2 + kSqOff, iq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    continue;
                }
                __hip_atomic_fetch_add(&s_acc[acc_slot(g0)], x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&s_acc[acc_slot(g0) + kSqOff], x0 * x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };
    // (all offsets of a row block fit 32 bits: block_geometry keeps rows_per_block x n_cols below 2^31)
    int64_t blk_lo, blk_hi_;
    seg_bounds(indptr, tp, n_rows, n_tiles, tile, r0 < n_rows ? r0 : 0, blk_lo, blk_hi_);
    const int64_t base = blk_lo & ~(int64_t)3;
    struct Batch {                   // wave-uniform but for `c`: five scalars per segment
        int eb[kBR];                 // (4-entry boundary at or before the segment start) - base - 4 S: entry of slot s = base + eb + 4 s
        int q[kBR];                  // 4 S + (segment start - its boundary): position of slot s's first entry in the segment = 4 s - q
        int len[kBR];                // segment length
        int S[kBR + 1];              // first slot of segment u; S[kBR] = slots of the batch
        double scale[kBR];
    };
    // slot `slot` of the batch -> its chunk's first entry, that entry's position in the segment, the segment's length / scale
    // (X_sel = X[0] + sum_u [slot >= S[u]] (X[u] - X[u - 1]) in wrapping 32-bit arithmetic — exact for integers and for the two
    //  halves of a double's bit pattern alike: the differences are wave-uniform (SALU), a lane pays one v_cndmask per segment
    //  boundary and word and one add; a chain of selects between SGPR values costs three VALU instructions per select, the
    //  constant bus taking one scalar operand per instruction)
    auto locate = [&](const Batch& g, int slot, int64_t& e0, int& rel, unsigned& len, double& scale) {
        const unsigned long long sc0 = (unsigned long long)__double_as_longlong(g.scale[0]);
        unsigned eb = (unsigned)g.eb[0], q = (unsigned)g.q[0], ln = (unsigned)g.len[0], slo = (unsigned)sc0, shi = (unsigned)(sc0 >> 32);
#pragma unroll
        for (int u = 1; u < kBR; ++u) {
            const bool in = slot >= g.S[u];
            const unsigned long long a = (unsigned long long)__double_as_longlong(g.scale[u]),
                                     b = (unsigned long long)__double_as_longlong(g.scale[u - 1]);
            eb += in ? (unsigned)g.eb[u] - (unsigned)g.eb[u - 1] : 0u;
            q += in ? (unsigned)g.q[u] - (unsigned)g.q[u - 1] : 0u;
            ln += in ? (unsigned)g.len[u] - (unsigned)g.len[u - 1] : 0u;
            slo += in ? (unsigned)a - (unsigned)b : 0u;
            shi += in ? (unsigned)(a >> 32) - (unsigned)(b >> 32) : 0u;
        }
        e0 = base + (int64_t)(int)(eb + 4u * (unsigned)slot);
        rel = (int)(4u * (unsigned)slot - q);
        len = ln;
        scale = __longlong_as_double((long long)(((unsigned long long)shi << 32) | slo));
    };
    auto describe = [&](Batch& g, uint64_t rbase) {
        int S = 0;
#pragma unroll
        for (int u = 0; u < kBR; ++u) {
            const uint64_t r = rbase + (uint64_t)u * kWaves;
            int64_t lo = base, hi = base;
            g.scale[u] = 1.0;
            if (r < r1) {
                seg_bounds(indptr, tp, n_rows, n_tiles, tile, r, lo, hi);
                if constexpr (XF) {
                    const double sr = row_sum[r];
                    g.scale[u] = sr == 0.0 ? 0.0 : target / sr;            // scale/mod.rs:9-15
                }
            }
            const int64_t b0 = lo & ~(int64_t)3;
            const int a = (int)(lo - b0);
            g.len[u] = (int)(hi - lo);
            g.S[u] = S;
            g.eb[u] = (int)(b0 - base) - 4 * S;
            g.q[u] = 4 * S + a;
            S += (a + g.len[u] + 3) >> 2;
        }
        g.S[kBR] = S;
    };
    // the load of step `st` of a batch: unconditional (slots past the batch read the first segment's first chunk again: a branch
    // around a load makes the compiler's wait for one step's data wait for every load issued since)
    auto issue = [&](const Batch& g, int st, Chunk& c) {
        const int slot = st * kWave + lane;
        int64_t e0;
        int rel;
        unsigned len;
        double sc;
        locate(g, slot < g.S[kBR] ? slot : 0, e0, rel, len, sc);
        load_chunk(e0, c);
    };
    auto step = [&](const Batch& g, int st, const Chunk& c) {
        int slot = st * kWave + lane;
        // (recomputed here, opaquely: left to itself the compiler keeps the located values of every prefetched step alive from
        //  `issue` on — spilled)
        asm volatile("" : "+v"(slot));
        if (slot < g.S[kBR]) {
            int64_t e0;
            int rel;
            unsigned len;
            double sc;
            locate(g, slot, e0, rel, len, sc);
            add_chunk(e0, rel, len, c, sc);
        }
    };
    {
        // ONE set of kBS chunk registers: step st of the current batch is evaluated, then step st of the NEXT batch is fetched into
        // the registers it has just freed — kBS steps of loads in flight all the time with half the registers two whole batches
        // took (which left the scheduler no room to overlap the logarithms of two steps)
        const uint64_t stride = (uint64_t)kWaves * kBR;
        Batch cur, nxt;
        Chunk c[kBS];
        uint64_t rbase = r0 + wave;
        describe(cur, rbase);
#pragma unroll
        for (int st = 0; st < kBS; ++st) issue(cur, st, c[st]);
        while (rbase < r1) {
            describe(nxt, rbase + stride);
#pragma unroll
            for (int st = 0; st < kBS; ++st) {
                if (st * kWave < cur.S[kBR]) step(cur, st, c[st]);          // (wave-uniform)
                issue(nxt, st, c[st]);
            }
            // batches longer than kBS steps (long rows): the rest, a step at a time
            for (int slot = kBS * kWave + lane; slot < cur.S[kBR]; slot += kWave) {
                int64_t e0;
                int rel;
                unsigned len;
                double sc;
                locate(cur, slot, e0, rel, len, sc);
                Chunk cc;
                load_chunk(e0, cc);
                add_chunk(e0, rel, len, cc, sc);
            }
            cur = nxt;
            rbase += stride;
        }
    }
    __syncthreads();
    for (int g = threadIdx.x; g < tile_genes; g += kMomThreads) {
        uint64_t gene = (uint64_t)gbase + g;
        if (gene < n_cols) {
            part_sum[rb * n_cols + gene] = s_acc[acc_slot(g)];
            part_sq[rb * n_cols + gene] = s_acc[acc_slot(g) + kSqOff];
        }
    }
}

// Per-gene non-zero counts of the stored pattern (csr.rs:24-36: the column histogram of the indices) — a property of the
// sparsity pattern alone: made once per pattern (srx_matrix_prepare, or the first pass that needs them), kept on the matrix,
// inherited by clones; the moments passes then carry no count atomic and no counters in LDS (16 B per gene: 3 gene tiles at
// 28k genes instead of 4).  Workgroup = (row block, gene tile) like the moments pass, 32-bit LDS counters, flushed with global
// partials `cnt[row block][gene]`.
template <typename I>
__global__ __launch_bounds__(kMomThreads) void k_gene_count(const int64_t* __restrict__ indptr, const int64_t* __restrict__ tp,
                                                            const I* __restrict__ idx, uint64_t n_rows, uint64_t n_cols, int n_tiles,
                                                            int tile_genes, uint64_t rows_per_block, uint32_t* __restrict__ cnt) {
    extern __shared__ double lds[];
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(lds);
    for (int g = threadIdx.x; g < tile_genes; g += kMomThreads) s_cnt[g] = 0u;
    __syncthreads();
    const int tile = blockIdx.x % n_tiles;
    const uint64_t rb = blockIdx.x / n_tiles;
    const int32_t gbase = tile * tile_genes;
    const uint64_t r0 = rb * rows_per_block;
    const uint64_t r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    constexpr int kWaves = kMomThreads / kWave;
    // four row segments per step; a lane takes 4 consecutive entries from the 4-entry boundary at or before the segment start
    // (one 8-byte load of 16-bit indices, 16 bytes of 32-bit ones: 256 entries per wave instruction — one entry per lane and
    // load left this pass at 1.3-1.7 TB/s), two such chunks of every segment in flight, entries outside [lo, hi) masked
    // (the arrays are padded by 16 entries)
    constexpr int kU = 2, kR = 4;
    auto load4 = [&](int64_t e0, int32_t (&g)[4]) {
        if constexpr (sizeof(I) == 4) {
            const int4 g4 = *reinterpret_cast<const int4*>(idx + e0);
            g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
        } else {
            const uint2 g2 = *reinterpret_cast<const uint2*>(idx + e0);
            g[0] = (int)(g2.x & 0xffffu); g[1] = (int)(g2.x >> 16);
            g[2] = (int)(g2.y & 0xffffu); g[3] = (int)(g2.y >> 16);
        }
    };
    auto add4 = [&](int64_t e0, int64_t lo, int64_t hi, const int32_t (&g)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (e0 + j >= lo && e0 + j < hi)
                __hip_atomic_fetch_add(&s_cnt[g[j] - gbase], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    for (uint64_t rb4 = r0 + wave; rb4 < r1; rb4 += (uint64_t)kWaves * kR) {
        int64_t lo[kR], hi[kR];
#pragma unroll
        for (int q = 0; q < kR; ++q) {
            const uint64_t r = rb4 + (uint64_t)q * kWaves;
            lo[q] = hi[q] = 0;
            if (r < r1) seg_bounds(indptr, tp, n_rows, n_tiles, tile, r, lo[q], hi[q]);
        }
        int32_t g[kR][kU][4];
#pragma unroll
        for (int q = 0; q < kR; ++q)
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int64_t b0 = lo[q] & ~(int64_t)3, e0 = b0 + 4 * (u * kWave + lane);
                load4(e0 < hi[q] ? e0 : b0, g[q][u]);
            }
#pragma unroll
        for (int q = 0; q < kR; ++q) {
            const int64_t b0 = lo[q] & ~(int64_t)3;
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int64_t e0 = b0 + 4 * (u * kWave + lane);
                if (e0 < hi[q]) add4(e0, lo[q], hi[q], g[q][u]);
            }
            // segments longer than kU x 256 entries: the rest, a chunk at a time
            for (int64_t e0 = b0 + 4 * (kU * kWave + lane); e0 < hi[q]; e0 += 4 * kWave) {
                int32_t gt[4];
                load4(e0, gt);
                add4(e0, lo[q], hi[q], gt);
            }
        }
    }
    __syncthreads();
    // per-row-block partials, summed by k_gene_count_reduce (global atomics into the 28k counters — 341 row blocks on every
    // address — took ~0.9 of this pass's 1.4 ms)
    for (int g = threadIdx.x; g < tile_genes; g += kMomThreads) {
        const uint64_t gene = (uint64_t)gbase + g;
        if (gene < n_cols) cnt[rb * n_cols + gene] = s_cnt[g];
    }
}
__global__ void k_gene_count_reduce(const uint32_t* __restrict__ part, uint64_t n_cols, uint64_t n_blocks, uint32_t* __restrict__ cnt) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cols) return;
    uint32_t c = 0;
    for (uint64_t b = 0; b < n_blocks; ++b) c += part[b * n_cols + j];
    cnt[j] = c;
}

// Fixed-order sum of the per-row-block partials -> packed f64 [cnt | sum | sq | n_rows].
// `cnt`: this shard's per-gene non-zero counts (pattern-only, k_gene_count).
// Workgroup = 64 genes x kRedSlices slices of the row blocks (a thread per gene alone: 111 workgroups of one dependent walk over
// all 341 partials each, 45 us at c3 — a fixed cost of the step however few rows a rank holds).  The fixed-point partials are
// integers: the slices add in any order.
constexpr int kRedSlices = 8;
__global__ __launch_bounds__(64 * kRedSlices) void k_moments_reduce(
    const double* __restrict__ part_sum, const double* __restrict__ part_sq, uint64_t n_cols, uint64_t n_blocks, uint64_t n_rows,
    const uint32_t* __restrict__ cnt, double inv_fx_sum, double inv_fx_sq, const uint32_t* __restrict__ poison,
    double* __restrict__ packed) {
    __shared__ unsigned long long red[2][kRedSlices][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const uint64_t j = (uint64_t)blockIdx.x * 64 + lane;
    if (blockIdx.x == 0 && threadIdx.x == 0) packed[3 * n_cols] = (double)n_rows;
    const bool live = j < n_cols;
    double s = 0.0, q = 0.0;
    if (inv_fx_sum != 0.0) {                 // fixed-point partials (transformed values): exact integer sums
        unsigned long long is = 0, iq = 0;
        if (live) {
            // (four blocks' partials in flight per thread)
            uint64_t b = slice;
            for (; b + 3 * kRedSlices < n_blocks; b += 4 * kRedSlices) {
                unsigned long long xs[4], xq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xs[u] = (unsigned long long)__double_as_longlong(part_sum[(b + u * kRedSlices) * n_cols + j]);
                    xq[u] = (unsigned long long)__double_as_longlong(part_sq[(b + u * kRedSlices) * n_cols + j]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    is += xs[u];
                    iq += xq[u];
                }
            }
            for (; b < n_blocks; b += kRedSlices) {
                is += (unsigned long long)__double_as_longlong(part_sum[b * n_cols + j]);
                iq += (unsigned long long)__double_as_longlong(part_sq[b * n_cols + j]);
            }
        }
        red[0][slice][lane] = is;
        red[1][slice][lane] = iq;
        __syncthreads();
        if (slice != 0 || !live) return;
#pragma unroll
        for (int u = 1; u < kRedSlices; ++u) {
            is += red[0][u][lane];
            iq += red[1][u][lane];
        }
        // every contribution carried the bit pattern of the 1.5 * 2^52 rounding constant along (k_gene_moments): off again
        const unsigned long long cn = (unsigned long long)cnt[j];
        const unsigned long long magic_bits = (unsigned long long)__double_as_longlong(6755399441055744.0);
        s = (double)(long long)(is - cn * magic_bits) * inv_fx_sum;
        q = (double)(long long)(iq - cn * magic_bits) * inv_fx_sq;
        if (poison[j]) s = q = __builtin_nan("");
    } else {
        // f64 partials of the raw values: ONE fixed order (the row blocks in sequence), by the first slice
        if (slice != 0 || !live) return;
        for (uint64_t b = 0; b < n_blocks; ++b) {
            s += part_sum[b * n_cols + j];
            q += part_sq[b * n_cols + j];
        }
    }
    packed[j] = (double)cnt[j];
    packed[n_cols + j] = s;
    packed[2 * n_cols + j] = q;
}

__global__ void k_moments_unpack(const double* __restrict__ packed, uint64_t n_cols, uint64_t* __restrict__ cnt,
                                 double* __restrict__ sum, double* __restrict__ sq) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cols) return;
    cnt[j] = (uint64_t)packed[j];
    sum[j] = packed[n_cols + j];
    sq[j] = packed[2 * n_cols + j];
}

// ---- per-gene min / max (csr.rs:212-221) ------------------------------------------------------
// Order-preserving map double -> u64 so LDS integer atomics (ds_min_u64 / ds_max_u64) apply.
__device__ __host__ __forceinline__ uint64_t f64_key(double x) {
    uint64_t b;
    memcpy(&b, &x, 8);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __host__ __forceinline__ double key_f64(uint64_t k) {
    uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double x;
    memcpy(&x, &b, 8);
    return x;
}

template <typename T>
__global__ __launch_bounds__(kMomThreads) void k_gene_minmax(
    const int64_t* __restrict__ indptr, const int64_t* __restrict__ tp, const int32_t* __restrict__ idx,
    const T* __restrict__ vals, uint64_t n_rows, uint64_t n_cols, int n_tiles, int tile_genes,
    uint64_t rows_per_block, unsigned long long* __restrict__ g_min, unsigned long long* __restrict__ g_max) {
    extern __shared__ double lds[];
    unsigned long long* s_min = reinterpret_cast<unsigned long long*>(lds);
    unsigned long long* s_max = s_min + tile_genes;
    const unsigned long long kInf = f64_key(INFINITY), kNinf = f64_key(-INFINITY);
    for (int g = threadIdx.x; g < tile_genes; g += kMomThreads) { s_min[g] = kInf; s_max[g] = kNinf; }
    __syncthreads();
    const int tile = blockIdx.x % n_tiles;
    const uint64_t rb = blockIdx.x / n_tiles;
    const int32_t gbase = tile * tile_genes;
    const uint64_t r0 = rb * rows_per_block;
    const uint64_t r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);      // row bounds and scales are then scalars
    constexpr int kWaves = kMomThreads / kWave;
    for (uint64_t r = r0 + wave; r < r1; r += kWaves) {
        int64_t lo, hi;
        seg_bounds(indptr, tp, n_rows, n_tiles, tile, r, lo, hi);
        for (int64_t p = lo + lane; p < hi; p += kWave) {
            double x = (double)vals[p];
            if (x != x) continue;                      // f64::min/max ignore a NaN operand
            int32_t g = idx[p] - gbase;
            unsigned long long k = f64_key(x);
            atomicMin(&s_min[g], k);
            atomicMax(&s_max[g], k);
        }
    }
    __syncthreads();
    for (int g = threadIdx.x; g < tile_genes; g += kMomThreads) {
        uint64_t gene = (uint64_t)gbase + g;
        if (gene < n_cols) {
            if (s_min[g] != kInf) atomicMin(&g_min[gene], s_min[g]);
            if (s_max[g] != kNinf) atomicMax(&g_max[gene], s_max[g]);
        }
    }
}

__global__ void k_fill_u64(unsigned long long* p, uint64_t n, unsigned long long v) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// normalize_total(Column) on CSR: scale/mod.rs:141-173, values[j] *= scale[col].
template <typename T>
__global__ void k_col_scale(const int32_t* __restrict__ idx, T* __restrict__ vals, uint64_t nnz,
                            const double* __restrict__ sum, double target) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < nnz; i += stride) {
        double s = sum[idx[i]];
        double sc = s == 0.0 ? 0.0 : target / s;      // scale/mod.rs:93-99
        vals[i] = (T)((double)vals[i] * sc);
    }
}

static void tile_geometry(const srx_mat* m, int& n_tiles, int& tile_genes) {
    uint64_t G = m->n_cols ? m->n_cols : 1;
    n_tiles = (int)((G + kMaxTileGenes - 1) / kMaxTileGenes);
    tile_genes = (int)((G + n_tiles - 1) / n_tiles);
}

int32_t launch_tile_ptr(srx_ctx* ctx, const int64_t* indptr, const int32_t* idx, uint64_t n_rows, int n_tiles,
                        int tile_genes, int64_t* tp) {
    uint64_t total = (uint64_t)(n_tiles - 1) * n_rows;
    uint64_t g = (total + 255) / 256;
    if (g < 1) g = 1;
    if (g > 65535) g = 65535;
    hipLaunchKernelGGL(k_tile_ptr, dim3((unsigned)g), dim3(256), 0, ctx->stream, indptr, idx, n_rows, n_tiles,
                       tile_genes, tp);
    SRX_HIP(ctx, hipGetLastError());
    return SRX_OK;
}

// d_indices -> 16-bit mirror (entries past nnz: 0, the arrays are padded for the vector walks)
__global__ void k_narrow16(const int32_t* __restrict__ idx, uint64_t nnz, uint64_t n_out, uint16_t* __restrict__ out) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; e < n_out; e += stride) out[e] = e < nnz ? (uint16_t)idx[e] : (uint16_t)0;
}

// Both pattern-only structures of a fresh CSR matrix in ONE walk over its 32-bit column indices: the 16-bit mirror and the
// NB = n_tiles - 1 interior gene-tile cuts of every row.  A wave per row, 256 entries per step; a row is sorted by column, so
// the cut at `bound` is the row's start + the number of entries below it = the popcounts of the waves' `idx < bound` ballots
// (scalar instructions; no search, no second pass).  k_tile_ptr's binary searches + k_narrow16 read the indices twice and
// moved 10.9 GB at c3 (2.3 ms of a cold step's 14.7); this moves 6.6.
// WIDEN: the other way round — the 16-bit mirror came over PCIe (upload_on: a quarter of the host's index bytes cross the
// link) and the 32-bit indices are made from it, with the same cuts.
// COUNT (round 6): the walk also makes the per-gene non-zero counts (csr.rs:24-36) — a 32-bit counter per gene in LDS (n_cols <= 38 000:
// 152 KB), one LDS atomic per entry, per-workgroup partials summed by k_gene_count_reduce — so that a handle leaves the upload / the
// generation with ALL its pattern-only structures and the first pipeline call on it pays no k_gene_count (0.54 + 0.12 ms at c3: VERDICT
// r5 item 5).  1024-thread workgroups then (the counters hold a CU to one workgroup).
template <int NB, bool WIDEN, bool COUNT = false>
__global__ __launch_bounds__(COUNT ? 1024 : 256) void k_narrow16_tiles(const int64_t* __restrict__ indptr, int32_t* idx,
                                                        uint64_t n_rows, int tile_genes, uint16_t* out,
                                                        int64_t* __restrict__ tp, uint32_t n_cols = 0, uint32_t* __restrict__ cnt_part = nullptr) {
    extern __shared__ uint32_t s_hist[];
    if constexpr (COUNT) {
        for (uint32_t g = threadIdx.x; g < n_cols; g += blockDim.x) s_hist[g] = 0u;
        __syncthreads();
    }
    const int lane = lane_id();
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x / kWave);
    constexpr int kU = 4;
    for (uint64_t r = global_wave_id(); r < n_rows; r += n_waves) {
        const int64_t lo = indptr[r], hi = indptr[r + 1];
        int cnt[NB > 0 ? NB : 1];
#pragma unroll
        for (int b = 0; b < NB; ++b) cnt[b] = 0;
        for (int64_t e0 = lo; e0 < hi; e0 += kU * kWave) {
            int32_t v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int64_t e = e0 + u * kWave + lane;
                if constexpr (WIDEN) v[u] = e < hi ? (int32_t)out[e] : 0x7fffffff;
                else v[u] = e < hi ? idx[e] : 0x7fffffff;
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int64_t e = e0 + u * kWave + lane;
                if (e < hi) {
                    if constexpr (WIDEN) idx[e] = v[u];
                    else out[e] = (uint16_t)v[u];
                    if constexpr (COUNT) {
                        if ((uint32_t)v[u] < n_cols) __hip_atomic_fetch_add(&s_hist[v[u]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) cnt[b] += __popcll(__ballot(v[u] < (b + 1) * tile_genes));
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b) tp[(uint64_t)b * n_rows + r] = lo + cnt[b];
        }
    }
    // (the 16 padding entries behind the last non-zero)
    if (blockIdx.x == 0 && threadIdx.x < 16) out[indptr[n_rows] + threadIdx.x] = (uint16_t)0;
    if constexpr (COUNT) {
        __syncthreads();
        for (uint32_t g = threadIdx.x; g < n_cols; g += blockDim.x) cnt_part[(uint64_t)blockIdx.x * n_cols + g] = s_hist[g];
    }
}

__global__ void k_gene_count_reduce(const uint32_t* __restrict__ part, uint64_t n_cols, uint64_t n_blocks, uint32_t* __restrict__ cnt);

template <bool WIDEN>
static int32_t launch_narrow16_tiles(srx_mat* m, int nt, int tg, hipStream_t stream) {
    srx_ctx* ctx = m->ctx;
    // the counting form where a counter per gene fits the LDS and the counts are still owed
    const bool count = m->n_cols * 4 <= 155648 && !m->cnt_pat_valid && m->n_rows > 0 && m->nnz > 0;
    const unsigned g = count ? (unsigned)std::min<uint64_t>((m->n_rows + 15) / 16, (uint64_t)ctx->n_cus * 2)
                             : (unsigned)std::min<uint64_t>((m->n_rows + 3) / 4, (uint64_t)ctx->n_cus * 32);
    uint32_t* part = nullptr;
    if (count) {
        if (!m->d_cnt_pat) SRX_HIP(ctx, dev_malloc(ctx, (void**)&m->d_cnt_pat, m->n_cols * sizeof(uint32_t)));
        SRX_TRY(scratch(ctx, stream == ctx->stream ? "n16_part_cnt" : "n16_part_cnt_side", (size_t)g * m->n_cols * sizeof(uint32_t), (void**)&part));
    }
    auto go = [&](auto nb) -> int32_t {
        constexpr int NBv = decltype(nb)::value;
        if (count) {
            const size_t lds = m->n_cols * sizeof(uint32_t);
            SRX_HIP(ctx, hipFuncSetAttribute((const void*)k_narrow16_tiles<NBv, WIDEN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_narrow16_tiles<NBv, WIDEN, true>), dim3(g ? g : 1), dim3(1024), lds, stream, m->d_indptr, m->d_indices,
                               m->n_rows, tg, m->d_idx16, m->d_tile_ptr, (uint32_t)m->n_cols, part);
        } else {
            hipLaunchKernelGGL((k_narrow16_tiles<NBv, WIDEN, false>), dim3(g ? g : 1), dim3(256), 0, stream, m->d_indptr, m->d_indices,
                               m->n_rows, tg, m->d_idx16, m->d_tile_ptr, 0u, (uint32_t*)nullptr);
        }
        return SRX_OK;
    };
    int32_t rc;
    switch (nt - 1) {
        case 0: rc = go(std::integral_constant<int, 0>{}); break;
        case 1: rc = go(std::integral_constant<int, 1>{}); break;
        case 2: rc = go(std::integral_constant<int, 2>{}); break;
        case 3: rc = go(std::integral_constant<int, 3>{}); break;
        case 4: rc = go(std::integral_constant<int, 4>{}); break;
        case 5: rc = go(std::integral_constant<int, 5>{}); break;
        case 6: rc = go(std::integral_constant<int, 6>{}); break;
        case 7: rc = go(std::integral_constant<int, 7>{}); break;
        default: rc = go(std::integral_constant<int, 8>{}); break;
    }
    SRX_TRY(rc);
    if (count) {
        hipLaunchKernelGGL(k_gene_count_reduce, dim3((unsigned)((m->n_cols + 255) / 256)), dim3(256), 0, stream, (const uint32_t*)part, m->n_cols,
                           (uint64_t)g, m->d_cnt_pat);
        m->cnt_pat_valid = true;
    }
    SRX_HIP(ctx, hipGetLastError());
    return SRX_OK;
}

// upload_on's second half for a matrix of at most 65536 columns: d_idx16 holds the column indices; d_indices and the tile cuts
// follow from it.  (m->d_indptr is in place; at most 7 tiles at 65536 columns.)
int32_t tiles_from_idx16(srx_mat* m, hipStream_t stream) {
    srx_ctx* ctx = m->ctx;
    int nt, tg;
    tile_geometry(m, nt, tg);
    if (nt > 9 || !m->d_idx16 || m->d_tile_ptr) return fail(ctx, SRX_E_ARG, "tiles_from_idx16: %d gene tiles", nt);
    if (m->n_rows == 0) return SRX_OK;
    if (nt > 1) SRX_HIP(ctx, dev_malloc(ctx, (void**)&m->d_tile_ptr, (size_t)(nt - 1) * m->n_rows * sizeof(int64_t)));
    SRX_TRY(launch_narrow16_tiles<true>(m, nt, tg, stream));
    m->n_tiles = nt;
    m->tile_genes = tg;
    return SRX_OK;
}

int32_t ensure_tiles(srx_mat* m) {
    srx_ctx* ctx = m->ctx;
    if (m->n_tiles) return SRX_OK;
    int nt, tg;
    tile_geometry(m, nt, tg);
    if (nt > 1 && nt <= 9 && m->n_cols <= 65536 && !m->d_idx16 && !m->d_tile_ptr && m->n_rows > 0) {
        SRX_HIP(ctx, dev_malloc(ctx, (void**)&m->d_tile_ptr, (size_t)(nt - 1) * m->n_rows * sizeof(int64_t)));
        SRX_HIP(ctx, dev_malloc(ctx, (void**)&m->d_idx16, (m->nnz + 16) * sizeof(uint16_t)));
        SRX_TRY(launch_narrow16_tiles<false>(m, nt, tg, ctx->stream));
        m->n_tiles = nt;
        m->tile_genes = tg;
        return SRX_OK;
    }
    if (nt > 1) {
        SRX_HIP(ctx, dev_malloc(ctx, (void**)&m->d_tile_ptr, (size_t)(nt - 1) * (m->n_rows ? m->n_rows : 1) * sizeof(int64_t)));
        SRX_TRY(launch_tile_ptr(ctx, m->d_indptr, m->d_indices, m->n_rows, nt, tg, m->d_tile_ptr));
    }
    if (m->n_cols <= 65536 && !m->d_idx16) {
        SRX_HIP(ctx, dev_malloc(ctx, (void**)&m->d_idx16, (m->nnz + 16) * sizeof(uint16_t)));
        uint64_t g = (m->nnz + 16 + 1023) / 1024;
        if (g < 1) g = 1;
        if (g > 65535) g = 65535;
        hipLaunchKernelGGL(k_narrow16, dim3((unsigned)g), dim3(256), 0, ctx->stream, m->d_indices, m->nnz, m->nnz + 16,
                           m->d_idx16);
        SRX_HIP(ctx, hipGetLastError());
    }
    m->n_tiles = nt;
    m->tile_genes = tg;
    return SRX_OK;
}

static int32_t block_geometry(const srx_mat* m, uint64_t& n_blocks, uint64_t& rows_per_block) {
    // four workgroups per CU, one resident at a time (the accumulators fill the LDS): c3, XF + write-back pass, 1 / 2 / 4 / 8 / 16
    // per CU: 3.23 / 3.12 / 3.06 / 3.10 / 3.19 ms (shorter workgroups even out the CUs; more of them means more partial sums)
    uint64_t want = (uint64_t)(4 * m->ctx->n_cus) / (uint64_t)m->n_tiles;
    if (want < 1) want = 1;
    uint64_t by_rows = (m->n_rows + 63) / 64;   // at least ~64 rows per block
    if (by_rows < 1) by_rows = 1;
    n_blocks = want < by_rows ? want : by_rows;
    rows_per_block = (m->n_rows + n_blocks - 1) / n_blocks;
    if (rows_per_block < 1) rows_per_block = 1;
    // the moments pass addresses a row block's entries with 32-bit offsets from the block's first entry (a row holds at most
    // n_cols entries)
    const uint64_t cap = std::max<uint64_t>(1, 0x7fffffffull / std::max<uint64_t>(m->n_cols, 1));
    if (rows_per_block > cap) {
        rows_per_block = cap;
        n_blocks = (m->n_rows + rows_per_block - 1) / rows_per_block;
        // every block leaves a partial (sum, sumsq) pair per gene: a matrix so wide that the 32-bit cap asks for thousands of
        // blocks would make those buffers larger than the matrix — refused, not attempted
        if ((double)n_blocks * (double)std::max<uint64_t>(m->n_cols, 1) * 16.0 > 8.0 * 1024 * 1024 * 1024)
            return fail(m->ctx, SRX_E_ARG, "gene statistics: %llu columns x %llu row blocks exceed the partial-sum budget (matrix too wide)",
                        (unsigned long long)m->n_cols, (unsigned long long)n_blocks);
    }
    return SRX_OK;
}

// This shard's (cnt, sum, sumsq) per gene and its row count, packed as 3G+1 doubles in a scratch buffer.
// `xf`: moments of the transformed values ln_1p(v * scale_row) formed on the fly from the raw matrix.
static int32_t local_moments(srx_mat* m, double** packed_out, RowXf xf = RowXf{}) {
    srx_ctx* ctx = m->ctx;
    SRX_TRY(ensure_tiles(m));
    const uint64_t G = m->n_cols;
    uint64_t nb, rpb;
    SRX_TRY(block_geometry(m, nb, rpb));
    double *p_sum, *p_sq, *packed;
    SRX_TRY(scratch(ctx, "mom_part_sum", nb * (G ? G : 1) * sizeof(double), (void**)&p_sum));
    SRX_TRY(scratch(ctx, "mom_part_sq", nb * (G ? G : 1) * sizeof(double), (void**)&p_sq));
    SRX_TRY(scratch(ctx, "mom_packed", (3 * G + 1) * sizeof(double), (void**)&packed));
    // the per-gene counts depend on the sparsity pattern only: counted once per pattern (here, or by srx_matrix_prepare), kept
    // on the matrix (clones inherit them)
    SRX_TRY(ensure_pattern_counts(m));
    // SRX_MOM_BLOCKED=1: the accumulators in blocks of 32 genes (no half-bank restriction; measured slower: 2.81 against 2.75 ms,
    // profiles/r06_knockouts.md); default: the {sum, sum of squares} pairs
    const bool blocked = xf.row_sum && getenv("SRX_MOM_BLOCKED") && atoi(getenv("SRX_MOM_BLOCKED"));
    const size_t lds = (size_t)(blocked ? (m->tile_genes + 31) & ~31 : m->tile_genes) * 16 + (xf.row_sum ? (size_t)kLog1pTabBytes : 0);
    // s_i = 2 when the 16-bit index mirror exists (n_cols <= 65536), 4 otherwise
    const double bytes = (double)m->nnz * ((m->n_cols <= 65536 ? 2.0 : 4.0) + val_bytes(m)) + (double)(m->n_rows + 1) * 8.0 +
                         (double)G * 24.0 + (xf.row_sum ? (double)m->n_rows * 8.0 : 0.0) +
                         (xf.write_back ? (double)m->nnz * val_bytes(m) : 0.0);
    // fixed-point scales of the transformed sums: 62 bits hold n_rows values of magnitude < 2^6 (ln_1p(x) < 64, any
    // x < 6e27) resp. their squares
    double fx_sum = 0.0, fx_sq = 0.0;
    if (xf.row_sum) {
        int lg = 0;
        while ((1ull << lg) < (m->n_rows ? m->n_rows : 1)) ++lg;
        fx_sum = std::ldexp(1.0, std::min(62 - 6 - lg, 44));
        fx_sq = std::ldexp(1.0, std::min(62 - 12 - lg, 38));
    }
    uint32_t* d_poison = nullptr;
    if (xf.row_sum) {
        SRX_TRY(scratch(ctx, "mom_poison", (G ? G : 1) * sizeof(uint32_t), (void**)&d_poison));
        SRX_HIP(ctx, hipMemsetAsync(d_poison, 0, (G ? G : 1) * sizeof(uint32_t), ctx->stream));
    }
    {
        ProfScope ps(ctx, SRX_K_MOMENTS, bytes);
        dim3 grid((unsigned)(nb * m->n_tiles));
        auto launch = [&](auto kern, const auto* idxp, auto* valp) -> int32_t {
            SRX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, grid, dim3(kMomThreads), lds, ctx->stream, m->d_indptr, m->d_tile_ptr, idxp, valp,
                               m->n_rows, G, m->n_tiles, m->tile_genes, rpb, xf.row_sum, xf.target, fx_sum, fx_sq, d_poison, p_sum, p_sq);
            return SRX_OK;
        };
        auto pick = [&](auto tval, auto tidx, const auto* idxp, const auto* valp) -> int32_t {
            using T = decltype(tval);
            using I = decltype(tidx);
            if (xf.row_sum && xf.write_back && blocked) return launch(k_gene_moments<T, I, true, true, true>, idxp, const_cast<T*>(valp));
            if (xf.row_sum && xf.write_back) return launch(k_gene_moments<T, I, true, true>, idxp, const_cast<T*>(valp));
            if (xf.row_sum && blocked) return launch(k_gene_moments<T, I, true, false, true>, idxp, valp);
            if (xf.row_sum) return launch(k_gene_moments<T, I, true>, idxp, valp);
            return launch(k_gene_moments<T, I, false>, idxp, valp);
        };
        if (is_f32(m)) {
            if (m->d_idx16) SRX_TRY(pick(float{}, uint16_t{}, (const uint16_t*)m->d_idx16, (const float*)m->d_values));
            else SRX_TRY(pick(float{}, int32_t{}, (const int32_t*)m->d_indices, (const float*)m->d_values));
        } else {
            if (m->d_idx16) SRX_TRY(pick(double{}, uint16_t{}, (const uint16_t*)m->d_idx16, (const double*)m->d_values));
            else SRX_TRY(pick(double{}, int32_t{}, (const int32_t*)m->d_indices, (const double*)m->d_values));
        }
        hipLaunchKernelGGL(k_moments_reduce, dim3((unsigned)((G + 63) / 64 ? (G + 63) / 64 : 1)), dim3(64 * kRedSlices), 0, ctx->stream, p_sum, p_sq, G, nb,
                           m->n_rows, (const uint32_t*)m->d_cnt_pat, fx_sum != 0.0 ? 1.0 / fx_sum : 0.0,
                           fx_sq != 0.0 ? 1.0 / fx_sq : 0.0, (const uint32_t*)d_poison, packed);
    }
    SRX_HIP(ctx, hipGetLastError());
    *packed_out = packed;
    return SRX_OK;
}

// The pattern-only per-gene counts (k_gene_count): one pass over the column indices, once per pattern — srx_matrix_prepare
// makes them ahead of time, so that clones inherit them.
int32_t ensure_pattern_counts(srx_mat* m) {
    if (m->cnt_pat_valid && m->d_cnt_pat) return SRX_OK;
    srx_ctx* ctx = m->ctx;
    SRX_TRY(ensure_tiles(m));
    const uint64_t G = m->n_cols;
    if (!m->d_cnt_pat) SRX_HIP(ctx, dev_malloc(ctx, (void**)&m->d_cnt_pat, (G ? G : 1) * sizeof(uint32_t)));
    SRX_HIP(ctx, hipMemsetAsync(m->d_cnt_pat, 0, (G ? G : 1) * sizeof(uint32_t), ctx->stream));
    if (m->n_rows && m->nnz) {
        uint64_t nb, rpb;
        SRX_TRY(block_geometry(m, nb, rpb));
        uint32_t* part;
        SRX_TRY(scratch(ctx, "mom_part_cnt", nb * G * sizeof(uint32_t), (void**)&part));
        const dim3 grid((unsigned)(nb * m->n_tiles));
        const size_t lds = (size_t)m->tile_genes * 4;
        ProfScope ps(ctx, SRX_K_MOMENTS, (double)m->nnz * (m->d_idx16 ? 2.0 : 4.0) + (double)(m->n_rows + 1) * 8.0 + (double)G * 4.0);
        if (m->d_idx16)
            hipLaunchKernelGGL(k_gene_count<uint16_t>, grid, dim3(kMomThreads), lds, ctx->stream, m->d_indptr, m->d_tile_ptr,
                               (const uint16_t*)m->d_idx16, m->n_rows, G, m->n_tiles, m->tile_genes, rpb, part);
        else
            hipLaunchKernelGGL(k_gene_count<int32_t>, grid, dim3(kMomThreads), lds, ctx->stream, m->d_indptr, m->d_tile_ptr,
                               (const int32_t*)m->d_indices, m->n_rows, G, m->n_tiles, m->tile_genes, rpb, part);
        hipLaunchKernelGGL(k_gene_count_reduce, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)part, G, nb,
                           m->d_cnt_pat);
        SRX_HIP(ctx, hipGetLastError());
    }
    m->cnt_pat_valid = true;
    return SRX_OK;
}

__global__ void k_add_f64(double* __restrict__ acc, const double* __restrict__ x, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += x[i];
}

// Backed mode: the moments of one row tile ADDED to `d_acc` (3G+1 doubles; counts stay exact below 2^53).
int32_t moments_accumulate(srx_mat* m, double* d_acc, RowXf xf) {
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    double* packed;
    SRX_TRY(local_moments(m, &packed, xf));
    const uint64_t n = 3 * m->n_cols + 1;
    hipLaunchKernelGGL(k_add_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_acc, packed, n);
    SRX_HIP(ctx, hipGetLastError());
    return SRX_OK;
}

static int32_t alloc_moments(srx_mat* m) {
    srx_ctx* ctx = m->ctx;
    const uint64_t G = m->n_cols;
    if (!m->d_cnt) {
        SRX_HIP(ctx, hipMalloc((void**)&m->d_cnt, (G ? G : 1) * sizeof(uint64_t)));
        SRX_HIP(ctx, hipMalloc((void**)&m->d_sum, (G ? G : 1) * sizeof(double)));
        SRX_HIP(ctx, hipMalloc((void**)&m->d_sq, (G ? G : 1) * sizeof(double)));
    }
    return SRX_OK;
}

// All-reduces `d_packed` (3G+1 doubles, this rank's totals) and installs it as the global moments of `m`.
int32_t moments_install(srx_mat* m, double* d_packed) {
    srx_ctx* ctx = m->ctx;
    const uint64_t G = m->n_cols;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    SRX_TRY(alloc_moments(m));
    SRX_TRY(allreduce_f64(ctx, d_packed, 3 * G + 1));
    hipLaunchKernelGGL(k_moments_unpack, dim3((unsigned)((G + 255) / 256 + 1)), dim3(256), 0, ctx->stream, d_packed, G,
                       m->d_cnt, m->d_sum, m->d_sq);
    SRX_HIP(ctx, hipGetLastError());
    double ng = 0.0;
    SRX_TRY(d2h(ctx, &ng, d_packed + 3 * G, sizeof(double)));
    m->n_rows_global = (uint64_t)ng;
    m->moments_version = m->version;
    return SRX_OK;
}

int32_t ensure_moments(srx_mat* m) {
    srx_ctx* ctx = m->ctx;
    if (m->moments_version == m->version) return SRX_OK;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t G = m->n_cols;
    SRX_TRY(alloc_moments(m));
    double* packed;
    SRX_TRY(local_moments(m, &packed));
    // one all-reduce for (cnt, sum, sumsq, N) across the row shards
    SRX_TRY(allreduce_f64(ctx, packed, 3 * G + 1));
    hipLaunchKernelGGL(k_moments_unpack, dim3((unsigned)((G + 255) / 256 + 1)), dim3(256), 0, ctx->stream, packed, G,
                       m->d_cnt, m->d_sum, m->d_sq);
    SRX_HIP(ctx, hipGetLastError());
    if (ctx->comm || ctx->host_allreduce) {
        double ng = 0.0;
        SRX_TRY(d2h(ctx, &ng, packed + 3 * G, sizeof(double)));
        m->n_rows_global = (uint64_t)ng;
    } else {
        m->n_rows_global = m->n_rows;
    }
    m->moments_version = m->version;
    return SRX_OK;
}

// Moments of the transformed values y = ln_1p(v * scale_row), computed from the RAW matrix (pipeline): installed as the
// moments of the version the in-place write-back is about to create (the caller bumps the version right after).
int32_t ensure_moments_xf(srx_mat* m, RowXf xf) {
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t G = m->n_cols;
    SRX_TRY(alloc_moments(m));
    double* packed;
    SRX_TRY(local_moments(m, &packed, xf));
    if (xf.write_back) {
        // the pass above has stored the transformed values in place: whatever fails from here on (all-reduce, read-back), X
        // must not be transformed a second time by the pipeline's epilogue — the bookkeeping of normalize_total + log1p
        // (scale/mod.rs:74-83 -> F64, transform/mod.rs:43-55) belongs to the moment the kernel is in the stream
        m->lazy_pending = false;
        m->dtype = SRX_F64;
        touch(m);
    }
    SRX_TRY(allreduce_f64(ctx, packed, 3 * G + 1));
    hipLaunchKernelGGL(k_moments_unpack, dim3((unsigned)((G + 255) / 256 + 1)), dim3(256), 0, ctx->stream, packed, G,
                       m->d_cnt, m->d_sum, m->d_sq);
    SRX_HIP(ctx, hipGetLastError());
    if (ctx->comm || ctx->host_allreduce) {
        double ng = 0.0;
        SRX_TRY(d2h(ctx, &ng, packed + 3 * G, sizeof(double)));
        m->n_rows_global = (uint64_t)ng;
    } else {
        m->n_rows_global = m->n_rows;
    }
    return SRX_OK;
}

static int32_t fetch_moments(srx_mat* m, std::vector<uint64_t>& cnt, std::vector<double>& sum, std::vector<double>& sq) {
    SRX_TRY(ensure_moments(m));
    const uint64_t G = m->n_cols;
    cnt.resize(G); sum.resize(G); sq.resize(G);
    SRX_TRY(d2h(m->ctx, cnt.data(), m->d_cnt, G * sizeof(uint64_t)));
    SRX_TRY(d2h(m->ctx, sum.data(), m->d_sum, G * sizeof(double)));
    SRX_TRY(d2h(m->ctx, sq.data(), m->d_sq, G * sizeof(double)));
    return SRX_OK;
}

// csr.rs:179-185, evaluated in the reference's operation order (no FMA contraction).
#pragma clang fp contract(off)
void finalize_variance(const uint64_t* cnt, const double* sum, const double* sq, uint64_t G, double* out) {
    for (uint64_t j = 0; j < G; ++j) {
        out[j] = 0.0;
        if (cnt[j] > 0) {
            double c = (double)(uint32_t)cnt[j];
            double mean = sum[j] / c;
            out[j] = sq[j] / c - mean * mean;
        }
    }
}

int32_t gene_variances(srx_mat* m, std::vector<double>& var) {
    std::vector<uint64_t> cnt;
    std::vector<double> sum, sq;
    SRX_TRY(fetch_moments(m, cnt, sum, sq));
    var.resize(m->n_cols);
    finalize_variance(cnt.data(), sum.data(), sq.data(), m->n_cols, var.data());
    return SRX_OK;
}

// dim_red/mod.rs:135-140.  Rust's sort_by is stable; comparator b.partial_cmp(a) = descending.
int32_t select_hvg_host(srx_ctx* ctx, const std::vector<double>& var, uint64_t n, std::vector<uint64_t>& out) {
    for (double v : var)
        if (v != v) return fail(ctx, SRX_E_NAN, "NaN gene variance: called `Option::unwrap()` on a `None` value (partial_cmp)");
    // stable descending order == the strict total order (variance desc, gene index asc), so a
    // partial sort of the first n under that order gives exactly the stable sort's prefix
    std::vector<uint64_t> order(var.size());
    for (uint64_t j = 0; j < order.size(); ++j) order[j] = j;
    uint64_t take = n < order.size() ? n : order.size();
    auto before = [&](uint64_t a, uint64_t b) { return var[a] > var[b] || (var[a] == var[b] && a < b); };
    std::nth_element(order.begin(), order.begin() + (take ? take - 1 : 0), order.end(), before);
    std::sort(order.begin(), order.begin() + take, before);
    out.assign(order.begin(), order.begin() + take);
    return SRX_OK;
}

// ---- FeatureSelection::HighlyVariable(n) on the device (the pipeline's route: no host round trip) ----
// Same semantics as select_hvg_host / dim_red/mod.rs:135-140: nz-only population variance in the reference's
// operation order (csr.rs:179-185: no FMA contraction — explicit _rn intrinsics), stable descending order
// (ties keep the ascending gene index), NaN -> status bit 1 (the reference panics).
__global__ void k_gene_var(const uint64_t* __restrict__ cnt, const double* __restrict__ sum, const double* __restrict__ sq,
                           uint64_t G, double* __restrict__ var, uint32_t* __restrict__ rank, uint32_t rank_init) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= G) return;
    double v = 0.0;
    if (cnt[j] > 0) {
        const double c = (double)(uint32_t)cnt[j];
        const double mean = __ddiv_rn(sum[j], c);
        v = __dsub_rn(__ddiv_rn(sq[j], c), __dmul_rn(mean, mean));
    }
    var[j] = v;
    rank[j] = rank_init;          // (candidate route: 0xffffffff = not a candidate; full ranking: the partial ranks start at 0)
}

// rank of every gene under (variance desc, index asc) by counting.  Block (x, y): genes 256 x .. 256 x + 255
// against the comparison slice y (kRankTile variances staged in LDS, read as broadcast b128 pairs); a tile
// wholly before the block's genes wins ties (>=), wholly after loses them (>), only the tile holding the
// block's own genes needs the index compare.  Partial ranks are summed with integer atomics (exact, any order).
constexpr int kRankTile = 2048;
// one (block of 256 genes, comparison tile) pair of the full ranking; `tile`: kRankTile doubles of LDS; all 256 threads
__device__ __forceinline__ void full_rank_pair(const double* __restrict__ var, uint32_t G, uint32_t* __restrict__ rank_out, uint32_t bx,
                                               uint32_t by, double* tile) {
    const uint32_t j = bx * 256 + threadIdx.x;
    const uint32_t base = by * kRankTile;
    for (uint32_t e = threadIdx.x; e < kRankTile; e += 256) tile[e] = base + e < G ? var[base + e] : -INFINITY;
    __syncthreads();
    if (j < G) {
        const double mine = var[j];
        const uint32_t j_lo = bx * 256, j_hi = j_lo + 255;
        uint32_t rank = 0;
        if (base + kRankTile - 1 < j_lo) {                       // every gene of the tile precedes every gene of the block
#pragma unroll 8
            for (int e = 0; e < kRankTile; ++e) rank += tile[e] >= mine ? 1u : 0u;
        } else if (base > j_hi) {
#pragma unroll 8
            for (int e = 0; e < kRankTile; ++e) rank += tile[e] > mine ? 1u : 0u;
        } else {
#pragma unroll 4
            for (int e = 0; e < kRankTile; ++e) {
                const double o = tile[e];
                rank += (o > mine || (o == mine && base + e < j)) ? 1u : 0u;
            }
        }
        atomicAdd(&rank_out[j], rank);
    }
    __syncthreads();                                             // (the tile is re-filled by the caller's next pair)
}
__global__ __launch_bounds__(256) void k_hvg_rank(const double* __restrict__ var, uint32_t G, uint32_t* __restrict__ rank_out) {
    __shared__ double tile[kRankTile];
    full_rank_pair(var, G, rank_out, blockIdx.x, blockIdx.y, tile);
}

// ---- the same ranks for the genes that can be selected only --------------------------------------------------------------
// HighlyVariable(n) needs the ranks below n.  A threshold taken from a SAMPLE of the variances (512 of them, the sample
// quantile n / G pushed down by four standard deviations of a sample quantile) leaves ~n + 10 % of G candidates; they are
// ranked among themselves, every gene below the threshold keeps rank 0xffffffff.  If the sample misled (fewer than n
// candidates: probability ~3e-5 per call; more than `cap`, the size the candidate kernel is launched for; NaN variances)
// the full ranking runs instead — decided on the device.
// O(M^2) instead of O(G^2) comparisons: 173 -> ~25 us at 28k genes, n = 2000.
constexpr int kRankSample = 512;
constexpr int kSelThreads = 1024;
// One workgroup: the threshold from the sample, the candidate list (gene ids in any order: the ranks are a strict total order),
// rank_out[candidate] = 0, and the decision: `counter` = number of candidates M if n <= M <= cap, else 0xffffffff with
// rank_out zeroed for EVERY gene — the full ranking takes over in k_rank_selected.  (Round 5: threshold, gather, the fallback's
// reset and its early-exit launch were four kernels of 5-17 us each on the step's critical path.)
__global__ __launch_bounds__(kSelThreads) void k_hvg_candidates(const double* __restrict__ var, uint32_t G, uint32_t n, uint32_t cap,
                                                                uint32_t* __restrict__ counter, uint32_t* __restrict__ cand,
                                                                uint32_t* __restrict__ rank_out,
                                                                int force_miss /* test switch: a threshold nothing reaches */) {
    __shared__ double sv[kRankSample];
    __shared__ double s_thr;
    __shared__ uint32_t s_cnt;
    const uint32_t S = G < (uint32_t)kRankSample ? G : (uint32_t)kRankSample;
    const uint32_t t = threadIdx.x, lane = t & 63;
    if (t < (uint32_t)kRankSample) sv[t] = t < S ? var[(uint64_t)t * G / S] : -INFINITY;
    if (t == 0) {
        s_thr = force_miss ? INFINITY : -INFINITY;      // every gene a candidate unless a sample element says otherwise
        s_cnt = 0u;
    }
    __syncthreads();
    if (!force_miss && S == (uint32_t)kRankSample) {          // (uniform)
        const double q = (double)n / (double)G;
        const double kq = (q + 4.0 * sqrt(q * (1.0 - q) / (double)S)) * (double)S + 1.0;
        // the sample in descending order (bitonic network in LDS, 256 compare-exchanges per step: ranking every element against
        // every other one took 17 us of one workgroup); equal values may stand in any order — the threshold is a VALUE
        for (uint32_t size = 2; size <= (uint32_t)kRankSample; size <<= 1)
            for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                if (t < (uint32_t)kRankSample / 2) {
                    const uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                    const bool desc = (lo & size) == 0;
                    const double a = sv[lo], b = sv[hi];
                    if (desc ? a < b : a > b) {
                        sv[lo] = b;
                        sv[hi] = a;
                    }
                }
                __syncthreads();
            }
        if (t == 0 && kq < (double)(S - 1)) s_thr = sv[(uint32_t)kq];       // (NaN variances in the sample: whatever stands there — the status word reports them)
    }
    __syncthreads();
    const double thr = s_thr;
    constexpr int kAhead = 8;                 // variances of eight rounds in flight (a round per load was 0.5 us of latency each)
    for (uint32_t base = 0; base < G; base += kSelThreads * kAhead) {
        double v[kAhead];
#pragma unroll
        for (int a = 0; a < kAhead; ++a) {
            const uint32_t j = base + a * kSelThreads + t;
            v[a] = j < G ? var[j] : -INFINITY;
        }
#pragma unroll
        for (int a = 0; a < kAhead; ++a) {
            const uint32_t j = base + a * kSelThreads + t;
            const bool in = j < G && v[a] >= thr;
            const unsigned long long m = __ballot(in);
            uint32_t at = 0;
            if (lane == 0 && m) at = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
            at = __shfl(at, 0, 64);
            if (in) {
                cand[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = j;
                rank_out[j] = 0u;
            }
        }
    }
    __syncthreads();
    const uint32_t M = s_cnt;
    const bool ok = M >= n && M <= cap;
    if (!ok)
        for (uint32_t j = t; j < G; j += kSelThreads) rank_out[j] = 0u;          // the full ranking's partial sums start from zero
    if (t == 0) *counter = ok ? M : 0xffffffffu;
}
// rank_out[gene] for every candidate: candidates above it under (variance desc, index asc).  Pair (x, y): candidates 256 x ..
// against the y-th tile of kCandTile of the candidate list; partial ranks are summed with integer atomics.  Where the
// candidates cannot be used (`counter` = 0xffffffff: fewer than n, more than the list the launch was sized for, NaN variances)
// the same workgroups walk the pairs of the FULL ranking instead.
constexpr int kCandTile = 128;      // (512: 44 us per launch at c3 — each thread walks a whole tile; 128: four times the workgroups, a quarter of the walk)
__global__ __launch_bounds__(256) void k_rank_selected(const double* __restrict__ var, uint32_t G, const uint32_t* __restrict__ counter,
                                                       const uint32_t* __restrict__ cand, uint32_t* __restrict__ rank_out) {
    __shared__ double tile[kRankTile];
    const uint32_t M = *counter;
    if (M == 0xffffffffu) {
        const uint32_t gb = (G + 255) / 256, gy = (G + kRankTile - 1) / kRankTile;
        for (uint32_t p = blockIdx.y * gridDim.x + blockIdx.x; p < gb * gy; p += gridDim.x * gridDim.y)
            full_rank_pair(var, G, rank_out, p % gb, p / gb, tile);
        return;
    }
    double* tv = tile;
    uint32_t* ti = reinterpret_cast<uint32_t*>(tile + kCandTile);
    const uint32_t base = blockIdx.y * kCandTile;
    if (blockIdx.x * 256 >= M || base >= M) return;           // (uniform)
    for (uint32_t e = threadIdx.x; e < kCandTile; e += 256) {
        const uint32_t g = base + e < M ? cand[base + e] : 0xffffffffu;
        ti[e] = g;
        tv[e] = g != 0xffffffffu ? var[g] : -INFINITY;
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const uint32_t gi = cand[i];
    const double mine = var[gi];
    uint32_t rank = 0;
#pragma unroll 4
    for (int e = 0; e < kCandTile; ++e) {
        const double o = tv[e];
        rank += (o > mine || (o == mine && ti[e] < gi)) ? 1u : 0u;
    }
    if (rank) atomicAdd(&rank_out[gi], rank);
}

// One workgroup: selection bitmask + per-word prefix counts (what the compaction kernels stage in LDS), and
// the per-slot (ascending gene order) centring / scaling vectors of the PCA from the all-cells moments
// (pca/mod.rs:87-91: mean = sum/N, var = sumsq/N - mean^2, ddof 0); trace = sum_s dinv_s^2 ss_s, summed in a
// fixed tree.  n_words <= 2048 (G <= 65536).
__global__ __launch_bounds__(1024) void k_sel_finish(const uint32_t* __restrict__ rank, const double* __restrict__ gene_var,
                                                     const double* __restrict__ sum,
                                                     const double* __restrict__ sq, uint32_t G, uint32_t take, int n_words,
                                                     double n_cells, int center, int scale, int32_t* __restrict__ sel_rank,
                                                     int* __restrict__ status, uint32_t* __restrict__ bits,
                                                     uint32_t* __restrict__ prefix, double* __restrict__ mu,
                                                     double* __restrict__ sd, double* __restrict__ dinv,
                                                     double* __restrict__ trace) {
    __shared__ uint32_t s_bits[2048], s_pre[2048], s_wave[16];
    __shared__ double s_tr[1024];
    __shared__ int s_nan;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) s_nan = 0;
    __syncthreads();
    // the gene of rank r < take goes to sel_rank[r] and into the mask (two words per thread); a NaN variance anywhere is the
    // status word's bit 0 (the reference panics there: select_hvg_host)
    bool nan = false;
    constexpr int kAhead = 8;                 // (the loads of eight rounds in flight)
    for (uint32_t base = 0; base < (uint32_t)n_words * 32; base += 1024 * kAhead) {       // (a wave: 64 consecutive genes = two mask words)
        uint32_t rk[kAhead];
        double v[kAhead];
#pragma unroll
        for (int a = 0; a < kAhead; ++a) {
            const uint32_t g = base + a * 1024 + t;
            rk[a] = g < G ? rank[g] : 0xffffffffu;
            v[a] = g < G ? gene_var[g] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < kAhead; ++a) {
            const uint32_t g = base + a * 1024 + t;
            nan |= v[a] != v[a];
            const bool in = rk[a] < take;      // (a gene behind G: 0xffffffff, never below `take`)
            if (in) sel_rank[rk[a]] = (int32_t)g;
            const unsigned long long m = __ballot(in);
            if (lane == 0 && (g >> 5) < (uint32_t)n_words) s_bits[g >> 5] = (uint32_t)m;
            if (lane == 32 && (g >> 5) < (uint32_t)n_words) s_bits[g >> 5] = (uint32_t)(m >> 32);
        }
    }
    if (nan) s_nan = 1;
    __syncthreads();
    uint32_t b[2] = {0u, 0u};
    if (2 * t < n_words) b[0] = s_bits[2 * t];
    if (2 * t + 1 < n_words) b[1] = s_bits[2 * t + 1];
    // exclusive scan of the popcounts: per-thread pair, wave scan, wave totals
    const uint32_t c0 = (uint32_t)__popc(b[0]), c1 = (uint32_t)__popc(b[1]);
    uint32_t inc = c0 + c1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wv; ++w) before += s_wave[w];
    const uint32_t excl = before + inc - (c0 + c1);
    if (2 * t < n_words) s_pre[2 * t] = excl;
    if (2 * t + 1 < n_words) s_pre[2 * t + 1] = excl + c0;
    __syncthreads();
    // never more than `take` selected genes, whatever the flags say (a ranking broken by NaN variances can flag more): the
    // passes over the matrix test the bit alone and the compacted columns must stay below `take`
    for (int w = t; w < n_words; w += 1024) {
        uint32_t b_ = s_bits[w];
        const uint32_t e = s_pre[w], room = e < take ? take - e : 0u;
        while ((uint32_t)__popc(b_) > room) b_ &= ~(1u << (31 - __clz((int)b_)));
        s_bits[w] = b_;
    }
    __syncthreads();
    for (int w = t; w < n_words; w += 1024) {
        bits[w] = s_bits[w];
        prefix[w] = s_pre[w];
    }
    double tr = 0.0;
    for (uint32_t g = t; g < G; g += 1024) {
        const uint32_t w = s_bits[g >> 5], bit = 1u << (g & 31);
        if (!(w & bit)) continue;
        const uint32_t s = s_pre[g >> 5] + (uint32_t)__popc(w & (bit - 1u));
        const double mean = sum[g] / n_cells;
        double var = sq[g] / n_cells - mean * mean;
        if (var < 0) var = 0;
        const double std_ = sqrt(var);
        mu[s] = (center || scale) ? mean : 0.0;          // pca/mod.rs:85-119: stored only if center || scale
        sd[s] = scale ? std_ : 1.0;
        const double di = (scale && std_ > 0) ? 1.0 / std_ : 1.0;      // zero-variance column: std treated as 1
        dinv[s] = di;
        double ss = center ? (sq[g] - n_cells * mean * mean) : sq[g];
        if (ss < 0) ss = 0;
        tr += di * di * ss;
    }
    s_tr[t] = tr;
    __syncthreads();
    for (int half = 512; half > 0; half >>= 1) {
        if (t < half) s_tr[t] += s_tr[t + half];
        __syncthreads();
    }
    if (t == 0) {
        *trace = s_tr[0];
        *status = s_nan;
    }
}

// Device-side HighlyVariable(n): fills the scratch buffers named below; nothing is read back.
int32_t select_hvg_device(srx_mat* m, uint64_t n, int center, int scale, HvgDev& out) {
    srx_ctx* ctx = m->ctx;
    SRX_TRY(ensure_moments(m));
    const uint64_t G = m->n_cols;
    const uint64_t take = n < G ? n : G;
    const int n_words = (int)((G + 31) / 32);
    double *d_var, *d_f;
    uint32_t* d_rank;
    SRX_TRY(scratch(ctx, "hvg_var", (G ? G : 1) * sizeof(double), (void**)&d_var));
    SRX_TRY(scratch(ctx, "hvg_rankv", (G ? G : 1) * sizeof(uint32_t), (void**)&d_rank));
    SRX_TRY(scratch(ctx, "hvg_rank", (take ? take : 1) * sizeof(int32_t), (void**)&out.d_sel_rank));
    SRX_TRY(scratch(ctx, "pca_selbits", (size_t)(2 * n_words ? 2 * n_words : 1) * sizeof(uint32_t), (void**)&out.d_bits));
    SRX_TRY(scratch(ctx, "hvg_f", (3 * (take ? take : 1) + 8) * sizeof(double), (void**)&d_f));
    SRX_TRY(scratch(ctx, "hvg_status", 256, (void**)&out.d_status));
    out.d_mu = d_f;
    out.d_sd = d_f + take;
    out.d_dinv = d_f + 2 * take;
    out.d_tr = nullptr;
    out.d_trace = d_f + 3 * take;
    out.k = (int)take;
    out.n_words = n_words;
    const unsigned gb = (unsigned)((G + 255) / 256 ? (G + 255) / 256 : 1);
    const unsigned gy = (unsigned)((G + kRankTile - 1) / kRankTile ? (G + kRankTile - 1) / kRankTile : 1);
    static const bool full_rank = getenv("SRX_HVG_FULL_RANK") != nullptr;       // A/B switch: rank every gene
    const bool all = full_rank || take == 0 || G < (uint64_t)kRankSample;
    // four launches (round 4: eleven): variances; threshold + candidate list + the decision; the ranks; mask, vectors, status
    hipLaunchKernelGGL(k_gene_var, dim3(gb), dim3(256), 0, ctx->stream, m->d_cnt, m->d_sum, m->d_sq, G, d_var, d_rank, all ? 0u : 0xffffffffu);
    if (all) {
        hipLaunchKernelGGL(k_hvg_rank, dim3(gb, gy), dim3(256), 0, ctx->stream, d_var, (uint32_t)G, d_rank);
    } else {
        uint32_t* d_cand;
        SRX_TRY(scratch(ctx, "hvg_cand", ((G ? G : 1) + 8) * sizeof(uint32_t), (void**)&d_cand));
        uint32_t* d_counter = d_cand + G + 4;
        static const int force_miss = getenv("SRX_HVG_FORCE_MISS") ? 1 : 0;       // exercises the device-side fallback
        const uint32_t cap = (uint32_t)std::min<uint64_t>(G, std::max<uint64_t>(4 * take, 4096));
        hipLaunchKernelGGL(k_hvg_candidates, dim3(1), dim3(kSelThreads), 0, ctx->stream, d_var, (uint32_t)G, (uint32_t)take, cap, d_counter, d_cand,
                           d_rank, force_miss);
        hipLaunchKernelGGL(k_rank_selected, dim3((cap + 255) / 256, (cap + kCandTile - 1) / kCandTile), dim3(256), 0, ctx->stream, d_var,
                           (uint32_t)G, d_counter, d_cand, d_rank);
    }
    hipLaunchKernelGGL(k_sel_finish, dim3(1), dim3(1024), 0, ctx->stream, d_rank, d_var, m->d_sum, m->d_sq, (uint32_t)G, (uint32_t)take, n_words,
                       (double)m->n_rows_global, center, scale, out.d_sel_rank, out.d_status, out.d_bits, out.d_bits + n_words, out.d_mu,
                       out.d_sd, out.d_dinv, out.d_trace);
    SRX_HIP(ctx, hipGetLastError());
    return SRX_OK;
}

int32_t row_number(srx_mat* m, uint32_t* out);
int32_t row_qc(srx_mat* m, uint32_t* num, double* sum, double* var);
int32_t row_stat(srx_mat* m, int which, double* out0, double* out1);

}  // namespace srx

using namespace srx;

extern "C" {

// per-column moments of the STORED matrix (for a CSC matrix: per cell)
static int32_t stored_col_moments(srx_mat* m, uint64_t* cnt, double* sum, double* sumsq) {
    SRX_TRY(ensure_moments(m));
    const uint64_t G = m->n_cols;
    if (cnt) SRX_TRY(d2h(m->ctx, cnt, m->d_cnt, G * sizeof(uint64_t)));
    if (sum) SRX_TRY(d2h(m->ctx, sum, m->d_sum, G * sizeof(double)));
    if (sumsq) SRX_TRY(d2h(m->ctx, sumsq, m->d_sq, G * sizeof(double)));
    return SRX_OK;
}

int32_t srx_gene_moments(srx_mat* m, uint64_t* cnt, double* sum, double* sumsq) {
    if (!m) return fail(nullptr, SRX_E_ARG, "null matrix");
    if (m->csc) return fail(m->ctx, SRX_E_FORMAT, "srx_gene_moments walks cells: convert the CSC matrix with srx_matrix_to_csr");
    return stored_col_moments(m, cnt, sum, sumsq);
}

int32_t srx_compute_number(srx_mat* m, int32_t direction, uint32_t* out) {
    if (!m || !out) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    SRX_HIP(m->ctx, hipSetDevice(m->ctx->device));
    direction = eff_dir(m, direction);           // CSC: csc.rs:15-35 is csr.rs:16-38 with the directions exchanged
    if (direction == SRX_ROW) return row_number(m, out);
    if (direction != SRX_COLUMN) return fail(m->ctx, SRX_E_ARG, "bad direction %d", direction);
    std::vector<uint64_t> cnt(m->n_cols);
    SRX_TRY(stored_col_moments(m, cnt.data(), nullptr, nullptr));
    for (uint64_t j = 0; j < m->n_cols; ++j) out[j] = (uint32_t)cnt[j];   // reference counts in u32
    return SRX_OK;
}

int32_t srx_compute_sum(srx_mat* m, int32_t direction, double* out) {
    if (!m || !out) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    SRX_HIP(m->ctx, hipSetDevice(m->ctx->device));
    direction = eff_dir(m, direction);           // csc.rs:71-95
    if (direction == SRX_ROW) return row_stat(m, 0, out, nullptr);
    if (direction != SRX_COLUMN) return fail(m->ctx, SRX_E_ARG, "bad direction %d", direction);
    return stored_col_moments(m, nullptr, out, nullptr);
}

int32_t srx_compute_variance(srx_mat* m, int32_t direction, double* out) {
    if (!m || !out) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    SRX_HIP(m->ctx, hipSetDevice(m->ctx->device));
    direction = eff_dir(m, direction);           // csc.rs:138-180: naive + guard on the scattered axis, two-pass on the major one
    if (direction == SRX_ROW) return row_stat(m, 1, out, nullptr);
    if (direction != SRX_COLUMN) return fail(m->ctx, SRX_E_ARG, "bad direction %d", direction);
    std::vector<double> var;
    SRX_TRY(gene_variances(m, var));
    memcpy(out, var.data(), var.size() * sizeof(double));
    return SRX_OK;
}

int32_t srx_compute_std_dev(srx_mat* m, int32_t direction, double* out) {
    SRX_TRY(srx_compute_variance(m, direction, out));
    uint64_t n = eff_dir(m, direction) == SRX_ROW ? m->n_rows : m->n_cols;
    for (uint64_t i = 0; i < n; ++i) out[i] = std::sqrt(out[i]);   // csr.rs:227
    return SRX_OK;
}

int32_t srx_compute_qc_variables(srx_mat* m, uint32_t* num_per_cell, uint32_t* num_per_gene, double* expr_per_gene,
                                 double* expr_per_cell, double* variance_per_gene, double* variance_per_cell,
                                 double* std_dev_per_cell, double* std_dev_per_gene) {
    if (!m) return fail(nullptr, SRX_E_ARG, "null matrix");
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    if (m->csc) {                                // stored rows are genes: exchange the roles of the outputs
        std::swap(num_per_cell, num_per_gene);
        std::swap(expr_per_cell, expr_per_gene);
        std::swap(variance_per_cell, variance_per_gene);
        std::swap(std_dev_per_cell, std_dev_per_gene);
    }
    // per cell: ONE row pass
    if (num_per_cell || expr_per_cell || variance_per_cell || std_dev_per_cell) {
        std::vector<double> var;
        double* vp = variance_per_cell;
        if (!vp && std_dev_per_cell) { var.resize(m->n_rows); vp = var.data(); }
        SRX_TRY(row_qc(m, num_per_cell, expr_per_cell, vp));
        if (std_dev_per_cell)
            for (uint64_t i = 0; i < m->n_rows; ++i) std_dev_per_cell[i] = std::sqrt(vp[i]);       // csr.rs:225-228
    }
    // per gene: ONE column pass (the cached moments)
    if (num_per_gene || expr_per_gene || variance_per_gene || std_dev_per_gene) {
        std::vector<uint64_t> cnt;
        std::vector<double> sum, sq;
        SRX_TRY(fetch_moments(m, cnt, sum, sq));
        const uint64_t G = m->n_cols;
        if (num_per_gene) for (uint64_t j = 0; j < G; ++j) num_per_gene[j] = (uint32_t)cnt[j];
        if (expr_per_gene) memcpy(expr_per_gene, sum.data(), G * sizeof(double));
        if (variance_per_gene || std_dev_per_gene) {
            std::vector<double> var(G);
            finalize_variance(cnt.data(), sum.data(), sq.data(), G, var.data());
            if (variance_per_gene) memcpy(variance_per_gene, var.data(), G * sizeof(double));
            if (std_dev_per_gene) for (uint64_t j = 0; j < G; ++j) std_dev_per_gene[j] = std::sqrt(var[j]);
        }
    }
    return SRX_OK;
}

int32_t srx_compute_min_max(srx_mat* m, int32_t direction, double* mn, double* mx) {
    if (!m || !mn || !mx) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    direction = eff_dir(m, direction);           // csc.rs:186-212
    if (direction == SRX_ROW) return row_stat(m, 2, mn, mx);
    if (direction != SRX_COLUMN) return fail(ctx, SRX_E_ARG, "bad direction %d", direction);
    // per-gene extrema of a row shard are the shard's own (the path has sum all-reduces only): refuse rather than
    // return a different answer on every rank
    if (ctx->n_ranks > 1)
        return fail(ctx, SRX_E_ARG, "compute_min_max(Column) on a row-sharded context is shard-local: reduce the per-rank "
                                    "results on the host (min of mins, max of maxs) — rank %d of %d", ctx->rank, ctx->n_ranks);
    SRX_TRY(ensure_tiles(m));
    const uint64_t G = m->n_cols;
    unsigned long long* d;
    SRX_TRY(scratch(ctx, "minmax_keys", 2 * (G ? G : 1) * sizeof(unsigned long long), (void**)&d));
    unsigned g1 = (unsigned)((G + 255) / 256 + 1);
    hipLaunchKernelGGL(k_fill_u64, dim3(g1), dim3(256), 0, ctx->stream, d, G, (unsigned long long)f64_key(INFINITY));
    hipLaunchKernelGGL(k_fill_u64, dim3(g1), dim3(256), 0, ctx->stream, d + G, G, (unsigned long long)f64_key(-INFINITY));
    uint64_t nb, rpb;
    SRX_TRY(block_geometry(m, nb, rpb));
    const size_t lds = (size_t)m->tile_genes * 16;
    dim3 grid((unsigned)(nb * m->n_tiles));
    if (is_f32(m)) {
        SRX_HIP(ctx, hipFuncSetAttribute((const void*)k_gene_minmax<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_gene_minmax<float>), grid, dim3(kMomThreads), lds, ctx->stream, m->d_indptr, m->d_tile_ptr,
                           m->d_indices, (const float*)m->d_values, m->n_rows, G, m->n_tiles, m->tile_genes, rpb, d, d + G);
    } else {
        SRX_HIP(ctx, hipFuncSetAttribute((const void*)k_gene_minmax<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_gene_minmax<double>), grid, dim3(kMomThreads), lds, ctx->stream, m->d_indptr, m->d_tile_ptr,
                           m->d_indices, (const double*)m->d_values, m->n_rows, G, m->n_tiles, m->tile_genes, rpb, d, d + G);
    }
    SRX_HIP(ctx, hipGetLastError());
    std::vector<unsigned long long> keys(2 * G);
    SRX_TRY(d2h(ctx, keys.data(), d, 2 * G * sizeof(unsigned long long)));
    for (uint64_t j = 0; j < G; ++j) { mn[j] = key_f64(keys[j]); mx[j] = key_f64(keys[G + j]); }
    return SRX_OK;
}

int32_t srx_normalize_total_inplace(srx_mat* m, double target_sum, int32_t direction) {
    if (!m) return fail(nullptr, SRX_E_ARG, "null matrix");
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    direction = eff_dir(m, direction);           // scale_row_csc / scale_col_csc, scale/mod.rs:25-57,104-139
    SRX_TRY(promote_to_f64(m));                  // X becomes DynCsrMatrix::F64 (scale/mod.rs:74-83)
    if (direction == SRX_ROW) return launch_normalize(m, target_sum, true, false);
    if (direction != SRX_COLUMN) return fail(ctx, SRX_E_ARG, "bad direction %d", direction);
    SRX_TRY(ensure_moments(m));
    uint64_t g = (m->nnz + 255) / 256;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    if (is_f32(m))
        hipLaunchKernelGGL((k_col_scale<float>), dim3((unsigned)g), dim3(256), 0, ctx->stream, m->d_indices,
                           (float*)m->d_values, m->nnz, m->d_sum, target_sum);
    else
        hipLaunchKernelGGL((k_col_scale<double>), dim3((unsigned)g), dim3(256), 0, ctx->stream, m->d_indices,
                           (double*)m->d_values, m->nnz, m->d_sum, target_sum);
    SRX_HIP(ctx, hipGetLastError());
    m->dtype = SRX_F64;
    touch(m);
    return SRX_OK;
}

int32_t srx_select_hvg(srx_mat* m, uint64_t n, uint64_t* idx_out, uint64_t* n_out) {
    if (!m || !n_out) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    SRX_HIP(m->ctx, hipSetDevice(m->ctx->device));
    std::vector<double> var;
    if (m->csc) {                                // compute_variance(Column) of a CSC matrix: csc.rs:164-177 (NaN for an empty gene)
        var.resize(m->n_rows);
        SRX_TRY(row_stat(m, 1, var.data(), nullptr));
    } else {
        SRX_TRY(gene_variances(m, var));
    }
    std::vector<uint64_t> sel;
    SRX_TRY(select_hvg_host(m->ctx, var, n, sel));
    if (idx_out) memcpy(idx_out, sel.data(), sel.size() * sizeof(uint64_t));
    *n_out = sel.size();
    return SRX_OK;
}

}  // extern "C"
