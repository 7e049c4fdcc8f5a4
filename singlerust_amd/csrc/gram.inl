// gram.inl — included by pca_form.hip inside namespace srx (the kernels share pca_internal.hpp's helpers and constants).
// G = A^T A from the row-major compacted matrix: record counts, owner buckets, the stripe kernel, expansion to C.

// ---- explicit sparse Gram: G = A^T A from the ROW-MAJOR compacted matrix ----------------------------
// Round 2 design.  The work is N m(m+1)/2 scalar products (m = kept entries of a cell), each ending in an
// f64 LDS atomic; what the round-1 kernel paid on top of that was two staged LDS reads and ~17 VALU
// instructions of index arithmetic per product slice, with 22-30 of 64 lanes busy per atomic
// (profiles/r01_pmc_gram_v6.md).  Here ONE wave instruction is one (cell, entry): the entry (ja, va) and the
// SUFFIX of its row — the entries with column >= ja, contiguous in the row-major layout — so that lane q
// holds (jb_q, vb_q) straight from a coalesced global load and adds va * vb_q to G[ja][jb_q].  No staging,
// no per-product index arithmetic, every cell's upper-triangle products exactly once.
//
// Ownership: G's upper triangle is cut into STRIPES of SR rows; workgroup w owns stripes w and
// n_stripes - 1 - w (long rows at the top, short ones at the bottom: SR (k + SR) doubles of LDS per
// workgroup whatever w — 64 KiB at k = 2000, SR = 4, two workgroups per CU).  The entries a workgroup needs
// are those whose column lies in its two stripes: k_bucket sorts the entries of every block of kBucketRows
// cells by owner, so that a wave fetches its share of a block as one contiguous run of 8-byte records
// (position of the entry relative to the block, suffix length).  All workgroups walk the row blocks in the
// same order at about the same pace, so the row-major matrix streams through L2 / Infinity Cache once per
// XCD while every cell is visited by the ~m workgroups that own one of its entries.
__device__ __forceinline__ double gram_product(float a, float b) { return (double)(a * b); }
__device__ __forceinline__ double gram_product(double a, double b) { return a * b; }

// Records of a block of `rblk` cells, grouped by owning workgroup (counting sort in LDS; the order inside a group is
// whatever the LDS atomics make it — the Gram sums are order-dependent in their last bits anyway).
// boff[rb][w] .. boff[rb][w + 1]: records of owner w, relative to the block's first record (rec_base[rb]).
constexpr int kBucketThreads = 1024;
constexpr int kBucketGroup = 8;           // consecutive rows a wave walks as one flat run
constexpr int kBucketUnroll = 8;          // 64-entry chunks of the run in flight

// per-block record totals (k_bucket's layout needs their prefix sums before it runs)
__global__ __launch_bounds__(256) void k_rec_count(const int64_t* __restrict__ rm_ptr, uint64_t n_rows, uint32_t rblk,
                                                   int64_t* __restrict__ blk_total) {
    const uint64_t r0 = (uint64_t)blockIdx.x * rblk;
    const uint64_t r1 = r0 + rblk < n_rows ? r0 + rblk : n_rows;
    uint64_t acc = 0;
    for (uint64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) acc += gram_row_records((uint64_t)(rm_ptr[r + 1] - rm_ptr[r]));
    acc = wave_sum(acc);
    __shared__ uint64_t part[4];
    if (lane_id() == 0) part[threadIdx.x / kWave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) blk_total[blockIdx.x] = (int64_t)(part[0] + part[1] + part[2] + part[3]);
}
// exclusive scan of the block totals by one workgroup: base[0 .. n], base[n] = all records
// (`also`, nullable: one more number the host wants with the same read-back — copied to base[n + 1])
__global__ __launch_bounds__(1024) void k_rec_scan(const int64_t* __restrict__ blk_total, uint64_t n, int64_t* __restrict__ base,
                                                   const int64_t* __restrict__ also = nullptr) {
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = lane_id(), wave = threadIdx.x / kWave;
    for (uint64_t i0 = 0; i0 < n; i0 += 1024) {
        const uint64_t i = i0 + threadIdx.x;
        const int64_t v = i < n ? blk_total[i] : 0;
        int64_t inc = v;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const int64_t o = __shfl_up(inc, off, kWave);
            if (lane >= off) inc += o;
        }
        if (lane == kWave - 1) wsum[wave] = inc;
        __syncthreads();
        int64_t before = carry_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (i < n) base[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        base[n] = carry_s;
        if (also) base[n + 1] = *also;
        base[n + 2] = 0;                  // the bucket pass's value statistics (gram_stat below): zeroed here, before it runs
        base[n + 3] = 0;
    }
}
// Value statistics of the compacted matrix, made by the bucket pass for the stripe kernel's fixed-point mode (f32 entries):
// [0] bits of max |v|, [1] ~bits of the smallest non-zero |v| (a max again: 0 = none), [2] != 0: a negative value was seen.
// The decision itself (host and device: srx_gram_mode_info applies it to the words the launch read): fixed point when no value is
// negative and the binary exponents of all non-zero |v| lie within 6 of the largest's; kq: p < 2^(2 emax + 2), so p 2^kq < 2^31.
__host__ __device__ inline bool gram_fixed_point_mode(uint32_t vmax_b, uint32_t nmin, uint32_t neg, int& kq) {
    const uint32_t vmin_b = ~nmin;
    const int emax = (int)(vmax_b >> 23) - 127, emin = (int)(vmin_b >> 23) - 127;
    kq = 29 - 2 * emax;
    return neg == 0u && vmax_b != 0u && vmax_b < 0x7f800000u && nmin != 0u && emax - emin <= 6 && emax > -48 && emax < 48;
}
// f64 entries (round 5): the same rule on the HIGH words of the doubles (11-bit exponents); products are formed in f64 already scaled,
// p 2^kq < 2^47 (kq = 45 - 2 emax), rounded to integers by the 1.5 x 2^52 constant — a chunk of at most 65536 cells (gram_plan's cap,
// asserted there) adds at most 2^16 of them to an accumulator: < 2^63, read back as UNSIGNED by the flush.  Half a unit = 2^-48 of the
// largest possible product.  ACCURACY BOUND: the exponent spread allowed is 4, not the f32 rule's 6 — a product of two
// values at the small end is then >= 2^-10 of the largest possible one and keeps 2^-38 of itself (3.6e-12; with 6: 2^-34 = 6e-11,
// too close to the f64 path's default residual tolerance of 1e-9); the sums of typical products keep ~2^-46.  The f64 atomics keep
// 1e-16 per product and depend on the order they land in; SRX_GRAM_F64_ATOMICS=1 forces them (either storage).  Resident solves
// decide the mode once, from statistics summed over the ranks; a backed session decides per tile and rank — its sums then depend
// on the tile / rank layout by the quantum.
__host__ __device__ inline bool gram_fixed_point_mode64(uint32_t vmax_hi, uint32_t nmin_hi, uint32_t neg, int& kq) {
    const uint32_t vmin_hi = ~nmin_hi;
    const int emax = (int)(vmax_hi >> 20) - 1023, emin = (int)(vmin_hi >> 20) - 1023;
    kq = 45 - 2 * emax;
    return neg == 0u && vmax_hi != 0u && vmax_hi < 0x7ff00000u && nmin_hi != 0u && emax - emin <= 4 && emax > -200 && emax < 200;
}
__device__ __forceinline__ uint32_t* gram_stat(int64_t* rec_base, uint64_t n_rblk) { return reinterpret_cast<uint32_t*>(rec_base + n_rblk + 2); }

// Sharded rows: the mode has to be the SAME on every rank (it decides how the products are rounded), so the statistics are
// combined over the ranks first.  The path's collective is a SUM of doubles: each rank marks the biased exponent of its
// largest |v| and of its smallest non-zero |v| in a 256-bin histogram each (+ one count of "a negative value was seen") ...
// (EB exponent bits: 8 for f32 entries, 11 for the high words of f64 entries; SH: the shift that brings the biased exponent down)
template <int EB> constexpr int gstat_bins() { return 2 * (1 << EB) + 1; }
constexpr int kGstatBins = gstat_bins<8>();
template <int EB>
__global__ void k_gstat_onehot(const uint32_t* __restrict__ gstat, double* __restrict__ bins) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    constexpr int NB = 1 << EB, SH = 31 - EB;
    const uint32_t vmax_b = gstat[0], nmin = gstat[1];
    if (vmax_b) bins[vmax_b >> SH] += 1.0;                    // (NaN / inf land in the last bin: the range test fails everywhere)
    if (nmin) bins[NB + ((~nmin) >> SH)] += 1.0;
    if (gstat[2]) bins[2 * NB] += 1.0;
}
// ... and after the sum every rank rebuilds the three words from the highest / lowest marked bin: the kernel's test only
// looks at the exponents, so a mantissa of zero stands for the values.  A rank without entries marks nothing.
template <int EB>
__global__ void k_gstat_decode(const double* __restrict__ bins, uint32_t* __restrict__ gstat) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    constexpr int NB = 1 << EB, SH = 31 - EB;
    int emax = -1, emin = -1;
    for (int b = 0; b < NB; ++b) {
        if (bins[b] != 0.0) emax = b;
        if (bins[NB + b] != 0.0 && emin < 0) emin = b;
    }
    gstat[0] = emax < 0 ? 0u : ((uint32_t)emax << SH) | (emax == NB - 1 ? 1u : 0u);
    gstat[1] = emin < 0 ? 0u : ~((uint32_t)emin << SH);
    gstat[2] = bins[2 * NB] != 0.0 ? 1u : 0u;
}

template <typename VT>
__global__ __launch_bounds__(kBucketThreads) void k_bucket(const int64_t* __restrict__ rm_ptr, const GramPk<VT>* __restrict__ rm,
                                                           uint64_t n_rows, uint32_t rblk, int k, int sr_shift, int n_wg, int n_stripes,
                                                           const int64_t* __restrict__ rec_base, uint32_t* __restrict__ boff,
                                                           GramRec<VT>* __restrict__ recs, uint32_t* __restrict__ gstat /* gram_stat() */,
                                                           int64_t n_recs, double* __restrict__ zero_g, uint64_t n_zero_g) {
    extern __shared__ double lds_raw[];
    __shared__ uint32_t s_stat[3];
    // side jobs that used to be memset launches in front of this kernel (4-5 us each on the step's critical path): the
    // kGramUnroll records behind the last one (the stripe kernel's batches read past the end) and, for a fresh sum, the packed
    // triangle the stripe kernel adds into
    if (blockIdx.x == 0)
        for (uint32_t e = threadIdx.x; e < kGramUnroll * sizeof(GramRec<VT>) / 4; e += kBucketThreads)
            reinterpret_cast<uint32_t*>(recs + n_recs)[e] = 0u;
    if (zero_g)
        for (uint64_t e = (uint64_t)blockIdx.x * kBucketThreads + threadIdx.x; e < n_zero_g; e += (uint64_t)gridDim.x * kBucketThreads)
            zero_g[e] = 0.0;
    if (threadIdx.x < 3) s_stat[threadIdx.x] = 0u;
    uint32_t* hist = reinterpret_cast<uint32_t*>(lds_raw);        // n_wg + 1 counters, then rblk + 1 row ends
    uint32_t* rptr = hist + n_wg + 1;                             // row starts of the block, relative to its first entry
    const uint64_t rb = blockIdx.x;
    const uint64_t r0 = rb * rblk;
    const uint64_t r1 = r0 + rblk < n_rows ? r0 + rblk : n_rows;
    const int nr = (int)(r1 - r0);
    const int lane = lane_id(), wave = threadIdx.x / kWave;
    const int64_t base = rm_ptr[r0];
    for (int e = threadIdx.x; e <= n_wg; e += kBucketThreads) hist[e] = 0u;
    for (int e = threadIdx.x; e <= nr + kBucketGroup; e += kBucketThreads)
        rptr[e] = (uint32_t)(rm_ptr[r0 + (e < nr ? e : nr)] - base);          // padded by a group of empty rows
    __syncthreads();
    const GramPk<VT>* rmb = rm + base;
    // A wave takes kBucketGroup consecutive rows at a time: their entries are one contiguous run, walked flat 64 at a time
    // (coalesced), and the row of an entry is found by comparing with the group's three inner row starts — the suffix
    // length of an entry (its record count) needs the row's end.  `visit(p, j, v, row_end)` for every entry of the block.
    auto walk = [&](auto visit) {
        for (int g0 = wave * kBucketGroup; g0 < nr; g0 += (kBucketThreads / kWave) * kBucketGroup) {
            uint32_t b[kBucketGroup + 1];
#pragma unroll
            for (int i = 0; i <= kBucketGroup; ++i) b[i] = rptr[g0 + i];
            for (uint32_t p0 = b[0]; p0 < b[kBucketGroup]; p0 += kBucketUnroll * kWave) {
                GramPk<VT> x[kBucketUnroll];
#pragma unroll
                for (int u = 0; u < kBucketUnroll; ++u) {
                    const uint32_t p = p0 + u * kWave + lane;
                    x[u].j = -1;
                    if (p < b[kBucketGroup]) x[u] = rmb[p];
                }
#pragma unroll
                for (int u = 0; u < kBucketUnroll; ++u) {
                    const uint32_t p = p0 + u * kWave + lane;
                    if (x[u].j < 0) continue;
                    uint32_t end = b[1];
#pragma unroll
                    for (int i = 1; i < kBucketGroup; ++i) end = p >= b[i] ? b[i + 1] : end;
                    visit(p, x[u].j, x[u].v, end);
                }
            }
        }
    };
    uint32_t vmax_b = 0u, vmin_nb = 0u, neg = 0u;
    walk([&](uint32_t p, int j, VT v, uint32_t end) {
        __hip_atomic_fetch_add(&hist[gram_owner(j, sr_shift, n_wg, n_stripes)], (end - p + 63u) >> 6, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
        {                                       // (a stored zero makes the smallest |v| zero: the range test fails, the f64 atomics run)
            uint32_t b;                             // f64 entries: the high word (sign, exponent, 20 mantissa bits)
            if constexpr (sizeof(VT) == 4) b = __float_as_uint(v);
            else b = (uint32_t)((unsigned long long)__double_as_longlong(v) >> 32);
            const uint32_t ab = b & 0x7fffffffu;
            vmax_b = ab > vmax_b ? ab : vmax_b;
            vmin_nb = ~ab > vmin_nb ? ~ab : vmin_nb;
            neg |= b;
        }
    });
    if (vmax_b) atomicMax(&s_stat[0], vmax_b);
    if (vmin_nb) atomicMax(&s_stat[1], vmin_nb);
    if (neg >> 31) atomicOr(&s_stat[2], 1u);
    __syncthreads();
    if (threadIdx.x < 3 && gstat && s_stat[threadIdx.x]) {
        if (threadIdx.x == 2) atomicOr(&gstat[2], 1u);
        else atomicMax(&gstat[threadIdx.x], s_stat[threadIdx.x]);
    }
    // exclusive scan of the n_wg counters by wave 0, 64 at a time
    if (wave == 0) {
        uint32_t carry = 0;
        for (int c0 = 0; c0 < n_wg; c0 += kWave) {
            const int i = c0 + lane;
            const uint32_t v = i < n_wg ? hist[i] : 0u;
            uint32_t inc = v;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const uint32_t o = __shfl_up(inc, off, kWave);
                if (lane >= off) inc += o;
            }
            if (i < n_wg) {
                hist[i] = carry + inc - v;
                boff[rb * (uint64_t)(n_wg + 1) + i] = carry + inc - v;
            }
            carry += __shfl(inc, kWave - 1, kWave);
        }
        if (lane == 0) boff[rb * (uint64_t)(n_wg + 1) + n_wg] = carry;
    }
    __syncthreads();
    GramRec<VT>* rcb = recs + rec_base[rb];
    walk([&](uint32_t p, int j, VT v, uint32_t end) {
        const uint32_t len = end - p, nch = (len + 63u) >> 6;
        uint32_t slot = __hip_atomic_fetch_add(&hist[gram_owner(j, sr_shift, n_wg, n_stripes)], nch, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (sizeof(VT) == 4) {
            // f32 entries (round 5): the record in the form the stripe kernel's scalar unit consumes it — the piece's BYTE offset in the
            // block, and n0 | n1 << 6 | row base << 12: n0 / n1 = the lanes of a HALF wave whose first / second entry lies inside the
            // piece (lane l of a half takes entries 2 l and 2 l + 1), the row's accumulators as a signed byte offset
            const uint32_t rb = (uint32_t)gram_row_base(j, k, sr_shift, n_wg, n_stripes) << 15;      // (x 8 bytes, above the 12 count bits)
            for (uint32_t o = 0; o < len; o += kWave, ++slot) {
                const uint32_t L = len - o < (uint32_t)kWave ? len - o : (uint32_t)kWave;
                rcb[slot] = GramRec<VT>{(p + o) * 8u, ((L + 1u) >> 1) | ((L >> 1) << 6) | rb, v};
            }
        } else {
            const uint32_t rb8 = (uint32_t)gram_row_base(j, k, sr_shift, n_wg, n_stripes) << 11;      // BYTE offset of the row's accumulators (x 8, signed), above the length byte
            for (uint32_t o = 0; o < len; o += kWave, ++slot)
                rcb[slot] = GramRec<VT>{p + o, (len - o < (uint32_t)kWave ? len - o : (uint32_t)kWave) | rb8, v};
        }
    });
}

__device__ __forceinline__ float readfirst_v(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}
__device__ __forceinline__ double readfirst_v(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// entries [0, len) of `base`, lane l taking entry l; lanes >= len return zeros without touching memory
__device__ __forceinline__ GramPk<float> suffix_load(const GramPk<float>* base, uint32_t len, int lane) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<GramPk<float>*>(base), (short)0, (int)(len * 8u), 0x00020000);
    const auto x = __builtin_amdgcn_raw_buffer_load_b64(rs, lane * 8, 0, 0);
    return GramPk<float>{(int)x[0], __builtin_bit_cast(float, (unsigned)x[1])};
}
__device__ __forceinline__ GramPk<double> suffix_load(const GramPk<double>* base, uint32_t len, int lane) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<GramPk<double>*>(base), (short)0, (int)(len * 16u), 0x00020000);
    const auto x = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, 0);
    GramPk<double> e;
    e.j = (int)x[0];
    e.pad_ = 0;
    e.v = __builtin_bit_cast(double, ((unsigned long long)(unsigned)x[3] << 32) | (unsigned)x[2]);
    return e;
}

template <typename VT>
__global__ __launch_bounds__(kGramWaves * 64, 8) void k_gram_stripes(
    const int64_t* __restrict__ rm_ptr, const GramPk<VT>* __restrict__ rm, const uint32_t* __restrict__ boff,
    const int64_t* __restrict__ rec_base, const GramRec<VT>* __restrict__ recs, uint64_t n_rblk, uint32_t rblk, int k,
    int sr_shift, int n_wg, int n_stripes, uint32_t n_chunk, int w0 /* first owner of this launch */, int n_w /* owners in it */,
    double* __restrict__ Gp /* packed upper triangle, ACCUMULATED into (global f64 atomics) */,
    const uint32_t* __restrict__ gstat /* gram_stat(), nullable */) {
    using Entry = GramPk<VT>;
    using Rec = GramRec<VT>;
    // suffix loads in flight per batch: 16-byte f64 entries take twice the registers (8 of them spilled)
    constexpr int kUnroll = sizeof(VT) == 8 ? kGramUnroll / 2 : kGramUnroll;
    extern __shared__ double acc[];
    // (Round 6, measured and not kept: the owners that land on one XCD — consecutive workgroup ids go round the 8 XCDs — as a CONTIGUOUS
    //  range of w, so that an XCD's L2 would only have to bring in the row suffixes from its first owner's columns on, 1 - x / 16 of
    //  the matrix instead of all of it: FETCH_SIZE did not move (7.64e6 KiB against 7.62e6: the traffic is capacity misses on
    //  re-reads, 59 % L2 hits either way, not first touches) and the launch went from 2.48 to 2.77 ms: profiles/r06_pmc_gram.md.)
    const int w = w0 + blockIdx.x % n_w, z = blockIdx.x / n_w;
    const int SR = 1 << sr_shift;
    const int a0 = w * SR, b0 = (n_stripes - 1 - w) * SR;
    const int WA = k - a0, WB = k - b0 > 0 ? k - b0 : 0;
    const int n_acc = SR * (WA + WB);
    for (int e = threadIdx.x; e < n_acc; e += blockDim.x) acc[e] = 0.0;
    {
        // the assembly core forms LDS addresses as (row base of the record) + 8 x column: the accumulators must start at LDS offset 0
        // (they do: the kernel has no static shared memory); anything else must fail loudly, not add into the wrong words
        typedef __attribute__((address_space(3))) double lds_double;
        if ((uint32_t)(uintptr_t)(lds_double*)acc != 0u) __builtin_trap();
    }
    __syncthreads();
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    // FIXED-POINT mode (f32 entries, decided from the bucket pass's statistics — the same for every workgroup of the launch):
    // no negative value and the binary exponents of all non-zero |v| within 6 of the largest's (normalised + log1p'd counts are).  A
    // product va * v is then formed in f32 ALREADY SCALED by a power of two (va carries it: p 2^kq < 2^31, kq from the exponent of
    // max |v|), rounded to an integer and added with a 64-bit INTEGER LDS atomic — `ds_add_u64` on random addresses runs at 1.9x the
    // rate of `ds_add_f64` (bench_micro/lds_atomic_banks.hip) and the LDS pipe was this kernel's wall; the conversion costs what the
    // f32 -> f64 conversion cost.  Every product is off by at most half a unit = 2^-30 of the largest possible one — what the f32
    // rounding of a typical product is (2^-24 of itself); a product above 2^-7 of the largest has >= 2^24 units and loses nothing.
    // Integer sums do not depend on the order the atomics land in.  Otherwise (negative values, a wide range, f64 entries): f32
    // products converted to f64, `ds_add_f64`, as before.
    bool fx = false;
    VT fx_scale = (VT)1;
    double fx_inv = 1.0;
    if (gstat) {
        int kq;
        const bool ok = sizeof(VT) == 4 ? gram_fixed_point_mode(gstat[0], gstat[1], gstat[2], kq)
                                        : gram_fixed_point_mode64(gstat[0], gstat[1], gstat[2], kq);
        if (ok) {
            fx = true;
            fx_scale = (VT)__longlong_as_double((long long)(1023 + kq) << 52);      // 2^kq (exact in either type)
            fx_inv = __longlong_as_double((long long)(1023 - kq) << 52);
        }
    }
    // Workgroup = (owner w, chunk z of n_chunk consecutive row blocks); blockIdx = z * n_wg + w, so the dispatcher starts
    // all owners of a chunk together and they walk its rows in the same order: what one workgroup pulls into its
    // XCD's L2 the ~60 others on that XCD hit (free-running persistent workgroups drift tens of MB apart: L2 hit rate
    // 12 %, 50 GB of fabric reads per launch at c3).  Wave v takes blocks v, v + 16, ... of the chunk.
    //
    // A wave reads its records 64 at a time (one 12-byte vector load per lane, the slab after this one in flight while this
    // one is worked through) and hands them out kUnroll at a time: v_readlane makes (pos, len, rbase, va) of a record
    // wave-uniform, the suffix load takes (scalar base, 32-bit lane offset), and what is left per record is lane < len, the
    // product, its conversion and the LDS address.  Two batches are in flight: one being fetched, one being added.
    // (Records fetched by scalar loads, one batch ahead: the stream alone cost 3.1 ms — every batch a dependent round trip
    // through the scalar cache; slabs fetched only when the previous one was used up: 4 of 5 slab loads exposed.)
    struct Blk {
        uint32_t n;
        const Entry* rmb;
        const Rec* rc;
    };
    const uint64_t rb0 = (uint64_t)z * n_chunk, rb1 = rb0 + n_chunk < n_rblk ? rb0 + n_chunk : n_rblk;
    auto scalars = [&](uint64_t rb) -> Blk {
        Blk k_{0u, rm, recs};
        if (rb < rb1) {
            const uint32_t* bo = boff + rb * (uint64_t)(n_wg + 1) + w;
            const uint32_t o0 = bo[0], o1 = bo[1];
            k_.n = o1 - o0;
            k_.rmb = rm + rm_ptr[rb * rblk];
            k_.rc = recs + rec_base[rb] + o0;
        }
        return k_;
    };
    uint64_t rb = rb0 + wave;
    uint32_t i0 = 0;
    Blk cur = scalars(rb), nxt = scalars(rb + kGramWaves);
    struct Slab {                         // up to 64 records of one block, lane l holding record l
        Rec r;
        uint32_t n;
        const Entry* rmb;
        unsigned long long base;          // rmb, made wave-uniform for the compiler (v_readfirstlane: once per slab)
    };
    auto next_slab = [&]() -> Slab {
        while (i0 >= cur.n && rb < rb1) {
            rb += kGramWaves;
            cur = nxt;
            nxt = scalars(rb + kGramWaves);
            i0 = 0;
        }
        Slab sl;
        sl.rmb = cur.rmb;
        {
            const unsigned long long bp = (unsigned long long)reinterpret_cast<uintptr_t>(cur.rmb);
            sl.base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bp >> 32)) << 32) |
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(bp & 0xffffffffull));
        }
        sl.n = i0 < cur.n ? (cur.n - i0 < (uint32_t)kWave ? cur.n - i0 : (uint32_t)kWave) : 0u;
        sl.r = Rec{0u, 0u, (VT)0};
        if ((uint32_t)lane < sl.n) sl.r = cur.rc[i0 + lane];
        if (fx) sl.r.va *= fx_scale;              // (a power of two: exact; once per 64 records)
        i0 += kWave;
        return sl;
    };
    struct Loaded {                       // a batch of records with their suffix entries on the way
        Entry e[kUnroll];
        uint32_t lenrb[kUnroll];
        VT va[kUnroll];
    };
    auto batch = [&](const Slab& sl, int u0) -> Loaded {
        Loaded l;
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {       // lanes past sl.n hold empty records: len 0, pos 0
            const uint32_t pos = (uint32_t)__builtin_amdgcn_readlane((int)sl.r.pos, u0 + u);
            l.lenrb[u] = (uint32_t)__builtin_amdgcn_readlane((int)sl.r.lenrb, u0 + u);
            l.va[u] = readlane_v(sl.r.va, u0 + u);
            // a buffer load whose range is the suffix itself: lanes past `len` are out of range and fetch nothing (the L1
            // works through a wave's load 64 bytes at a time — reading all 64 lanes of every ~36-entry suffix was 2 of the
            // kernel's 4.2 ms), the address is (scalar base, constant lane offset), and there is no branch or exec mask
            // around the load for the compiler's wait counting to trip over
            l.e[u] = suffix_load(sl.rmb + pos, l.lenrb[u] & 0xffu, lane);
        }
        return l;
    };
    auto process = [&](const Loaded& l, auto fxc) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            double* const row = reinterpret_cast<double*>(reinterpret_cast<char*>(acc) + ((int)l.lenrb[u] >> 8));      // (signed: a row base may be negative)
            const uint32_t len = l.lenrb[u] & 0xffu;
            if ((uint32_t)lane < len) {
                if constexpr (decltype(fxc)::value && sizeof(VT) == 8) {
                    // va carries 2^kq: the product rounded to an integer by the 1.5 x 2^52 constant (0 <= p 2^kq < 2^47), its bits minus
                    // the constant's = the integer; ds_add_u64 runs at 1.9x the rate of ds_add_f64
                    const double kMagic = 6755399441055744.0;
                    const unsigned long long bits =
                        (unsigned long long)__double_as_longlong(__builtin_fma((double)l.va[u], (double)l.e[u].v, kMagic)) - 0x4338000000000000ull;
                    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(row) + l.e[u].j, bits, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    __hip_atomic_fetch_add(row + l.e[u].j, gram_product(l.va[u], l.e[u].v), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    };
    // All batches of slab `sl`, two in flight (ping-pong between two register sets: a `now = next` copy makes the compiler
    // wait for next's loads).  The OTHER slab (used up before this one) is refilled right after this slab's first batch has
    // gone out: when its records are first read, a slab later, every load issued after it has long been waited for — the
    // compiler waits for ALL outstanding loads at that point (vmcnt is in order and it cannot count across the loop), so a
    // refill issued last would be a full round trip exposed per slab.
    // f32 entries (8 bytes): TWO records per load instruction.  What the operand fetch costs is the load INSTRUCTION — one wave
    // load per ~18.6 clocks and CU whether it serves 16, 36 or 64 lanes, one run of addresses or four (bench_micro/l2_gather.hip:
    // 3.0-3.2 ms per 1e8 loads on 256 CUs, the rate this kernel ran at with one ~36-entry suffix per load) — so lanes 0-31 take
    // record 2p and lanes 32-63 record 2p + 1, 16 bytes (two consecutive entries) per lane: half the load instructions, the same
    // number of LDS atomics (two per lane).  Lanes past a record's end re-read its first two entries (same line: nothing more is
    // fetched; no branch or exec mask around the load).
    constexpr bool kPair = sizeof(VT) == 4;
    constexpr int kL = kUnroll / 2;                  // load instructions of a batch of kUnroll records
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    struct LoadedP {                                 // the two records of each load as SCALARS (v_readlane results), the loaded entries
        u4 raw[kL];                                  // (j, v) of entries 2 l and 2 l + 1 of the lane's record
        uint32_t lrA[kL], lrB[kL];                   // n0 | n1 << 6 | row base (bytes) << 12 of the record of lanes 0-31 / 32-63
        uint32_t vaA[kL], vaB[kL];                   // the records' values (bit patterns)
    };
    const bool hi = lane >= 32;
    const uint32_t l16 = (uint32_t)(lane & 31) * 16u;        // a lane's byte offset inside its record's piece
    auto va_word = [](VT v) -> int {
        if constexpr (sizeof(VT) == 4) return __builtin_bit_cast(int, v);
        else return 0;                                       // (the pair path is the f32 one)
    };
    // Round 5: the delivery of a record's fields to its half of the wave is gone.  Until then every field (position, counts | row
    // base, value) went from two v_readlane results through two v_mov and a v_cndmask into a per-lane register — a VOP3 takes ONE
    // scalar operand and the lane mask is one — and the clamp of the lanes past a piece's end, the 64-bit address and the two
    // `lane < count` compares were VALU work too: 28.6 VALU instructions per load, VALU 100 % busy (profiles/r05_pmc_gram.md).
    // Now the CONSUMERS run once per half with the half's scalars as operands and the lane masks come from the scalar unit:
    // exec = s_bfm_b64(count, 0 | 32) — the lanes of a half whose entry exists; the load is issued for exactly those lanes (no clamp,
    // no compare), a product is v_fma_f32(s_va, v_b, 0.5) -> v_cvt_u32 -> v_lshl_add(v_j, 3, s_row) -> ds_add_u64.  2 + 13 VALU and
    // ~18 scalar instructions per load beside the six v_readlane.  Inline assembly: the compiler turns every per-half choice back
    // into selects.  The loads are invisible to its wait counting, so the waits are explicit: vmcnt(after + kL - 1 - u) for load u
    // of a set, `after` = the loads issued since that set's (the other set's kL, or none behind a slab's last set; memory reads
    // return in order; the compiler's own waits — for a slab's records — can only come out too strict, not too lax).
    // (The blocks below end with `s_mov_b64 exec, -1`: they run under wave-uniform control flow only — a 1024-thread workgroup, slab /
    //  batch loops whose bounds are scalars — so the mask on entry IS -1; saving and restoring it would be two more scalar instructions
    //  per block on a kernel that issues 18 of them per load already.  tests/test_abi_cpu.py walks the listing: no instruction outside
    //  the assembly blocks may touch a register whose load is still in flight.)
    auto batchP = [&](const Slab& sl, int u0) -> LoadedP {
        LoadedP l;
        const unsigned long long base = sl.base;     // (the block's first entry as a SCALAR pair: the load's saddr operand)
      if constexpr (kPair) {                           // (f32 entries only: this lambda is instantiated for f64 too)
        static_assert(kL == 4, "the load block below is written for four loads");
        uint32_t pA[kL], pB[kL];
#pragma unroll
        for (int u = 0; u < kL; ++u) {               // lanes past sl.n hold empty records: counts 0, position 0
            pA[u] = (uint32_t)__builtin_amdgcn_readlane((int)sl.r.pos, u0 + 2 * u);
            pB[u] = (uint32_t)__builtin_amdgcn_readlane((int)sl.r.pos, u0 + 2 * u + 1);
            l.lrA[u] = (uint32_t)__builtin_amdgcn_readlane((int)sl.r.lenrb, u0 + 2 * u);
            l.lrB[u] = (uint32_t)__builtin_amdgcn_readlane((int)sl.r.lenrb, u0 + 2 * u + 1);
            l.vaA[u] = (uint32_t)__builtin_amdgcn_readlane(va_word(sl.r.va), u0 + 2 * u);
            l.vaB[u] = (uint32_t)__builtin_amdgcn_readlane(va_word(sl.r.va), u0 + 2 * u + 1);
        }
        // (s_bfm_b64 takes its width from bits 0-5 of the operand: n0 sits there, no extraction; one block for the four loads: one
        //  restore of exec)
        uint32_t off;
        unsigned long long m;
#define SRX_GRAM_LOAD(U)                                                           \
        "s_bfm_b64 %[m], %[lrA" U "], 0\n\t"                                       \
        "s_mov_b64 exec, %[m]\n\t"                                                 \
        "v_add_u32 %[off], %[pA" U "], %[l16]\n\t"                                  \
        "s_bfm_b64 exec, %[lrB" U "], 32\n\t"                                      \
        "v_add_u32 %[off], %[pB" U "], %[l16]\n\t"                                  \
        "s_or_b64 exec, exec, %[m]\n\t"                                            \
        "global_load_dwordx4 %[raw" U "], %[off], %[base]\n\t"
        asm volatile(SRX_GRAM_LOAD("0") SRX_GRAM_LOAD("1") SRX_GRAM_LOAD("2") SRX_GRAM_LOAD("3") "s_mov_b64 exec, -1"
                     : [raw0] "=&v"(l.raw[0]), [raw1] "=&v"(l.raw[1]), [raw2] "=&v"(l.raw[2]), [raw3] "=&v"(l.raw[3]), [off] "=&v"(off),
                       [m] "=&s"(m)
                     : [lrA0] "s"(l.lrA[0]), [lrB0] "s"(l.lrB[0]), [pA0] "s"(pA[0]), [pB0] "s"(pB[0]), [lrA1] "s"(l.lrA[1]),
                       [lrB1] "s"(l.lrB[1]), [pA1] "s"(pA[1]), [pB1] "s"(pB[1]), [lrA2] "s"(l.lrA[2]), [lrB2] "s"(l.lrB[2]),
                       [pA2] "s"(pA[2]), [pB2] "s"(pB[2]), [lrA3] "s"(l.lrA[3]), [lrB3] "s"(l.lrB[3]), [pA3] "s"(pA[3]),
                       [pB3] "s"(pB[3]), [l16] "v"(l16), [base] "s"(base)
                     : "memory");
#undef SRX_GRAM_LOAD
      }
        return l;
    };
    auto processP = [&](const LoadedP& l, auto fxc, auto after) {
        constexpr int kAfter = decltype(after)::value;
        if constexpr (decltype(fxc)::value) {
            // The two HALVES of the wave run the product and the LDS address with their own record's scalars as operands (exec = the
            // half's lanes whose entry exists: s_bfm_b64 on the count the record carries); they write disjoint lanes of the same two
            // registers, and the conversion and the LDS atomic are then issued ONCE for both halves — an LDS atomic costs the pipe its
            // issue whatever the number of active lanes (a first version with one atomic per half doubled SQ_INSTS_LDS and ran at
            // 2.97 ms against the compiler's 2.73: profiles/r05_knockouts.md).  v62 / v63: the 64-bit addend (the product's integer,
            // zero above it) — fixed registers because inline assembly cannot name the halves of a 64-bit operand.
#define SRX_GRAM_ENTRY(VAA, VAB, J, B, NA, NB)                                     \
            NA                                                                     \
            "s_mov_b64 exec, %[m]\n\t"                                             \
            "v_fma_f32 v62, %[" VAA "], %[" B "], 0.5\n\t"                         \
            "v_lshl_add_u32 %[a], %[" J "], 3, %[rowA]\n\t"                       \
            NB                                                                     \
            "v_fma_f32 v62, %[" VAB "], %[" B "], 0.5\n\t"                         \
            "v_lshl_add_u32 %[a], %[" J "], 3, %[rowB]\n\t"                       \
            "s_or_b64 exec, exec, %[m]\n\t"                                        \
            "v_cvt_u32_f32 v62, v62\n\t"                                          \
            "ds_add_u64 %[a], v[62:63]\n\t"
#define SRX_GRAM_LOADP(LRA, LRB, VAA, VAB, J0, B0, J1, B1)                          \
            "s_ashr_i32 %[rowA], %[" LRA "], 12\n\t"                               \
            "s_ashr_i32 %[rowB], %[" LRB "], 12\n\t"                               \
            SRX_GRAM_ENTRY(VAA, VAB, J0, B0,                                       \
                           "s_bfm_b64 %[m], %[" LRA "], 0\n\t",                    \
                           "s_bfm_b64 exec, %[" LRB "], 32\n\t")                   \
            SRX_GRAM_ENTRY(VAA, VAB, J1, B1,                                       \
                           "s_lshr_b32 %[t], %[" LRA "], 6\n\ts_bfm_b64 %[m], %[t], 0\n\t",   \
                           "s_lshr_b32 %[t], %[" LRB "], 6\n\ts_bfm_b64 exec, %[t], 32\n\t")
#pragma unroll
            for (int u = 0; u < kL; u += 2) {
                const unsigned j0 = l.raw[u].x, b0 = l.raw[u].y, j1 = l.raw[u].z, b1 = l.raw[u].w;
                const unsigned j2 = l.raw[u + 1].x, b2 = l.raw[u + 1].y, j3 = l.raw[u + 1].z, b3 = l.raw[u + 1].w;
                uint32_t a, t, rowA, rowB;
                unsigned long long m;
                asm volatile(
                    "v_mov_b32 v63, 0\n\t"
                    "s_waitcnt vmcnt(%[cnt0])\n\t"
                    SRX_GRAM_LOADP("lrA0", "lrB0", "vaA0", "vaB0", "j0", "b0", "j1", "b1")
                    "s_waitcnt vmcnt(%[cnt1])\n\t"
                    SRX_GRAM_LOADP("lrA1", "lrB1", "vaA1", "vaB1", "j2", "b2", "j3", "b3")
                    "s_mov_b64 exec, -1"
                    : [a] "=&v"(a), [t] "=&s"(t), [rowA] "=&s"(rowA), [rowB] "=&s"(rowB), [m] "=&s"(m)
                    : [lrA0] "s"(l.lrA[u]), [lrB0] "s"(l.lrB[u]), [vaA0] "s"(l.vaA[u]), [vaB0] "s"(l.vaB[u]), [lrA1] "s"(l.lrA[u + 1]),
                      [lrB1] "s"(l.lrB[u + 1]), [vaA1] "s"(l.vaA[u + 1]), [vaB1] "s"(l.vaB[u + 1]), [j0] "v"(j0), [b0] "v"(b0), [j1] "v"(j1),
                      [b1] "v"(b1), [j2] "v"(j2), [b2] "v"(b2), [j3] "v"(j3), [b3] "v"(b3), [cnt0] "n"(kAfter + kL - 1 - u),
                      [cnt1] "n"(kAfter + kL - 2 - u)
                    : "memory", "v62", "v63");
            }
#undef SRX_GRAM_LOADP
#undef SRX_GRAM_ENTRY
        } else {
            // f64 atomics (a negative value, a wide range): the per-lane form, products converted to f64
#pragma unroll
            for (int u = 0; u < kL; ++u) {
                u4 rw = l.raw[u];                    // (the wait names the registers it is for: nothing that reads them moves above it)
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rw) : "n"(kAfter + kL - 1 - u) : "memory");
                const uint32_t lr = hi ? l.lrB[u] : l.lrA[u];
                const VT va = __uint_as_float(hi ? l.vaB[u] : l.vaA[u]);
                char* const rowb = reinterpret_cast<char*>(acc) + ((int)lr >> 12);
                const uint32_t n0 = lr & 63u, n1 = (lr >> 6) & 63u, lh = (uint32_t)(lane & 31);
                const unsigned j0 = rw.x, b0 = rw.y, j1 = rw.z, b1 = rw.w;
                double* const accd = reinterpret_cast<double*>(rowb);
                if (lh < n0)
                    __hip_atomic_fetch_add(accd + (int)j0, gram_product(va, (VT)__uint_as_float(b0)), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                if (lh < n1)
                    __hip_atomic_fetch_add(accd + (int)j1, gram_product(va, (VT)__uint_as_float(b1)), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };
    auto consume = [&](const Slab& sl, Slab& other, auto fxc) {
        const int n = (int)sl.n;
        if constexpr (kPair) {
            using LoadsAfter = std::integral_constant<int, kL>;
            using NoneAfter = std::integral_constant<int, 0>;
            if constexpr (decltype(fxc)::value) {
                LoadedP A = batchP(sl, 0), B;
                other = next_slab();
#pragma unroll
                for (int u0 = 0; u0 < kWave; u0 += 2 * kUnroll) {
                    B = batchP(sl, u0 + kUnroll);
                    processP(A, fxc, LoadsAfter{});
                    if (u0 + 2 * kUnroll < kWave) {
                        A = batchP(sl, u0 + 2 * kUnroll);
                        processP(B, fxc, LoadsAfter{});
                    } else {
                        processP(B, fxc, NoneAfter{});
                    }
                    if (u0 + 2 * kUnroll >= n) break;
                }
            } else {
                // (the f64-atomics form of the f32 kernel — a negative value, a wide range — takes one set of loads at a time: it is the
                //  rare route, and two sets in flight beside the fixed-point route's cost the kernel a spilled register)
                other = next_slab();
                for (int u0 = 0; u0 < kWave && u0 < n; u0 += kUnroll) {
                    LoadedP A = batchP(sl, u0);
                    processP(A, fxc, NoneAfter{});
                }
            }
        } else {
            Loaded A = batch(sl, 0), B;
            other = next_slab();
#pragma unroll
            for (int u0 = 0; u0 < kWave; u0 += 2 * kUnroll) {
                // no branch around a batch's loads (slots past n are empty records, their loads hit the block's first line):
                // after a conditional load the compiler's wait for A's entries also waits for B's
                B = batch(sl, u0 + kUnroll);
                process(A, fxc);
                if (u0 + 2 * kUnroll < kWave) A = batch(sl, u0 + 2 * kUnroll);
                process(B, fxc);
                if (u0 + 2 * kUnroll >= n) break;
            }
        }
    };
    auto run = [&](auto fxc) {
        Slab S = next_slab(), T;
        T.n = 0;
        while (S.n > 0) {
            consume(S, T, fxc);
            if (T.n == 0) break;
            consume(T, S, fxc);
        }
    };
    if (fx) run(std::true_type{});
    else run(std::false_type{});
    // (the assembly core's ds_add_u64 are invisible to the compiler's wait counting: drained explicitly before the barrier the flush
    //  reads the accumulators behind — the compiler's own release fence put a wait here too, but nothing obliged it to.  ADVICE r5)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // flush: the upper-triangle part of both stripes, added to the packed matrix (row splits and, in backed
    // mode, earlier row tiles have been there before)
    for (int e = threadIdx.x; e < SR * WA; e += blockDim.x) {
        const int r = e / WA, c = a0 + e % WA, row = a0 + r;
        const double v = fx ? (double)(unsigned long long)__double_as_longlong(acc[e]) * fx_inv : acc[e];
        if (row < k && c >= row && v != 0.0) atomicAdd(&Gp[tri_index(row, c, k)], v);
    }
    for (int e = threadIdx.x; e < SR * WB; e += blockDim.x) {
        const int r = e / WB, c = b0 + e % WB, row = b0 + r;
        const double v = fx ? (double)(unsigned long long)__double_as_longlong(acc[SR * WA + e]) * fx_inv : acc[SR * WA + e];
        if (row < k && c >= row && v != 0.0) atomicAdd(&Gp[tri_index(row, c, k)], v);
    }
}

