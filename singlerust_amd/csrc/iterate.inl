// iterate.inl — included by pca_solve.hip inside namespace srx (one translation unit: the kernels share its helpers and constants).
// Kernels of the k x 64 subspace iteration: dense application of C, CholeskyQR, Rayleigh-Ritz (jacobi.inl), Chebyshev filter, result assembly.

// C (k x k, both triangles) from the packed upper triangle: (i, j) and (j, i) read the same entry, so C is
// EXACTLY symmetric (k_dense_apply reads it transposed).  With d != nullptr: C = D (G - cen N mu mu^T) D.
__global__ void k_gram_expand(const double* __restrict__ P, int k, const double* __restrict__ d,
                              const double* __restrict__ mu, int cen, double n_cells, double* __restrict__ C) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)k * k) return;
    const int i = (int)(e / k), j = (int)(e % k);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    double g = P[tri_index(lo, hi, k)];
    if (d) {
        if (cen) g -= n_cells * (mu[i] * mu[j]);            // (mu_i mu_j) first: symmetric to the last bit
        g = d[i] * d[j] * g;
    }
    C[e] = g;
}

// Wp += C[:, krange] W[krange, :] for the dense SYMMETRIC k x k matrix C and a k x 64 block (f64), on the
// f64 matrix cores.  One wave = 32 output rows x 64 columns x one K slice: eight v_mfma_f64_16x16x4
// accumulators.  Both operands are read straight from global memory in fragment order with no LDS
// staging: lane l of the A fragment needs C[row0 + (l & 15)][kk + (l >> 4)], which by symmetry is
// C[kk + (l >> 4)][row0 + (l & 15)] — 16 consecutive doubles per K index, fully coalesced; the B
// fragment W[kk + (l >> 4)][16 t + (l & 15)] is coalesced as it stands.  The K slices (split-K 16 across
// workgroups x 4 waves inside one: ~4000 waves for k = 2000) are combined in LDS, then with f64 atomics into the zeroed Wp.
// C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg (not the f32 map).
// measured at k = 2000 (split x waves: us): 16x4 21.8, 8x8 21.9, 8x4 18.8, 4x4 17.8, 4x8 17.5, 8x2 28.3, 4x16 36.3 — the f64
// atomics into Wp (k x 64 x split) weigh more than the number of waves in flight
constexpr int kDenseSplit = 4;
constexpr int kDenseWaves = 8;                           // waves of a workgroup: consecutive quarters of the workgroup's K slice
typedef double dvec4 __attribute__((ext_vector_type(4)));
// Workgroup = 32 output rows x 64 columns x one K slice, its four waves on consecutive quarters of the slice (four waves per
// SIMD keep ~4x the loads in flight: one wave per SIMD left the load latency of every group of 16 K indices exposed, 27 us
// per application against ~7 us of MFMA time); the waves' partial tiles meet in LDS (ds_add_f64), then one f64 atomic per
// output element and K slice into the zeroed Wp.
// (Round 5: the waves' partial tiles used to meet in LDS through ds_add_f64 — 32 atomic instructions per wave at ~25 clocks each,
//  2.7 us of an 18 us launch.  Now every wave STORES its tile in accumulator-register order ([wave][register][lane]: conflict-free,
//  4 clocks a store), and thread t adds the eight partials of the four (register, lane) slots it is given and sends them out.)
constexpr size_t kDenseLds = (size_t)kDenseWaves * 32 * 64 * sizeof(double);      // 128 KiB, dynamic: one workgroup per CU (252 of them)
__global__ __launch_bounds__(kDenseWaves * 64) void k_dense_apply(const double* __restrict__ C, const double* __restrict__ W, int k,
                                                                  double* __restrict__ Wp) {
    extern __shared__ double red[];                // [wave][acc register 0 .. 31][lane]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int row0 = blockIdx.x * 32;
    const int kchunk = (((k + kDenseSplit - 1) / kDenseSplit) + 4 * kDenseWaves - 1) / (4 * kDenseWaves) * (4 * kDenseWaves);      // per workgroup: waves x a multiple of 4
    const int kq = kchunk / kDenseWaves;
    const int kbeg = blockIdx.y * kchunk + wv * kq;
    const int kend = kbeg + kq < k ? kbeg + kq : k;
    dvec4 acc[2][4];
#pragma unroll
    for (int sI = 0; sI < 2; ++sI)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[sI][t] = dvec4{0.0, 0.0, 0.0, 0.0};
    const bool r0ok = row0 + li < k, r1ok = row0 + 16 + li < k;
    // groups of 4 K-steps (16 K indices), two register buffers: the 24 loads of group g+1 are in flight
    // while the 32 MFMAs of group g issue
    struct Frag { double a0[4], a1[4], bq[4][4]; };
    auto load = [&](Frag& f, int kk) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kr = kk + 4 * u + lk;
            const bool kok = kr < kend;
            const double* crow = C + (size_t)(kok ? kr : 0) * k + row0 + li;
            const double* wrow = W + (size_t)(kok ? kr : 0) * L + li;
            f.a0[u] = (kok && r0ok) ? crow[0] : 0.0;
            f.a1[u] = (kok && r1ok) ? crow[16] : 0.0;
#pragma unroll
            for (int t = 0; t < 4; ++t) f.bq[u][t] = kok ? wrow[16 * t] : 0.0;
        }
    };
    auto fma = [&](const Frag& f) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a0[u], f.bq[u][t], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a1[u], f.bq[u][t], acc[1][t], 0, 0, 0);
            }
    };
    Frag f0, f1;
    load(f0, kbeg);
    for (int kk = kbeg; kk < kend; kk += 32) {
        load(f1, kk + 16);
        fma(f0);
        load(f0, kk + 32);
        fma(f1);
    }
    // register index q = (sI * 4 + t) * 4 + v; C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * v
#pragma unroll
    for (int sI = 0; sI < 2; ++sI)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) red[((size_t)wv * 32 + (sI * 4 + t) * 4 + v) * 64 + lane] = acc[sI][t][v];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = wv + kDenseWaves * i;          // 512 threads x 4 slots = the 32 registers x 64 lanes of a tile
        double part[kDenseWaves];
#pragma unroll
        for (int w2 = 0; w2 < kDenseWaves; ++w2) part[w2] = red[((size_t)w2 * 32 + q) * 64 + lane];
        double sum = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < kDenseWaves; ++w2) sum += part[w2];
        const int sI = q >> 4, t = (q >> 2) & 3, v = q & 3;
        const int r = row0 + 16 * sI + lk + 4 * v;
        if (r < k) atomicAdd(&Wp[(size_t)r * L + 16 * t + li], sum);
    }
}


// ---- k x l helper kernels (f64, replicated per rank) --------------------------------------------
// P = PT(d .* W * sign), cvec = cen * mu^T P (exact f64 sum of the ROUNDED panel, so Y's column
// sums vanish to rounding).
template <typename PT>
__global__ __launch_bounds__(1024) void k_make_panel(const double* __restrict__ W, const double* __restrict__ d,
                                                     const double* __restrict__ mu, const double* __restrict__ sgn,
                                                     int k, int cen, PT* __restrict__ P, PT* __restrict__ cvec) {
    __shared__ double s_part[16][L];
    const int c = threadIdx.x & (L - 1), part = threadIdx.x / L;   // 16 row slices x 64 columns
    double acc = 0.0;
    const double sg = sgn ? sgn[c] : 1.0;
    for (int j = part; j < k; j += 16) {
        PT p = (PT)(d[j] * W[(size_t)j * L + c] * sg);
        P[(size_t)j * L + c] = p;
        acc += mu[j] * (double)p;
    }
    s_part[part][c] = acc;
    __syncthreads();
    if (part == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += s_part[w][c];
        cvec[c] = cen ? (PT)t : PT(0);
    }
}

// The same for the END of a round, on many workgroups and with the round's bookkeeping folded in (one workgroup walking the
// k x 64 block took 48 us, and three device-to-device copies and k_signs followed it): sign of each Ritz vector from its
// largest entry, P = PT(d .* V * sign), per-block partial sums of mu^T P (k_cvec_reduce adds them in fixed order), and the
// copy of the Ritz vectors, values and signs into the matrix's result block.
constexpr int kPanelBlocks = 32;
template <typename PT>
__global__ __launch_bounds__(1024) void k_make_panel_mb(const double* __restrict__ W, const double* __restrict__ d,
                                                        const double* __restrict__ mu, const double* __restrict__ colmax,
                                                        const double* __restrict__ theta, int k, PT* __restrict__ P,
                                                        double* __restrict__ part /* [blocks][64] */, double* __restrict__ sgn_out,
                                                        double* __restrict__ blk /* k*64 vectors | 64 values | 64 signs */) {
    __shared__ double s_part[16][L];
    const int c = threadIdx.x & (L - 1), sl = threadIdx.x / L;   // 16 row slices x 64 columns
    const double sg = colmax[c] < 0 ? -1.0 : 1.0;
    if (blockIdx.x == 0 && sl == 0) {
        sgn_out[c] = sg;
        blk[(size_t)k * L + c] = theta[c];
        blk[(size_t)k * L + L + c] = sg;
    }
    double acc = 0.0;
    for (int j = blockIdx.x * 16 + sl; j < k; j += gridDim.x * 16) {
        const double wv = W[(size_t)j * L + c];
        blk[(size_t)j * L + c] = wv;
        const PT p = (PT)(d[j] * wv * sg);
        P[(size_t)j * L + c] = p;
        acc += mu[j] * (double)p;
    }
    s_part[sl][c] = acc;
    __syncthreads();
    if (sl == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += s_part[w][c];
        part[(size_t)blockIdx.x * L + c] = t;
    }
}
template <typename PT>
__global__ void k_cvec_reduce(const double* __restrict__ part, int n_blocks, int cen, PT* __restrict__ cvec) {
    const int c = threadIdx.x;
    double t = 0.0;
    for (int b = 0; b < n_blocks; ++b) t += part[(size_t)b * L + c];
    cvec[c] = cen ? (PT)t : PT(0);
}

// up to four small device-to-device copies in one launch (32-bit words)
struct CopySegs {
    const uint32_t* src[4];
    uint32_t* dst[4];
    uint32_t words[4];
};
__global__ void k_copy_segs(CopySegs sg) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        for (uint32_t e = t; e < sg.words[i]; e += stride) sg.dst[i][e] = sg.src[i][e];
}

// W' = d .* (T - cen * mu s^T)
__global__ void k_finish_t(const double* __restrict__ T, const double* __restrict__ d, const double* __restrict__ mu,
                           int k, int cen, double* __restrict__ Wp) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)k * L) return;
    int j = (int)(e / L), c = (int)(e % L);
    double s = T[(size_t)k * L + c];
    Wp[e] = d[j] * (T[e] - (cen ? mu[j] * s : 0.0));
}

// H = A^T B, G = B^T B for two k x 64 blocks (f64).  Each workgroup reduces a slice of the k rows
// (staged through LDS) into a partial 64 x 64 pair; k_gram2_reduce sums the slices in fixed order.
constexpr int kGram2Blocks = 64;
__global__ __launch_bounds__(1024) void k_gram2_part(const double* __restrict__ A, const double* __restrict__ B, int k,
                                                     double* __restrict__ part /* [blocks][2][64*64] */) {
    constexpr int R = 32;
    __shared__ double sa[R][L], sb[R][L];
    double h[4] = {0, 0, 0, 0}, g[4] = {0, 0, 0, 0};
    const int b = threadIdx.x & (L - 1), a0 = threadIdx.x / L;    // entries (a0 + 16u, b), u < 4
    const int rows_per = (k + gridDim.x - 1) / gridDim.x;
    const int jb0 = blockIdx.x * rows_per;
    const int jb1 = jb0 + rows_per < k ? jb0 + rows_per : k;
    for (int j0 = jb0; j0 < jb1; j0 += R) {
        for (int e = threadIdx.x; e < R * L; e += 1024) {
            int j = j0 + e / L;
            sa[e / L][e % L] = j < jb1 ? A[(size_t)j * L + (e % L)] : 0.0;
            sb[e / L][e % L] = j < jb1 ? B[(size_t)j * L + (e % L)] : 0.0;
        }
        __syncthreads();
#pragma unroll 4
        for (int r = 0; r < R; ++r) {
            double bv = sb[r][b];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                h[u] += sa[r][a0 + 16 * u] * bv;
                g[u] += sb[r][a0 + 16 * u] * bv;
            }
        }
        __syncthreads();
    }
    double* out = part + (size_t)blockIdx.x * 2 * L * L;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        out[(a0 + 16 * u) * L + b] = h[u];
        out[L * L + (a0 + 16 * u) * L + b] = g[u];
    }
}
__global__ void k_gram2_reduce(const double* __restrict__ part, int n_blocks, double* __restrict__ HG /* 2*64*64 */) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * L * L) return;
    double s = 0.0;
    for (int b = 0; b < n_blocks; ++b) s += part[(size_t)b * 2 * L * L + e];
    HG[e] = s;
}

// ONE product of two k x 64 blocks, H = A^T B (A == B: the Gram matrix of a block): kGram1Blocks workgroups, one row slice each,
// their partial 64 x 64 sums ADDED into one zeroed matrix with global f64 atomics (round 5; until then 16 partial matrices that the
// consumer summed on load — 512 KB through the one CU the Cholesky / Jacobi workgroup runs on: ~6 us of the factorisation's 34 and
// ~15 of the eigen-solve's 190, the 384 threads of the latter taking 11 trips of 16 loads).  The consumer reads H once and leaves it
// ZEROED for the next product; half the arithmetic of k_gram2_part, which forms both products whichever is wanted.
constexpr int kGram1Blocks = 16;
// On the f64 matrix cores: wave w of a workgroup owns the 16 x 16 output tile (w / 4, w % 4); both operands come straight
// from global memory in fragment order (lane l: row kk + (l >> 4), column 16 t + (l & 15) — 128 contiguous bytes per
// 16 lanes), eight K-steps of loads in flight.  (The LDS-staged scalar version was LDS-read bound: 24 us a launch.)
__global__ __launch_bounds__(1024) void k_gram1_part(const double* __restrict__ A, const double* __restrict__ B, int k,
                                                     double* __restrict__ H /* 64 x 64, accumulated into */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int ti = wv >> 2, tj = wv & 3;
    const int rows_per = (k + gridDim.x - 1) / gridDim.x;
    const int jb0 = blockIdx.x * rows_per;
    const int jb1 = jb0 + rows_per < k ? jb0 + rows_per : k;
    dvec4 acc = dvec4{0.0, 0.0, 0.0, 0.0};
    constexpr int kU = 8;
    for (int j0 = jb0; j0 < jb1; j0 += 4 * kU) {
        double av[kU], bv[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int r = j0 + 4 * u + lk;
            const bool ok = r < jb1;
            av[u] = ok ? A[(size_t)r * L + 16 * ti + li] : 0.0;
            bv[u] = ok ? B[(size_t)r * L + 16 * tj + li] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
    }
    // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int v = 0; v < 4; ++v) atomicAdd(&H[(16 * ti + lk + 4 * v) * L + 16 * tj + li], acc[v]);
}

// The tail of a Rayleigh–Ritz step in one pass over the rows: A1 = Wp U (= C W U), A2 = W U (the Ritz vectors), and per
// block the partial sums of || A1[:, c] - theta_c A2[:, c] ||^2 and the entry of largest |.| of A2[:, c] (ties: the smallest
// row).  k_resid_final adds the partials in fixed order.  (Round 2: two k_right_mul launches, the residuals on ONE workgroup, a scalar kernel.)
constexpr int kRitzBlocks = 128;
// `zero_wp`: the rows of Wp are left ZEROED once read (Wp is the destination of the next application of C).
__global__ __launch_bounds__(256) void k_ritz_post(const double* __restrict__ W, double* __restrict__ Wp,
                                                   const double* __restrict__ M, const double* __restrict__ theta, int k,
                                                   double* __restrict__ A1, double* __restrict__ A2,
                                                   double* __restrict__ part /* [blocks][3][64]: r2, best value, its row */, int zero_wp) {
    __shared__ double sm[L][L + 1];
    __shared__ double s_r[4][L], s_v[4][L], s_j[4][L];
    for (int e = threadIdx.x; e < L * L; e += 256) sm[e / L][e % L] = M[e];
    __syncthreads();
    const int c = threadIdx.x & (L - 1), sub = threadIdx.x / L;    // 4 rows per pass
    const double th = theta[c];
    double acc = 0.0, best = 0.0, best_j = 0.0;
    for (int j = blockIdx.x * 4 + sub; j < k; j += gridDim.x * 4) {
        const double* rw = W + (size_t)j * L;
        double* rp = Wp + (size_t)j * L;
        double a1 = 0.0, a2 = 0.0;
#pragma unroll 8
        for (int b = 0; b < L; ++b) {
            const double m = sm[b][c];
            a1 += rp[b] * m;
            a2 += rw[b] * m;
        }
        if (zero_wp) {                           // (the 64 threads of a row are the lanes of ONE wave: all have read the row)
            asm volatile("" ::: "memory");
            rp[c] = 0.0;
        }
        A1[(size_t)j * L + c] = a1;
        A2[(size_t)j * L + c] = a2;
        const double r = a1 - th * a2;
        acc += r * r;
        if (fabs(a2) > fabs(best)) {          // rows come in increasing order: the first one of the largest magnitude stays
            best = a2;
            best_j = (double)j;
        }
    }
    s_r[sub][c] = acc;
    s_v[sub][c] = best;
    s_j[sub][c] = best_j;
    __syncthreads();
    if (sub == 0) {
        double t = 0.0, bv = 0.0, bj = 0.0;
        for (int w = 0; w < 4; ++w) {
            t += s_r[w][c];
            const double v = s_v[w][c], jj = s_j[w][c];
            if (fabs(v) > fabs(bv) || (fabs(v) == fabs(bv) && v != 0.0 && jj < bj)) {
                bv = v;
                bj = jj;
            }
        }
        double* out = part + (size_t)blockIdx.x * 3 * L;
        out[c] = t;
        out[L + c] = bv;
        out[2 * L + c] = bj;
    }
}

// Out = In * M  (k x 64 times 64 x 64), M row-major.
__global__ __launch_bounds__(256) void k_right_mul(const double* __restrict__ In, const double* __restrict__ M, int k,
                                                   double* __restrict__ Out) {
    __shared__ double sm[L][L + 1];
    for (int e = threadIdx.x; e < L * L; e += 256) sm[e / L][e % L] = M[e];
    __syncthreads();
    const int c = threadIdx.x & (L - 1), sub = threadIdx.x / L;    // 4 rows per pass
    for (int j = blockIdx.x * 4 + sub; j < k; j += gridDim.x * 4) {
        const double* row = In + (size_t)j * L;
        double acc = 0.0;
#pragma unroll 8
        for (int b = 0; b < L; ++b) acc += row[b] * sm[b][c];
        Out[(size_t)j * L + c] = acc;
    }
}

constexpr int kStatChol = 1, kStatEig = 2;

// Start block: counter-based N(0,1) entries, deterministic in (seed, gene slot, column).
__device__ __forceinline__ uint64_t dmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void k_identity_block(int k, double* __restrict__ W) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < k * L) W[e] = (e / L == e % L) ? 1.0 : 0.0;
}
// (z0 .. z2, nullable: blocks of the same shape left ZEROED — the destinations of the warm-up's applications of C, which
//  accumulate into zeroed blocks: no memset launch in front of any of them)
__global__ void k_init_block(uint64_t seed, int k, int l_act, double* __restrict__ Wp, double* __restrict__ z0, double* __restrict__ z1,
                             double* __restrict__ z2, uint32_t* __restrict__ zw, int n_zw /* words zeroed besides (the solver's status block) */) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = e; i < n_zw; i += gridDim.x * blockDim.x) zw[i] = 0u;
    if (e >= k * L) return;
    if (z0) z0[e] = 0.0;
    if (z1) z1[e] = 0.0;
    if (z2) z2[e] = 0.0;
    const int j = e / L, cc = e % L;
    double v = 0.0;
    if (cc < l_act) {
        const uint64_t base = dmix64(seed ^ dmix64((uint64_t)j));
        const uint64_t h1 = dmix64(base + 2 * (uint64_t)cc), h2 = dmix64(base + 2 * (uint64_t)cc + 1);
        const double u1 = ((double)(h1 >> 11) + 0.5) / 9007199254740992.0;
        const double u2 = ((double)(h2 >> 11) + 0.5) / 9007199254740992.0;
        v = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    }
    Wp[e] = v;
}

// CholeskyQR, device side: G = R^T R (upper Cholesky of the leading n x n block of G, ld = L), then
// W = Wp R^-1 by one forward substitution per ROW of Wp (k independent rows) — no explicit inverse.
//
// k_chol_factor: one workgroup, right-looking: at step j every thread subtracts a_jr a_jc / a_jj from the
// trailing elements it owns (one barrier per step).  Rout (L x L, row-major) receives R, zero outside
// the upper triangle of the leading block; dinv[j] = 1 / R[j][j] (0 for j >= n).
// `shifted` (the robust mode of the driver): a pivot that has fallen below 1e-13 of the largest diagonal entry of G is
// held at that floor instead of being reported — the factor then belongs to a slightly shifted G, W = Wp R^-1 stays
// bounded and of full rank, and a second plain pass (CholeskyQR2) makes it orthonormal (shifted CholeskyQR3 idea).
__global__ __launch_bounds__(1024) void k_chol_factor(const double* __restrict__ G, int n, double* __restrict__ Rout,
                                                      double* __restrict__ dinv, int* __restrict__ status, int shifted) {
    __shared__ double A[L][L + 1];
    __shared__ double s_floor, s_diag[L];
    const int tid = threadIdx.x;
    for (int e = tid; e < L * L; e += 1024) {
        const int r = e >> 6, c = e & 63;
        A[r][c] = (r < n && c < n) ? G[(size_t)r * L + c] : 0.0;
    }
    __syncthreads();
    if (tid < L) s_diag[tid] = A[tid][tid];
    if (tid == 0) {
        double mx = 0.0;
        for (int j = 0; j < n; ++j) mx = A[j][j] > mx ? A[j][j] : mx;
        s_floor = shifted ? 1e-13 * mx : 0.0;
    }
    __syncthreads();
    bool bad = false;
    for (int j = 0; j < n; ++j) {
        double d = A[j][j];
        // shifted (last-resort) mode: a pivot below 1e-13 of the largest diagonal entry means the column depends on the
        // ones before it — the block is wider than the numerical rank of the data (a handful of cells, most selected
        // columns empty).  The column is DROPPED (zero in Q: dinv = 0, empty row of R) instead of being scaled up from
        // rounding noise; it stays zero under C, and its Ritz pair comes out as (0, 0).  "Dependent" = the pivot is
        // below 1e-13 of the column's OWN squared norm (s_diag, taken before the elimination).  (A zero or NaN matrix
        // keeps its non-positive pivot: reported below.)
        const bool drop = shifted && s_floor > 0.0 && d == d && !(d > 1e-13 * s_diag[j]);
        if (!drop && shifted && !(d > s_floor)) d = s_floor;   // independent but tiny next to the others: lifted as before
        if (!drop && !(d > 0.0)) bad = true;
        const double inv = drop ? 0.0 : rsqrt(d), inv2 = inv * inv;
        if (tid < L) {
            Rout[(size_t)j * L + tid] = (tid >= j && tid < n) ? (drop ? (tid == j ? 1.0 : 0.0) : A[j][tid] * inv) : 0.0;
            if (tid == j) dinv[j] = inv;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 1024 * u, r = e >> 6, c = e & 63;
            if (r > j && c >= r && c < n) A[r][c] -= A[j][r] * A[j][c] * inv2;
        }
        __syncthreads();
    }
    for (int e = tid; e < L * L; e += 1024)
        if ((e >> 6) >= n) Rout[e] = 0.0;
    if (tid < L && tid >= n) dinv[tid] = 0.0;
    if (bad && tid == 0) atomicOr(status, kStatChol);
}

// The plain factorisation (no shift, no dropped columns) in panels of kCholPanel rows: wave 0 factors a panel on its own, in
// registers — eight dependent steps of (v_readlane, rsqrt, multiply-subtract), no LDS round trip and no barrier — then all
// threads subtract the panel's rank-8 update from the trailing rows: 16 barriers for l = 64 instead of 64 (the CholeskyQR
// runs five times per solve).  Same outputs and status as k_chol_factor(shifted = 0).
// 1 / sqrt(x), normal positive x: the hardware seed (2^-24, bench_micro/rsq_precision.hip) and one third-order correction
__device__ __forceinline__ double fast_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-x * y, y, 1.0);
    return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}
constexpr int kCholPanel = 8;
// Round 5: the trailing update on the f64 matrix cores and the inverted 16 x 16 diagonal blocks of R as a second output
// (k_trsm_mfma's operands).  Round 3's scalar update (k_chol_factor_panels, gone) read two LDS words per multiply-add — ~900 wave-level LDS reads
// per panel, 2 us of the 4.5 us a panel took (36 us a launch, three launches per solve); as 16 x 16 tiles, D -= P_K^T P_L with the
// panel's 8 rows as the contraction index (two v_mfma_f64_16x16x4 per tile, <= 10 tiles on as many waves), a lane reads 2 + 2
// operands and moves its 4 tile entries in and out.  Rows of a tile that are finished (< j1) get a zero A operand: untouched.
// Xi[J][a][b] = (R_JJ^-1)[a][b], upper triangular, by back substitution (one thread per column; dinv = 0 — a column >= n —
// gives a zero column).
__global__ __launch_bounds__(1024) void k_chol_factor_mfma(double* __restrict__ G /* k_gram1_part's sum: read once, left zeroed */, int n,
                                                           double* __restrict__ Rout, double* __restrict__ dinv,
                                                           double* __restrict__ Xi_out, int* __restrict__ status) {
    __shared__ double A[L][L + 1];           // the rows of a finished panel hold R
    __shared__ double s_dinv[L];
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    {
        double g4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) g4[u] = G[tid + 1024 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 1024 * u, r = e >> 6, c = e & 63;
            A[r][c] = (r < n && c < n) ? g4[u] : 0.0;
            G[e] = 0.0;
        }
    }
    if (tid == 0) s_bad = 0;
    if (tid < L) s_dinv[tid] = 0.0;
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += kCholPanel) {
        const int j1 = j0 + kCholPanel < n ? j0 + kCholPanel : n;
        if (tid < kWave) {
            // (as in k_chol_factor_panels: lane c holds column c of the panel's rows in registers; pivots and multipliers by v_readlane)
            const int c = tid;
            double a[kCholPanel];
#pragma unroll
            for (int jj = 0; jj < kCholPanel; ++jj) a[jj] = j0 + jj < j1 ? A[j0 + jj][c] : 0.0;
#pragma unroll
            for (int jj = 0; jj < kCholPanel; ++jj) {
                const int j = j0 + jj;
                if (j < j1) {                                   // (uniform)
                    const double d = readlane_v(a[jj], j);
                    if (!(d > 0.0) && c == 0) s_bad = 1;
                    const double inv = d > 1e-290 ? fast_rsqrt(d) : rsqrt(d);
                    const double rjc = (c >= j && c < n) ? a[jj] * inv : 0.0;
                    a[jj] = rjc;                                // row j of R (zero left of the diagonal and right of n)
                    if (c == j) s_dinv[j] = inv;
#pragma unroll
                    for (int rr = jj + 1; rr < kCholPanel; ++rr) {
                        const int r = j0 + rr;
                        if (r < j1) {
                            const double rjr = readlane_v(rjc, r);
                            if (c >= r && c < n) a[rr] -= rjr * rjc;
                        }
                    }
                }
            }
#pragma unroll
            for (int jj = 0; jj < kCholPanel; ++jj)
                if (j0 + jj < j1) A[j0 + jj][c] = a[jj];      // (R and dinv go out once, at the end: a global store in front of a
                                                                //  barrier is a round trip to L2 the whole workgroup waits for — 8 panels of it)
        }
        __syncthreads();
        // trailing tiles (K, Lt), K0 <= K <= Lt <= 3, K0 = the tile that holds row j1: wave w takes the w-th of them
        if (j1 < L) {
            const int K0 = j1 >> 4;
            int K = -1, Lt = -1, t = wv;
            for (int kk = K0; kk < 4 && K < 0; ++kk) {
                const int cnt = 4 - kk;
                if (t < cnt) {
                    K = kk;
                    Lt = kk + t;
                } else t -= cnt;
            }
            if (K >= 0) {                                       // (wave-uniform)
                const int m = lane & 15, g = lane >> 4;
                dvec4 d;
#pragma unroll
                for (int v = 0; v < 4; ++v) d[v] = A[16 * K + g + 4 * v][16 * Lt + m];
#pragma unroll
                for (int sI = 0; sI < kCholPanel / 4; ++sI) {
                    const int j = j0 + 4 * sI + g;
                    const bool live = j < j1;
                    const double a = (live && 16 * K + m >= j1) ? A[live ? j : 0][16 * K + m] : 0.0;
                    const double b = live ? A[j][16 * Lt + m] : 0.0;
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(-a, b, d, 0, 0, 0);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) A[16 * K + g + 4 * v][16 * Lt + m] = d[v];
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < L * L; e += 1024) {
        const int r = e >> 6, c = e & 63;
        Rout[e] = (r < n && c >= r) ? A[r][c] : 0.0;          // (a finished row is zero right of n; below the diagonal the tiles hold leftovers)
    }
    if (tid < L) dinv[tid] = tid < n ? s_dinv[tid] : 0.0;
    if (tid == 0 && s_bad) atomicOr(status, kStatChol);
    if (tid < L) {                                             // the four inverted diagonal blocks
        const int J = tid >> 4, c = tid & 15, o = 16 * J;
        double x[16];
#pragma unroll
        for (int i = 15; i >= 0; --i) {
            double acc = 0.0;
#pragma unroll
            for (int mm = i + 1; mm < 16; ++mm) acc = __builtin_fma(A[o + i][o + mm], x[mm], acc);      // (x[mm] = 0 for mm > c)
            x[i] = i == c ? s_dinv[o + i] : (i < c ? -acc * s_dinv[o + i] : 0.0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) Xi_out[(J * 16 + i) * 16 + c] = x[i];
    }
}

// W[row] R = Wp[row]: w_j = (wp_j - sum_{i<j} w_i R[i][j]) / R[j][j], one thread per row, the row in
// registers (the j / i loops are fully unrolled: static register indices), R transposed in LDS so that
// the i-loop of a column reads consecutive words (wave-uniform addresses: broadcast, no conflicts).
// One wave per workgroup: k / 64 workgroups spread over as many compute units.
// (Round 4, measured and not kept: the RIGHT-LOOKING form — w_j = wp_j / R[j][j], then wp_c -= w_j R[j][c], 63 - j independent
//  multiply-subtracts per step — with R[j][c] as wave-uniform scalar loads at compile-time offsets: 33.6 us against 32.3, 153
//  exposed scalar-cache round trips on the one wave of a SIMD; with the rows of R prefetched from LDS into a second register
//  buffer the compiler hoists every row's reads to the top and spills 14 KB.)
// z0 .. z2 (nullable): blocks whose rows this thread leaves ZEROED once its source row is in registers — the source itself and
// the other scratch blocks of the sweep, i.e. the destinations of the next applications of C (accumulated into zeroed blocks:
// eight memset launches per solve gone); a thread owns its row in every block, so the source may be one of them (or W itself).
__global__ __launch_bounds__(64) void k_trsm_rows(const double* Wp, const double* __restrict__ R,
                                                  const double* __restrict__ dinv, int k, double* W, double* z0, double* z1, double* z2) {
    __shared__ double Rt[L][L];          // Rt[j][i] = R[i][j]
    __shared__ double di[L];
    for (int e = threadIdx.x; e < L * L; e += 64) Rt[e & 63][e >> 6] = R[e];
    di[threadIdx.x] = dinv[threadIdx.x];
    __syncthreads();
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= k) return;
    double w[L];
    const double* src = Wp + (size_t)row * L;
#pragma unroll
    for (int j = 0; j < L; ++j) w[j] = src[j];
    asm volatile("" ::: "memory");                 // the row is read before anything of it is overwritten
    {
        typedef double d2z __attribute__((ext_vector_type(2)));
        double* const zs[3] = {z0, z1, z2};
#pragma unroll
        for (int b = 0; b < 3; ++b)
            if (zs[b] && zs[b] != W) {
                d2z* zr = reinterpret_cast<d2z*>(zs[b] + (size_t)row * L);
#pragma unroll
                for (int j = 0; j < L / 2; ++j) zr[j] = d2z{0.0, 0.0};
            }
    }
#pragma unroll
    for (int j = 0; j < L; ++j) {
        double acc0 = w[j], acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;      // (eight chains instead of four: 32 -> 56 us, the row spills)
#pragma unroll
        for (int i = 0; i + 3 < j; i += 4) {
            acc0 -= w[i] * Rt[j][i];
            acc1 -= w[i + 1] * Rt[j][i + 1];
            acc2 -= w[i + 2] * Rt[j][i + 2];
            acc3 -= w[i + 3] * Rt[j][i + 3];
        }
#pragma unroll
        for (int i = j & ~3; i < j; ++i) acc0 -= w[i] * Rt[j][i];
        w[j] = ((acc0 + acc1) + (acc2 + acc3)) * di[j];
    }
    double* dst = W + (size_t)row * L;
#pragma unroll
    for (int j = 0; j < L; ++j) dst[j] = w[j];
}

// The same substitution on the f64 matrix cores, for the plain CholeskyQR (round 5; k_trsm_rows — one thread per row, a
// 2016-step dependent chain with an LDS read per step — took 33 us a launch, three launches per solve).  Blocked by 16 columns
// and TRANSPOSED, X = W^T:   X_J = R_JJ^-T (Wp_J^T - sum_{I<J} R_IJ^T X_I),   J = 0 .. 3,
// a wave owning 16 rows of W (= 16 columns of X).  Every product is D(16x16) += A(16x4) B(4x16) on v_mfma_f64_16x16x4 with
// the contraction index taken in the order the accumulator registers hold it: a finished tile X_I sits in the C/D layout
// (lane: column l & 15, rows (l >> 4) + 4 reg), which IS the B operand of step `reg` when the A operand supplies
// R[16 I + (l >> 4) + 4 reg][16 J + (l & 15)] — so the tiles never leave the registers (no LDS relayout between the stages),
// and R and the four inverted diagonal blocks are read straight from global memory, all loads in flight at once (no LDS, no
// barrier: 18 -> see profiles/r05_rocprof_c3.md).  Those inverses (16 x 16 upper triangular, by back substitution, one thread per
// column) are k_chol_factor_mfma's second output.
// A forward substitution through explicitly inverted DIAGONAL BLOCKS is what blocked TRSM implementations do; it is backward
// stable up to the condition of a 16 x 16 diagonal block.  Columns >= n (R's rows zero, dinv = 0) come out zero, as from
// k_trsm_rows; the robust mode (dropped columns: dinv = 0 with a unit row in R) keeps k_trsm_rows.
// z0 .. z2: as in k_trsm_rows (the rows of those blocks are left zeroed once the wave's own 16 source rows are in registers).
constexpr int kTrsmTiles = 4;              // waves (16-row tiles) per workgroup
__global__ __launch_bounds__(kTrsmTiles * 64) void k_trsm_mfma(const double* Wp, const double* __restrict__ R,
                                                               const double* __restrict__ Xi /* k_chol_factor_mfma: [4][16][16] */,
                                                               int k, double* W, double* z0, double* z1, double* z2) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int row = (blockIdx.x * kTrsmTiles + wv) * 16 + m;
    const bool ok = row < k;
    const size_t base = (size_t)(ok ? row : 0) * L;
    // every operand is asked for up front (the tile's 16 entries per lane, the 24 entries of R's six blocks above the diagonal,
    // the 16 of the inverted diagonal blocks — R and Xi are 34 KB, L2-resident): one round trip, then the chain of 40 MFMAs
    dvec4 T[4], X[4];
    double ra[4][4][4], xa[4][4];           // ra[I][J][r] (I < J): -R[16 I + g + 4 r][16 J + m];  xa[J][r]: Xi[J][g + 4 r][m]
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[J][r] = ok ? Wp[base + 16 * J + g + 4 * r] : 0.0;
#pragma unroll
    for (int J = 0; J < 4; ++J) {
#pragma unroll
        for (int r = 0; r < 4; ++r) xa[J][r] = Xi[(J * 16 + g + 4 * r) * 16 + m];
#pragma unroll
        for (int I = 0; I < J; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) ra[I][J][r] = -R[(size_t)(16 * I + g + 4 * r) * L + 16 * J + m];
    }
    asm volatile("" ::: "memory");               // the wave's source rows are read before any block's rows are overwritten
    if (ok) {
        double* const zs[3] = {z0, z1, z2};
#pragma unroll
        for (int b = 0; b < 3; ++b)
            if (zs[b] && zs[b] != W) {
#pragma unroll
                for (int J = 0; J < 4; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) zs[b][base + 16 * J + g + 4 * r] = 0.0;
            }
    }
#pragma unroll
    for (int J = 0; J < 4; ++J) {
        dvec4 t = T[J];
#pragma unroll
        for (int I = 0; I < J; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) t = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[I][J][r], X[I][r], t, 0, 0, 0);
        dvec4 d = dvec4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) d = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[J][r], t[r], d, 0, 0, 0);
        X[J] = d;
    }
    if (ok) {
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) W[base + 16 * J + g + 4 * r] = X[J][r];
    }
}

// ---- Chebyshev filter between two Rayleigh–Ritz steps ----------------------------------------------
// After a Ritz step the block holds Ritz vectors V (A2) with values theta and C V (A1).  The eigenvalues
// that are NOT wanted lie in [0, b] with b <= theta_l (the smallest Ritz value of the block bounds
// lambda_{l+1} from above), so instead of plain powers C^m V the block is filtered with the Chebyshev
// polynomial T_d((2C - bI)/b): |T_d| <= 1 on [0, b] and grows like cosh(d acosh t) outside — for the bench
// spectrum (theta_l/theta_npc = 0.24) a factor 14.9 per application of C against 4.2 for a plain power.
//   Y0 = V,  Y1 = a C V - V,  Y_{j+1} = 2 (a C Y_j - Y_j) - Y_{j-1},   a = 2 / b
// b is read from the device (theta[l_act - 1], floored at 1e-10 theta_0 so that a rank-deficient C cannot
// divide by zero: any b at or above the unwanted spectrum is a valid filter).
__device__ __forceinline__ double cheb_b(const double* __restrict__ theta, int l_act) {
    const double b = theta[l_act - 1], floor_ = 1e-10 * theta[0];
    return b > floor_ ? b : (floor_ > 0 ? floor_ : 1.0);
}
// A1 <- a A1 - A2   (Y1 from C V and V)
__global__ void k_cheb_first(double* __restrict__ A1, const double* __restrict__ A2, const double* __restrict__ theta,
                             int l_act, size_t n) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const double a = 2.0 / cheb_b(theta, l_act);
    A1[e] = a * A1[e] - A2[e];
}
// A second filter straight behind a CholeskyQR, no Ritz step between (round 5): the block W is orthonormal, Z = C W; Y0 = W goes into
// `prev`, Y1 = a Z - W into `cur`, Z is left zeroed — the state k_cheb_first leaves after a Ritz step, with W in the place of the
// Ritz vectors (its columns still are, nearly: a filtered, re-orthonormalised set of Ritz vectors).
__global__ void k_cheb_first2(double* __restrict__ Z, const double* __restrict__ W, double* __restrict__ cur, double* __restrict__ prev,
                              const double* __restrict__ theta, int l_act, size_t n) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const double a = 2.0 / cheb_b(theta, l_act);
    const double wv = W[e];
    cur[e] = a * Z[e] - wv;
    prev[e] = wv;
    Z[e] = 0.0;
}
// prev <- 2 (a Z - cur) - prev   (Y_{j+1} from Z = C Y_j, Y_j, Y_{j-1})
// (Z is left ZEROED: it is the destination of the next application of C, which accumulates into a zeroed block)
__global__ void k_cheb_step(double* __restrict__ Z, const double* __restrict__ cur, double* __restrict__ prev,
                            const double* __restrict__ theta, int l_act, size_t n) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const double a = 2.0 / cheb_b(theta, l_act);
    prev[e] = 2.0 * (a * Z[e] - cur[e]) - prev[e];
    Z[e] = 0.0;
}
// column i divided by T_d(t_i), t_i = (2 theta_i - b) / b: the filtered columns are (nearly) eigenvectors
// scaled by T_d(t_i); taking the known factor out keeps the CholeskyQR that follows well conditioned
__global__ void k_cheb_scale(double* __restrict__ Y, const double* __restrict__ theta, int l_act, int d, size_t n) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int c = (int)(e % L);
    if (c >= l_act) return;
    const double b = cheb_b(theta, l_act);
    const double t = (2.0 * theta[c] - b) / b;
    const double Td = t > 1.0 ? cosh((double)d * acosh(t)) : 1.0;
    Y[e] /= Td;
}

// sign convention of the components: the largest-|.| entry of each Ritz vector is positive
__global__ void k_signs(const double* __restrict__ colmax, double* __restrict__ sgn) {
    sgn[threadIdx.x] = colmax[threadIdx.x] < 0 ? -1.0 : 1.0;
}

#include "jacobi.inl"

// k_ritz_post's partials -> rho[c], colmax[c], then the scalars the host reads back per Ritz step:
//   out[0] = max_{i < n_pc} rho_i / max(theta_i, 1e-5 theta_1) (NaN-propagating; pairs of a numerically zero eigenvalue — more
//            components asked than the data have rank — are judged on the scale of the problem), out[1] = status bits,
//   out[2] = theta[l_act - 1] / theta[n_pc - 1]: the smallest Ritz value of the block over the last wanted one — an upper
//            estimate of the per-application convergence factor of the wanted pairs,
//   out[3] = status of the device-side feature selection (bit 0 = NaN variance), out[4] = theta_1 / theta_l (spread of the
//            block: bounds the filter degree).  1024 threads:
// 16 slices of the blocks per column (a single wave walking 128 x 3 dependent loads took 45 us), combined in fixed order.
__global__ __launch_bounds__(1024) void k_resid_final(const double* __restrict__ part, int n_blocks, const double* __restrict__ theta,
                                                      int n_pc, int l_act, const int* __restrict__ status,
                                                      const int* __restrict__ status_sel, double* __restrict__ rho,
                                                      double* __restrict__ colmax, double* __restrict__ out) {
    __shared__ double s_t[16][L], s_v[16][L], s_j[16][L], s_rho[L];
    const int c = threadIdx.x & (L - 1), sl = threadIdx.x / L;
    double t = 0.0, bv = 0.0, bj = 0.0;
    for (int b = sl; b < n_blocks; b += 16) {
        const double* p = part + (size_t)b * 3 * L;
        t += p[c];
        const double v = p[L + c], jj = p[2 * L + c];
        if (fabs(v) > fabs(bv) || (fabs(v) == fabs(bv) && v != 0.0 && jj < bj)) {
            bv = v;
            bj = jj;
        }
    }
    s_t[sl][c] = t;
    s_v[sl][c] = bv;
    s_j[sl][c] = bj;
    __syncthreads();
    if (sl == 0) {
        t = 0.0; bv = 0.0; bj = 0.0;
        for (int w = 0; w < 16; ++w) {
            t += s_t[w][c];
            const double v = s_v[w][c], jj = s_j[w][c];
            if (fabs(v) > fabs(bv) || (fabs(v) == fabs(bv) && v != 0.0 && jj < bj)) {
                bv = v;
                bj = jj;
            }
        }
        const double r = sqrt(t);
        rho[c] = r;
        colmax[c] = bv;
        s_rho[c] = r;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double resid = 0.0;
    for (int i = 0; i < n_pc; ++i) {
        const double den = theta[i] > 1e-5 * theta[0] ? theta[i] : 1e-5 * theta[0];
        const double q = den > 0 ? s_rho[i] / den : s_rho[i];
        if (!(q <= resid)) resid = q;
    }
    out[0] = resid;
    out[1] = (double)*status;
    out[2] = theta[n_pc - 1] > 0 ? theta[l_act - 1] / theta[n_pc - 1] : 1.0;
    out[3] = status_sel ? (double)*status_sel : 0.0;
    out[4] = theta[l_act - 1] > 0 ? theta[0] / theta[l_act - 1] : 1.0;
    out[5] = (double)status[1];               // full sweeps the eigen-solve of this step took (trace output)
}
