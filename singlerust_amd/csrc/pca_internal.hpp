// pca_internal.hpp — what the three translation units of the PCA path share (pca_form.hip: compaction + Gram formation;
// pca_solve.hip: the sparse products, the k x 64 subspace iteration and the solver driver; pca.hip: selection, the per-matrix
// driver, backed sessions and the C entry points).  Constants, the entry / record formats, small device helpers, the
// descriptors of the compacted matrix, and the functions that cross a translation-unit boundary.
#pragma once

#include <algorithm>
#include <cmath>
#include <numeric>
#include <string>
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace srx {

constexpr int L = 64;               // panel width l
constexpr int kTThreads = 1024;     // transposed / Gram kernels: one workgroup per CU

// Tile-major layouts of the compacted matrix: the selected features are cut into tiles of kt compacted columns:
constexpr int KT = 256;
constexpr int KG = 128;

template <typename VT> struct GramPk;
template <> struct __attribute__((aligned(8))) GramPk<float> { int32_t j; float v; };
template <> struct __attribute__((aligned(16))) GramPk<double> { int32_t j; int32_t pad_; double v; };

constexpr int kGramWaves = 16;            // waves per Gram workgroup
constexpr int kGramUnroll = 8;            // suffix loads in flight per wave

// One unit of Gram work: entry (ja, va) times up to 64 consecutive entries of its row's suffix (the suffix starts at the
// entry itself — the diagonal product — and a suffix longer than a wave is cut into several records).
// pos: first suffix entry of this record, relative to its block's first entry; lenrb = lanes | rbase << 8, where
// rbase + jb is the index of G[ja][jb] among the owner's LDS accumulators (gram_row_base: may be negative, jb >= ja).
// (8-byte records — the entry's value fetched in the kernel instead of carried in the record, the piece index in the spare bits
//  of lenrb — were measured in round 3: the bucket pass gains 0.13 ms (0.93 -> 0.80) and the stripe kernel loses 0.45 with a
//  scalar load of the value (it shares lgkmcnt with the LDS atomics: waiting for it drains them) and 1.35 with a wave-uniform
//  vector load (one more L1 access per record).  The value stays in the record.)
template <typename VT> struct GramRec { uint32_t pos, lenrb; VT va; };
// records of a row with n kept entries: sum over suffix lengths L = 1 .. n of ceil(L / 64)
__host__ __device__ __forceinline__ uint64_t gram_row_records(uint64_t n) {
    const uint64_t q = n >> 6, r = n & 63;
    return 32 * q * (q + 1) + r * (q + 1);
}

__device__ __forceinline__ int gram_owner(int c, int sr_shift, int n_wg, int n_stripes) {
    const int s = c >> sr_shift;
    return s < n_wg ? s : n_stripes - 1 - s;
}

// accumulator layout of an owner: stripe A (rows a0 .. a0 + SR - 1, WA = k - a0 columns from a0), then stripe B (rows from
// b0 = the mirrored stripe, WB = k - b0 columns): the offset to which a column index jb >= ja is added
__device__ __forceinline__ int gram_row_base(int ja, int k, int sr_shift, int n_wg, int n_stripes) {
    const int s = ja >> sr_shift, SR = 1 << sr_shift;
    const int s0 = s << sr_shift;                       // first row of ja's stripe
    if (s < n_wg) return (ja - s0) * (k - s0) - s0;
    const int a0 = (n_stripes - 1 - s) << sr_shift;     // the owner's stripe A
    return SR * (k - a0) + (ja - s0) * (k - s0) - s0;
}

// index of (i, j), i <= j, in the packed upper triangle (row-major, row i holds columns i .. k - 1)
__host__ __device__ __forceinline__ size_t tri_index(int i, int j, int k) {
    return (size_t)i * (size_t)k - (size_t)i * (size_t)(i - 1) / 2 + (size_t)(j - i);
}

// ---- small vector helpers ------------------------------------------------------------------------
template <typename PT> struct Vec4;
template <> struct Vec4<float> {
    float4 v;
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
    __device__ __forceinline__ float& operator[](int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
};
template <> struct Vec4<double> {
    double2 a, b;
    __device__ __forceinline__ void load(const double* p) {
        a = *reinterpret_cast<const double2*>(p);
        b = *reinterpret_cast<const double2*>(p + 2);
    }
    __device__ __forceinline__ void store(double* p) const {
        *reinterpret_cast<double2*>(p) = a;
        *reinterpret_cast<double2*>(p + 2) = b;
    }
    __device__ __forceinline__ double& operator[](int i) { return i == 0 ? a.x : i == 1 ? a.y : i == 2 ? b.x : b.y; }
};

// f(integral_constant<int, I>) for I = 0 .. N-1: a loop whose index is a compile-time constant in the body
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// DPP row rotate inside each 16-lane row (v_mov_b32_dpp row_ror:S) — no LDS traffic.
template <int S>
__device__ __forceinline__ int ror16(int x) {
    if constexpr (S == 0) return x;
    else return __builtin_amdgcn_update_dpp(0, x, 0x120 + S, 0xf, 0xf, false);
}
template <int S>
__device__ __forceinline__ float ror16(float x) {
    return __builtin_bit_cast(float, ror16<S>(__builtin_bit_cast(int, x)));
}
template <int S>
__device__ __forceinline__ double ror16(double x) {
    long long b = __builtin_bit_cast(long long, x);
    int lo = ror16<S>((int)(b & 0xffffffffll)), hi = ror16<S>((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ int64_t readlane64(int64_t x, int l) {
    int lo = __builtin_amdgcn_readlane((int)(x & 0xffffffffll), l);
    int hi = __builtin_amdgcn_readlane((int)(x >> 32), l);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ float readlane_v(float x, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}
__device__ __forceinline__ double readlane_v(double x, int l) {
    return __builtin_bit_cast(double, readlane64(__builtin_bit_cast(long long, x), l));
}

// ---- compacted matrix: row-major records + the tile-major view of the forward SpMM ------------------
struct CompactCsr {
    uint64_t n_rows = 0, nnz = 0;
    int k = 0;
    int64_t* indptr = nullptr;
    int32_t* idx = nullptr;
    void* vals = nullptr;
};
struct Tiled {
    uint64_t n_rows = 0, nnz = 0;
    int k = 0, kt = 0, nt = 0;
    int64_t* tptr = nullptr;   // nt * n_rows + 1
    void* tpk = nullptr;       // GramPk<VT> records: (local column within the tile, value)
};
// How G's upper triangle is cut into stripes of SR rows and paired into workgroups (k_gram_stripes)
struct GramPlan {
    int k = 0, sr_shift = 0, n_stripes = 0, n_wg = 0;
    int n_z = 1;               // chunks: the grid is n_wg x n_z workgroups
    uint32_t rblk = 512;       // cells per bucket block
    uint32_t n_chunk = 0;      // consecutive row blocks per workgroup
    uint64_t n_rblk = 0;
    size_t lds_bytes = 0;
};
// X[:, sel] row by row: GramPk<VT> records (compacted column in [0, k), value), columns ascending within a row —
// what the Gram kernel walks (a suffix of a row is one contiguous run)
struct RowMajor {
    uint64_t n_rows = 0, nnz = 0;
    int k = 0;
    int64_t* ptr = nullptr;    // n_rows + 1
    void* pk = nullptr;
    uint32_t* perm = nullptr;  // rows ordered by their number of kept entries (forward SpMM), or null
    int64_t n_recs = -1;       // >= 0: the Gram kernel's record counts were made with the compaction (scratch pca_brtot / pca_brbase
                               // hold them): launch_gram starts at the bucket pass, one host wait less per step
};
struct Resolved {
    int n_pc, center, scale, max_iter, solver;
    double bail_ratio = 0.0; // > 0: give the round up after its first Ritz step when theta_l / theta_npc exceeds this (the
                             // caller has a plan with more guard columns per round)
    bool direct = false;     // k <= 64 with the explicit matrix: the block is the identity, one exact eigen-solve of C
    bool robust = false;     // last resort after a breakdown: CholeskyQR after every application of C, shifted
                             // CholeskyQR3, plain sweeps instead of Chebyshev filters
    int power = 1;           // applications of C per Rayleigh–Ritz step
    int warm = 0;            // leading sweeps of `power` applications + CholeskyQR WITHOUT a Rayleigh–Ritz step
    double tol;
    uint64_t seed;
};


// ---- genes.hip -------------------------------------------------------------------------------------
int32_t gene_variances(srx_mat* m, std::vector<double>& var);
int32_t select_hvg_host(srx_ctx* ctx, const std::vector<double>& var, uint64_t n, std::vector<uint64_t>& out);
int32_t launch_tile_ptr(srx_ctx* ctx, const int64_t* indptr, const int32_t* idx, uint64_t n_rows, int n_tiles,
                        int tile_genes, int64_t* tp);

// ---- pca_form.hip: the HVG-compacted matrix and G = A^T A --------------------------------------------
// (entries_per_cell, entry_bytes: the compacted matrix's mean row length and entry size — they size the stripe kernel's chunks; 0 where
//  only the blocks are wanted)
int32_t gram_plan(srx_ctx* ctx, int k, uint64_t n_rows, GramPlan& g, double entries_per_cell = 0.0, int entry_bytes = 0);
int grid_rows(const srx_ctx* ctx, uint64_t n_rows, int rows_per_block);
int32_t scan_exclusive(srx_ctx* ctx, const int64_t* d_in, uint64_t n, int64_t* d_out, int64_t** total_dev);
int32_t build_compact(srx_mat* m, const std::vector<int32_t>& remap, int k, CompactCsr& c, RowMajor& rm);
int32_t retile(srx_mat* m, const CompactCsr& c, int kt, Tiled& t);
// `d_sel`: n_words selection bits followed by n_words prefix counts, on the device (the compacted column of a gene is its rank
// among the selected genes in ascending gene order); the second form takes a host remap table (explicit feature lists)
int32_t build_tiled_fused(srx_mat* m, const uint32_t* d_sel, int n_words, int k, RowMajor& rm, Tiled* t256p, RowXf xf = RowXf{},
                          bool want_recs = false);
int32_t build_tiled_fused(srx_mat* m, const std::vector<int32_t>& remap, int k, RowMajor& rm, Tiled* t256, RowXf xf = RowXf{},
                          bool want_recs = false);
template <typename VT>
int32_t launch_gram(srx_ctx* ctx, const RowMajor& rm, double* Gp, bool* reduce = nullptr, bool zero_first = false);
inline size_t gram_packed_count(int k) { return (size_t)k * (size_t)(k + 1) / 2; }

// ---- pca_solve.hip: products, iteration, solver ------------------------------------------------------
int32_t build_row_order(srx_ctx* ctx, RowMajor& r);
template <typename VT, typename PT>
bool fwd_rows_fits(int k);
int scores_ld(int n_pc);
void result_layout(uint64_t n_rows, int k, int n_pc, int dim, size_t& score_bytes, size_t& small_doubles,
                   std::vector<int>* plan_a_out = nullptr, std::vector<int>* plan_b_out = nullptr);
int32_t ensure_result_capacity(srx_ctx* ctx, srx_pca_state& st, size_t need);
int32_t resolve_opts(srx_ctx* ctx, const srx_pca_opts* opts, int k, uint64_t Ng, bool f32, Resolved& o, int& l_act);
template <typename VT, typename PT>
int32_t run_pca(srx_ctx* ctx, const RowMajor* parts, int n_parts, const Tiled* t256p, double* gram_packed, const Resolved& o,
                const std::vector<double>& mu, const std::vector<double>& dinv, const HvgDev* hv, int l_act, double n_cells,
                srx_pca_state& st);
// the small results of a solve move from the context's scratch into the matrix's own result block (host copies: pca_materialize)
int32_t stash_results(srx_ctx* ctx, srx_pca_state& st, int k, int n_pc, const HvgDev* hv,
                             const std::vector<double>& mu, const std::vector<double>& sd, double trace,
                             const std::vector<uint64_t>& selv);

}  // namespace srx
