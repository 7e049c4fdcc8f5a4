// The reference's own tests for the path (src/memory/processing/mod.rs:421-481 test_normalize_total
// with check_row_sums / check_column_sums; the path's remaining stages have no reference tests),
// restated against the C++ host mirror singlerust_amd/host/single_rust.hpp.  Same workload shape:
// 1000 x 100, nnz = nrows*ncols/10 COO draws of Uniform(0, 50) with duplicates summed, f64 values,
// assert |sum - 1e4| < 1e-6 on every row / column with at least one entry.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>

#include "../../singlerust_amd/host/single_rust.hpp"

using namespace single_rust;
namespace proc = single_rust::memory::processing;
namespace stats = single_rust::memory::statistics;

static int failures = 0;
#define EXPECT(cond, ...)                                     \
    do {                                                      \
        if (!(cond)) {                                        \
            ++failures;                                       \
            std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            std::fprintf(stderr, __VA_ARGS__);                \
            std::fprintf(stderr, "\n");                       \
        }                                                     \
    } while (0)

// CsrMatrix::from(&CooMatrix): duplicates are summed, columns sorted within a row
static CsrMatrix<double> create_large_test_data(std::size_t nrows, std::size_t ncols, double sparsity, unsigned seed) {
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> value(0.0, 50.0);
    const std::size_t nnz = (std::size_t)std::llround((double)(nrows * ncols) * (1.0 / sparsity));
    std::vector<std::map<std::uint64_t, double>> rows(nrows);
    for (std::size_t t = 0; t < nnz; ++t) {
        std::size_t r = rng() % nrows, c = rng() % ncols;
        rows[r][c] += value(rng);
    }
    CsrMatrix<double> m;
    m.nrows = nrows;
    m.ncols = ncols;
    m.row_offsets.push_back(0);
    for (auto& r : rows) {
        for (auto& kv : r) {
            m.col_indices.push_back(kv.first);
            m.values.push_back(kv.second);
        }
        m.row_offsets.push_back(m.col_indices.size());
    }
    return m;
}

static std::vector<std::string> names(const char* p, std::size_t n) {
    std::vector<std::string> v;
    for (std::size_t i = 0; i < n; ++i) v.push_back(p + std::to_string(i));
    return v;
}

static void check_row_sums(const CsrMatrix<double>& csr, double target) {
    for (std::size_t r = 0; r < csr.nrows; ++r) {
        if (csr.row_offsets[r] == csr.row_offsets[r + 1]) continue;
        double s = 0.0;
        for (auto p = csr.row_offsets[r]; p < csr.row_offsets[r + 1]; ++p) s += csr.values[p];
        EXPECT(std::fabs(s - target) < 1e-6, "row %zu sum %.12g does not match target sum %g", r, s, target);
    }
}

static void check_column_sums(const CsrMatrix<double>& csr, double target) {
    std::vector<double> sums(csr.ncols, 0.0);
    std::vector<int> seen(csr.ncols, 0);
    for (std::size_t p = 0; p < csr.values.size(); ++p) {
        sums[csr.col_indices[p]] += csr.values[p];
        seen[csr.col_indices[p]] = 1;
    }
    for (std::size_t c = 0; c < csr.ncols; ++c)
        if (seen[c]) EXPECT(std::fabs(sums[c] - target) < 1e-6, "column %zu sum %.12g does not match %g", c, sums[c], target);
}

static void test_normalize_total(Context& ctx) {
    auto x = create_large_test_data(1000, 100, 10.0, 7);
    IMAnnData adata = IMAnnData::new_basic(ctx, x, names("obs", 1000), names("var", 100));
    const double target_sum = 1e4;

    IMAnnData normalized = proc::normalize_total(adata, target_sum, Direction::Row);
    EXPECT(normalized.x_is_f64(), "expected the F64 arm after normalisation");
    check_row_sums(normalized.x_f64(), target_sum);

    IMAnnData by_col = proc::normalize_total(adata, target_sum, Direction::Column);
    check_column_sums(by_col.x_f64(), target_sum);

    // the input was not touched by the cloning variants
    auto sums = stats::compute_sum(adata, Direction::Row);
    double ref0 = 0.0;
    for (auto p = x.row_offsets[0]; p < x.row_offsets[1]; ++p) ref0 += x.values[p];
    // (summation order differs from the serial loop for non-integer values: compare to 1e-13 relative)
    EXPECT(std::fabs(sums[0] - ref0) <= 1e-13 * ref0, "input modified by normalize_total: %.17g vs %.17g", sums[0], ref0);
}

static void test_path_end_to_end(Context& ctx) {
    auto x = create_large_test_data(1000, 100, 10.0, 11);
    IMAnnData adata = IMAnnData::new_basic(ctx, x, names("obs", 1000), names("var", 100));
    proc::normalize_total_inplace(adata, 1e4, Direction::Row);
    proc::log1p_transform_inplace(adata);
    auto vals = adata.x_f64().values;
    // spot-check log1p(v * (1e4 / rowsum)) on the first row, two roundings as the reference has them
    double s = 0.0;
    for (auto p = x.row_offsets[0]; p < x.row_offsets[1]; ++p) s += x.values[p];
    for (auto p = x.row_offsets[0]; p < x.row_offsets[1]; ++p) {
        double want = std::log1p(x.values[p] * (1e4 / s));
        EXPECT(std::fabs(vals[p] - want) <= 4e-16 * std::fabs(want), "log1p mismatch at %zu: %.17g vs %.17g", (size_t)p,
               vals[p], want);
    }
    auto hv = proc::dim_red::select_features(adata, FeatureSelection::HighlyVariable(40));
    EXPECT(hv.size() == 40, "HVG count %zu", hv.size());
    auto var = stats::compute_variance(adata, Direction::Column);
    for (std::size_t i = 1; i < hv.size(); ++i)
        EXPECT(var[hv[i - 1]] > var[hv[i]] || (var[hv[i - 1]] == var[hv[i]] && hv[i - 1] < hv[i]),
               "HVG order broken at %zu", i);

    // the reference's call, argument for argument: pca_inplace(&mut adata, Some(10), None, None, None, &HighlyVariable(40), FaerSVD)
    auto info = proc::dim_red::pca_inplace(adata, 10, {}, {}, {}, FeatureSelection::HighlyVariable(40), single_rust::memory::processing::dim_red::FaerSVD{});
    const Array2& pcs = adata.obsm().at("X_pca");
    EXPECT(pcs.nrows == 1000 && pcs.ncols == 10, "X_pca shape %zu x %zu", pcs.nrows, pcs.ncols);
    EXPECT(info.n_pc == 10 && info.k == 40, "pca info n_pc=%d k=%llu", info.n_pc, (unsigned long long)info.k);
    // scores are centred and their variances descend
    double prev = INFINITY;
    for (std::size_t c = 0; c < pcs.ncols; ++c) {
        double m = 0.0, q = 0.0;
        for (std::size_t r = 0; r < pcs.nrows; ++r) m += pcs(r, c);
        m /= (double)pcs.nrows;
        for (std::size_t r = 0; r < pcs.nrows; ++r) q += (pcs(r, c) - m) * (pcs(r, c) - m);
        EXPECT(std::fabs(m) < 1e-8, "PC %zu mean %.3g", c, m);
        EXPECT(q <= prev * (1 + 1e-9), "PC variances not descending at %zu", c);
        prev = q;
    }
    // qc_vars_inplace (statistics/mod.rs:74-103): the eight columns under the reference's names
    {
        IMAnnData q = adata.deep_clone();
        stats::qc_vars_inplace(q);
        EXPECT(q.obs_columns().size() == 4 && q.var_columns().size() == 4, "qc_vars_inplace column counts");
        EXPECT(q.obs_columns().at("num_genes_per_cell").size() == 1000 && q.var_columns().at("std_dev_per_gene").size() == 100,
               "qc_vars_inplace column lengths");
        auto n = stats::compute_number(q, Direction::Column);
        for (std::size_t j = 0; j < n.size(); ++j)
            EXPECT(q.var_columns().at("num_cells_per_gene")[j] == (double)n[j], "num_cells_per_gene[%zu]", j);
    }
    // the remaining FeatureSelection arms (dim_red/mod.rs:125-134, 141-153)
    {
        std::vector<bool> flag(100, false);
        for (std::size_t i = 0; i < hv.size(); ++i) flag[hv[i]] = true;
        adata.var_bool()["highly_variable"] = flag;
        auto by_col = proc::dim_red::select_features(adata, FeatureSelection::HighlyVariableCol("highly_variable"));
        auto sorted = hv;
        std::sort(sorted.begin(), sorted.end());
        EXPECT(by_col == sorted, "HighlyVariableCol returns the true entries in index order");
        auto rnd = proc::dim_red::select_features(adata, FeatureSelection::Randomized(30));
        std::sort(rnd.begin(), rnd.end());
        EXPECT(rnd.size() == 30 && std::adjacent_find(rnd.begin(), rnd.end()) == rnd.end() && rnd.back() < 100, "Randomized(30)");
        auto thr = proc::dim_red::select_features(adata, FeatureSelection::VarianceThreshold(var[hv[39]]));
        EXPECT(thr.size() == 39, "VarianceThreshold keeps the %zu genes above the 40th variance", thr.size());
    }
    // defaults: n_components None -> 2 (dim_red/mod.rs:52)
    IMAnnData again = adata.deep_clone();
    proc::dim_red::pca_inplace(again, {}, {}, {}, {}, FeatureSelection::None());
    EXPECT(again.obsm().at("X_pca").ncols == 2, "default n_components");
}

// mod.rs:385-417 (test_filter_cells / test_filter_genes): every filter drops something on the random matrix
static void test_filters(Context& ctx) {
    auto x = create_large_test_data(1000, 100, 10.0, 19);
    IMAnnData adata = IMAnnData::new_basic(ctx, x, names("obs", 1000), names("var", 100));
    IMAnnData f1 = proc::filter_cells(adata, FlexValue::Absolute(5), FlexValue::Absolute(15));
    EXPECT(f1.n_obs() < adata.n_obs(), "cells after absolute filtering: %zu", f1.n_obs());
    auto num = stats::compute_number(f1, Direction::Row);
    for (auto c : num) EXPECT(c >= 5 && c <= 15, "kept cell with %u genes", c);
    IMAnnData f2 = proc::filter_cells(adata, FlexValue::Relative(0.1), FlexValue::Relative(0.9));
    EXPECT(f2.n_obs() < adata.n_obs(), "cells after relative filtering: %zu", f2.n_obs());
    IMAnnData g1 = proc::filter_genes(adata, FlexValue::Absolute(50), FlexValue::Absolute(100));
    EXPECT(g1.n_vars() < adata.n_vars(), "genes after absolute filtering: %zu", g1.n_vars());
    IMAnnData g2 = proc::filter_genes(adata, FlexValue::Relative(0.1), FlexValue::Relative(0.9));
    EXPECT(g2.n_vars() < adata.n_vars(), "genes after relative filtering: %zu", g2.n_vars());
    proc::filter_genes_inplace(adata, FlexValue::Relative(0.1), FlexValue::None());
    EXPECT(adata.n_vars() == 90 && adata.var_names().size() == 90, "in-place gene filter kept %zu", adata.n_vars());
}

// src/backed/statistics/mod.rs:5-45 over row chunks == the resident statistics; the whole path over chunks == resident
static void test_backed(Context& ctx) {
    namespace bk = single_rust::backed;
    auto x = create_large_test_data(1000, 100, 10.0, 23);
    IMAnnData adata = IMAnnData::new_basic(ctx, x, names("obs", 1000), names("var", 100));
    bk::BackedAnnData b = bk::BackedAnnData::of(ctx, x);
    for (std::size_t chunk : {std::size_t(64), std::size_t(333), std::size_t(5000)}) {
        for (Direction d : {Direction::Row, Direction::Column}) {
            auto n1 = bk::statistics::compute_number(b, d, ComputationMode::Chunked(chunk));
            EXPECT(n1 == stats::compute_number(adata, d), "chunked compute_number differs (chunk %zu)", chunk);
            auto s1 = bk::statistics::compute_sum(b, d, ComputationMode::Chunked(chunk));
            auto s0 = stats::compute_sum(adata, d);
            for (std::size_t i = 0; i < s0.size(); ++i)
                EXPECT(std::fabs(s1[i] - s0[i]) <= 1e-12 * std::fabs(s0[i]), "chunked compute_sum differs at %zu", i);
        }
    }
    EXPECT(bk::statistics::compute_number(b, Direction::Row, ComputationMode::Whole()) == stats::compute_number(adata, Direction::Row),
           "ComputationMode::Whole differs");
    auto r = bk::processing::pca_pipeline(b, 300, 1e4, 40, 10);
    proc::normalize_total_inplace(adata, 1e4, Direction::Row);
    proc::log1p_transform_inplace(adata);
    auto hv = proc::dim_red::select_features(adata, FeatureSelection::HighlyVariable(40));
    EXPECT(r.selected == hv, "backed HVG list differs from the resident one");
    proc::dim_red::pca_inplace(adata, 10, {}, {}, {}, FeatureSelection::HighlyVariable(40));
    const Array2& pcs = adata.obsm().at("X_pca");
    for (std::size_t c = 0; c < 10; ++c) {
        double dot = 0, nn = 0, dd = 0;
        for (std::size_t i = 0; i < 1000; ++i) dot += pcs(i, c) * r.x_pca(i, c);
        const double sgn = dot >= 0 ? 1.0 : -1.0;
        for (std::size_t i = 0; i < 1000; ++i) {
            nn += pcs(i, c) * pcs(i, c);
            dd += (r.x_pca(i, c) - sgn * pcs(i, c)) * (r.x_pca(i, c) - sgn * pcs(i, c));
        }
        EXPECT(std::sqrt(dd / nn) < 1e-8, "backed PC %zu differs from the resident one by %.3g", c, std::sqrt(dd / nn));
    }
}

// ArrayData::CscMatrix: the same X as CSR and as CSC gives the same statistics and the same normalised row sums
static void test_csc(Context& ctx) {
    auto x = create_large_test_data(300, 40, 10.0, 29);
    CscMatrix<double> c;
    c.nrows = x.nrows;
    c.ncols = x.ncols;
    std::vector<std::vector<std::pair<std::uint64_t, double>>> cols(x.ncols);
    for (std::size_t r = 0; r < x.nrows; ++r)
        for (auto p = x.row_offsets[r]; p < x.row_offsets[r + 1]; ++p) cols[x.col_indices[p]].push_back({r, x.values[p]});
    c.col_offsets.push_back(0);
    for (auto& col : cols) {
        for (auto& e : col) {
            c.row_indices.push_back(e.first);
            c.values.push_back(e.second);
        }
        c.col_offsets.push_back(c.row_indices.size());
    }
    IMAnnData a_csr = IMAnnData::new_basic(ctx, x, names("obs", 300), names("var", 40));
    IMAnnData a_csc = IMAnnData::new_basic(ctx, c, names("obs", 300), names("var", 40));
    EXPECT(a_csc.x_is_csc() && !a_csr.x_is_csc(), "storage format flags");
    EXPECT(a_csc.n_obs() == 300 && a_csc.n_vars() == 40, "CSC shape %zu x %zu", a_csc.n_obs(), a_csc.n_vars());
    for (Direction d : {Direction::Row, Direction::Column}) {
        EXPECT(stats::compute_number(a_csc, d) == stats::compute_number(a_csr, d), "compute_number CSC vs CSR");
        auto s1 = stats::compute_sum(a_csc, d), s0 = stats::compute_sum(a_csr, d);
        for (std::size_t i = 0; i < s0.size(); ++i)
            EXPECT(std::fabs(s1[i] - s0[i]) <= 1e-12 * std::fabs(s0[i]), "compute_sum CSC vs CSR at %zu", i);
    }
    proc::normalize_total_inplace(a_csc, 1e4, Direction::Row);          // scale_row_csc
    auto sums = stats::compute_sum(a_csc, Direction::Row);
    auto num = stats::compute_number(a_csc, Direction::Row);
    for (std::size_t i = 0; i < sums.size(); ++i)
        if (num[i]) EXPECT(std::fabs(sums[i] - 1e4) < 1e-6, "CSC row %zu sums to %.12g after normalize_total", i, sums[i]);
    proc::log1p_transform_inplace(a_csc);
    auto info = proc::dim_red::pca_inplace(a_csc, 5, {}, {}, {}, FeatureSelection::None());
    EXPECT(info.n_pc == 5 && a_csc.obsm().at("X_pca").nrows == 300, "PCA from a CSC matrix");
}

static void test_errors(Context& ctx) {
    auto x = create_large_test_data(4, 6, 2.0, 3);
    IMAnnData tiny = IMAnnData::new_basic(ctx, x, names("obs", 4), names("var", 6));
    bool threw = false;
    try {
        proc::dim_red::pca_inplace(tiny, 2, {}, {}, {}, FeatureSelection::None());   // N < 5 panics in the reference
    } catch (const Error& e) {
        threw = e.code == SRX_E_SHAPE;
    }
    EXPECT(threw, "PCA on 4 cells must fail with SRX_E_SHAPE");
}

int main() {
    try {
        Context ctx(0);
        test_normalize_total(ctx);
        test_path_end_to_end(ctx);
        test_filters(ctx);
        test_backed(ctx);
        test_csc(ctx);
        test_errors(ctx);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 2;
    }
    if (failures) return 1;
    std::puts("host mirror: all checks passed");
    return 0;
}
