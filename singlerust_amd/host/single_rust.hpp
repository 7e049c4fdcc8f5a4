// single_rust.hpp — C++ host-side mirror of the reference's API for the hot path, over the C ABI
// of include/srx.h.
//
// The reference is Rust (src/lib.rs:5-15 re-exports `memory::{processing, statistics}`); this image
// has no Rust toolchain, so the host side above the C ABI is C++ (the reference is compiled code).
// Names, argument order and error behaviour follow the reference so that a test written against
// this header reads like the reference's own (`src/memory/processing/mod.rs:334-482`):
//
//   single_rust::IMAnnData adata = single_rust::IMAnnData::new_basic(x, obs_names, var_names);
//   single_rust::memory::processing::normalize_total_inplace(adata, 1e4, Direction::Row);
//   single_rust::memory::processing::log1p_transform_inplace(adata);
//   single_rust::memory::processing::dim_red::pca_inplace(adata, 50, {}, {}, {}, FeatureSelection::HighlyVariable(2000));
//   const Array2& x_pca = adata.obsm().at("X_pca");
//
// `anyhow::Result<T>` becomes "returns T or throws single_rust::Error" (message = srx_last_error).
// Header-only; link with -lsrx_hip.
#pragma once

#include <algorithm>
#include <cstdint>
#include <map>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "../../include/srx.h"

namespace single_rust {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

enum class Direction { Row = 0, Column = 1 };                   // src/shared/mod.rs:39-42

struct FeatureSelection {                                        // src/shared/mod.rs:17-23
    enum Kind { HighlyVariableColK, HighlyVariableK, RandomizedK, VarianceThresholdK, NoneK } kind = NoneK;
    std::string col;
    std::size_t n = 0;
    double threshold = 0.0;
    static FeatureSelection HighlyVariableCol(std::string c) { FeatureSelection f; f.kind = HighlyVariableColK; f.col = std::move(c); return f; }
    static FeatureSelection HighlyVariable(std::size_t n) { FeatureSelection f; f.kind = HighlyVariableK; f.n = n; return f; }
    static FeatureSelection Randomized(std::size_t n) { FeatureSelection f; f.kind = RandomizedK; f.n = n; return f; }
    static FeatureSelection VarianceThreshold(double t) { FeatureSelection f; f.kind = VarianceThresholdK; f.threshold = t; return f; }
    static FeatureSelection None() { return FeatureSelection(); }
};

struct FlexValue {                                               // src/shared/mod.rs:62-66
    srx_flex c{SRX_FLEX_NONE, 0, 0.0};
    static FlexValue Absolute(std::uint32_t v) { FlexValue f; f.c = srx_flex{SRX_FLEX_ABSOLUTE, v, 0.0}; return f; }
    static FlexValue Relative(double p) { FlexValue f; f.c = srx_flex{SRX_FLEX_RELATIVE, 0, p}; return f; }
    static FlexValue None() { return FlexValue(); }
};

// nalgebra_sparse::CsrMatrix<T> as the reference holds it (usize offsets/indices, typed values).
template <typename T>
struct CsrMatrix {
    std::size_t nrows = 0, ncols = 0;
    std::vector<std::uint64_t> row_offsets, col_indices;
    std::vector<T> values;
};
template <typename T>
struct CscMatrix {                                               // nalgebra_sparse::CscMatrix<T>
    std::size_t nrows = 0, ncols = 0;
    std::vector<std::uint64_t> col_offsets, row_indices;
    std::vector<T> values;
};
template <typename T> struct DTypeOf;
template <> struct DTypeOf<std::int8_t> { static constexpr int v = SRX_I8; };
template <> struct DTypeOf<std::int16_t> { static constexpr int v = SRX_I16; };
template <> struct DTypeOf<std::int32_t> { static constexpr int v = SRX_I32; };
template <> struct DTypeOf<std::uint8_t> { static constexpr int v = SRX_U8; };
template <> struct DTypeOf<std::uint16_t> { static constexpr int v = SRX_U16; };
template <> struct DTypeOf<std::uint32_t> { static constexpr int v = SRX_U32; };
template <> struct DTypeOf<float> { static constexpr int v = SRX_F32; };
template <> struct DTypeOf<double> { static constexpr int v = SRX_F64; };

struct Array2 {                                                  // ndarray::Array2<f64>, row-major
    std::size_t nrows = 0, ncols = 0;
    std::vector<double> data;
    double operator()(std::size_t r, std::size_t c) const { return data[r * ncols + c]; }
};

class Context {
public:
    explicit Context(int device_id = 0) {
        int rc = srx_ctx_create(device_id, &h_);
        if (rc != SRX_OK) throw Error(rc, srx_last_error(nullptr));
    }
    ~Context() { srx_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    srx_ctx* handle() const { return h_; }
    void check(int rc) const {
        if (rc != SRX_OK) throw Error(rc, srx_last_error(h_));
    }

private:
    srx_ctx* h_ = nullptr;
};

// anndata_memory::IMAnnData reduced to what the path touches: X (device-resident), names, obsm/varm.
class IMAnnData {
public:
    template <typename T>
    static IMAnnData new_basic(Context& ctx, const CsrMatrix<T>& x, std::vector<std::string> obs_names,
                               std::vector<std::string> var_names, int store = SRX_STORE_AUTO) {
        IMAnnData a(ctx);
        srx_csr h{x.nrows, x.ncols, x.values.size(), x.row_offsets.data(), x.col_indices.data(),
                  const_cast<T*>(x.values.data()), DTypeOf<T>::v};
        ctx.check(srx_matrix_upload(ctx.handle(), &h, store, &a.x_));
        a.row_offsets_ = x.row_offsets;
        a.col_indices_ = x.col_indices;
        a.obs_names_ = std::move(obs_names);
        a.var_names_ = std::move(var_names);
        return a;
    }
    // X as ArrayData::CscMatrix: same entry points, the reference's CSC arithmetic (helper/csc.rs, scale_*_csc)
    template <typename T>
    static IMAnnData new_basic(Context& ctx, const CscMatrix<T>& x, std::vector<std::string> obs_names,
                               std::vector<std::string> var_names, int store = SRX_STORE_AUTO) {
        IMAnnData a(ctx);
        srx_csr h{x.nrows, x.ncols, x.values.size(), x.col_offsets.data(), x.row_indices.data(),
                  const_cast<T*>(x.values.data()), DTypeOf<T>::v};
        ctx.check(srx_matrix_upload_csc(ctx.handle(), &h, store, &a.x_));
        a.row_offsets_ = x.col_offsets;
        a.col_indices_ = x.row_indices;
        a.obs_names_ = std::move(obs_names);
        a.var_names_ = std::move(var_names);
        return a;
    }
    bool x_is_csc() const {
        int32_t f = 0;
        ctx_->check(srx_matrix_format(x_, &f));
        return f == SRX_FORMAT_CSC;
    }
    IMAnnData(IMAnnData&& o) noexcept : ctx_(o.ctx_) { *this = std::move(o); }
    IMAnnData& operator=(IMAnnData&& o) noexcept {
        std::swap(x_, o.x_);
        row_offsets_ = std::move(o.row_offsets_);
        col_indices_ = std::move(o.col_indices_);
        obs_names_ = std::move(o.obs_names_);
        var_names_ = std::move(o.var_names_);
        obsm_ = std::move(o.obsm_);
        varm_ = std::move(o.varm_);
        var_bool_ = std::move(o.var_bool_);
        obs_cols_ = std::move(o.obs_cols_);
        var_cols_ = std::move(o.var_cols_);
        return *this;
    }
    ~IMAnnData() { srx_matrix_free(x_); }

    std::size_t n_obs() const { return info().n_rows; }
    std::size_t n_vars() const { return info().n_cols; }
    srx_mat* x() const { return x_; }
    Context& ctx() const { return *ctx_; }
    srx_mat_info info() const {
        srx_mat_info i{};
        ctx_->check(srx_matrix_info(x_, &i));
        return i;
    }
    // x().get_data(): the current values as the DynCsrMatrix variant X now is (F32 or F64)
    bool x_is_f64() const { return info().dtype != SRX_F32; }
    CsrMatrix<double> x_f64() const {
        CsrMatrix<double> m;
        auto i = info();
        m.nrows = i.n_rows;
        m.ncols = i.n_cols;
        m.row_offsets = row_offsets_;
        m.col_indices = col_indices_;
        m.values.resize(i.nnz);
        ctx_->check(srx_matrix_download_values(x_, m.values.data(), SRX_F64));
        return m;
    }
    IMAnnData deep_clone() const {                               // processing/mod.rs:315,330
        IMAnnData c(*ctx_);
        ctx_->check(srx_matrix_clone(x_, &c.x_));
        c.row_offsets_ = row_offsets_;
        c.col_indices_ = col_indices_;
        c.obs_names_ = obs_names_;
        c.var_names_ = var_names_;
        c.obsm_ = obsm_;
        c.varm_ = varm_;
        c.var_bool_ = var_bool_;
        c.obs_cols_ = obs_cols_;
        c.var_cols_ = var_cols_;
        return c;
    }
    // a matrix the library produced (filter / subset): the host copy of the pattern is downloaded
    static IMAnnData from_device(Context& ctx, srx_mat* x, std::vector<std::string> obs_names,
                                 std::vector<std::string> var_names) {
        IMAnnData a(ctx);
        a.x_ = x;
        auto i = a.info();
        a.row_offsets_.resize(i.n_rows + 1);
        a.col_indices_.resize(i.nnz);
        ctx.check(srx_matrix_download_pattern(x, a.row_offsets_.data(), a.col_indices_.data()));
        a.obs_names_ = std::move(obs_names);
        a.var_names_ = std::move(var_names);
        return a;
    }
    const std::vector<std::string>& obs_names() const { return obs_names_; }
    const std::vector<std::string>& var_names() const { return var_names_; }
    // numeric columns of the `obs` / `var` DataFrames (what qc_vars_inplace writes)
    std::map<std::string, std::vector<double>>& obs_columns() { return obs_cols_; }
    std::map<std::string, std::vector<double>>& var_columns() { return var_cols_; }
    // boolean columns of the `var` DataFrame (what FeatureSelection::HighlyVariableCol reads)
    std::map<std::string, std::vector<bool>>& var_bool() { return var_bool_; }
    const std::map<std::string, std::vector<bool>>& var_bool() const { return var_bool_; }
    std::map<std::string, Array2>& obsm() { return obsm_; }
    std::map<std::string, Array2>& varm() { return varm_; }

private:
    explicit IMAnnData(Context& ctx) : ctx_(&ctx) {}
    Context* ctx_;
    srx_mat* x_ = nullptr;
    std::vector<std::uint64_t> row_offsets_, col_indices_;
    std::vector<std::string> obs_names_, var_names_;
    std::map<std::string, Array2> obsm_, varm_;
    std::map<std::string, std::vector<bool>> var_bool_;
    std::map<std::string, std::vector<double>> obs_cols_, var_cols_;
};

namespace memory {

namespace statistics {                                           // src/memory/statistics/mod.rs:10-46
inline std::size_t len(const IMAnnData& a, Direction d) { return d == Direction::Row ? a.n_obs() : a.n_vars(); }
inline std::vector<std::uint32_t> compute_number(const IMAnnData& a, Direction d) {
    std::vector<std::uint32_t> v(len(a, d));
    a.ctx().check(srx_compute_number(a.x(), (int)d, v.data()));
    return v;
}
inline std::vector<double> compute_sum(const IMAnnData& a, Direction d) {
    std::vector<double> v(len(a, d));
    a.ctx().check(srx_compute_sum(a.x(), (int)d, v.data()));
    return v;
}
inline std::vector<double> compute_variance(const IMAnnData& a, Direction d) {
    std::vector<double> v(len(a, d));
    a.ctx().check(srx_compute_variance(a.x(), (int)d, v.data()));
    return v;
}
inline std::vector<double> compute_std_dev(const IMAnnData& a, Direction d) {
    std::vector<double> v(len(a, d));
    a.ctx().check(srx_compute_std_dev(a.x(), (int)d, v.data()));
    return v;
}
inline std::pair<std::vector<double>, std::vector<double>> compute_min_max(const IMAnnData& a, Direction d) {
    std::vector<double> mn(len(a, d)), mx(len(a, d));
    a.ctx().check(srx_compute_min_max(a.x(), (int)d, mn.data(), mx.data()));
    return {mn, mx};
}
// StatisticsContainer (src/memory/statistics/structs/mod.rs:1-10) and compute_qc_variables (mod.rs:48-72)
struct StatisticsContainer {
    std::vector<std::uint32_t> num_per_cell, num_per_gene;
    std::vector<double> expr_per_gene, expr_per_cell, variance_per_gene, variance_per_cell, std_dev_per_cell,
        std_dev_per_gene;
};
inline StatisticsContainer compute_qc_variables(const IMAnnData& a) {
    const std::size_t n = a.n_obs(), g = a.n_vars();
    StatisticsContainer c;
    c.num_per_cell.resize(n); c.num_per_gene.resize(g);
    c.expr_per_gene.resize(g); c.expr_per_cell.resize(n);
    c.variance_per_gene.resize(g); c.variance_per_cell.resize(n);
    c.std_dev_per_cell.resize(n); c.std_dev_per_gene.resize(g);
    a.ctx().check(srx_compute_qc_variables(a.x(), c.num_per_cell.data(), c.num_per_gene.data(), c.expr_per_gene.data(),
                                           c.expr_per_cell.data(), c.variance_per_gene.data(), c.variance_per_cell.data(),
                                           c.std_dev_per_cell.data(), c.std_dev_per_gene.data()));
    return c;
}
// qc_vars_inplace (src/memory/statistics/mod.rs:74-103): the eight vectors become columns of obs / var under the
// reference's names (counts are widened to f64: the mirror's DataFrames hold f64 columns)
inline void qc_vars_inplace(IMAnnData& a) {
    const StatisticsContainer d = compute_qc_variables(a);
    auto widen = [](const std::vector<std::uint32_t>& v) { return std::vector<double>(v.begin(), v.end()); };
    a.obs_columns()["num_genes_per_cell"] = widen(d.num_per_cell);
    a.obs_columns()["sum_expr_per_cell"] = d.expr_per_cell;
    a.obs_columns()["var_expr_per_cell"] = d.variance_per_cell;
    a.obs_columns()["std_dev_per_cell"] = d.std_dev_per_cell;
    a.var_columns()["num_cells_per_gene"] = widen(d.num_per_gene);
    a.var_columns()["sum_expr_per_gene"] = d.expr_per_gene;
    a.var_columns()["var_expr_per_gene"] = d.variance_per_gene;
    a.var_columns()["std_dev_per_gene"] = d.std_dev_per_gene;
}
}  // namespace statistics

namespace processing {                                           // src/memory/processing/mod.rs:303-332
inline void normalize_total_inplace(IMAnnData& a, double target_sum, Direction d) {
    a.ctx().check(srx_normalize_total_inplace(a.x(), target_sum, (int)d));
}
inline IMAnnData normalize_total(const IMAnnData& a, double target_sum, Direction d) {
    IMAnnData n = a.deep_clone();
    normalize_total_inplace(n, target_sum, d);
    return n;
}
inline void log1p_transform_inplace(IMAnnData& a) { a.ctx().check(srx_log1p_inplace(a.x())); }
inline IMAnnData log1p_transform(const IMAnnData& a) {
    IMAnnData n = a.deep_clone();
    log1p_transform_inplace(n);
    return n;
}

// filter_cells / filter_genes (processing/mod.rs:118-146, :271-299) and their in-place forms (:86-116, :245-269)
inline IMAnnData filter_impl(const IMAnnData& a, const FlexValue& lo, const FlexValue& hi, bool genes) {
    const std::size_t n = genes ? a.n_vars() : a.n_obs();
    std::vector<std::uint8_t> mask(n ? n : 1);
    srx_mat* out = nullptr;
    a.ctx().check(genes ? srx_filter_genes(a.x(), lo.c, hi.c, &out, mask.data())
                        : srx_filter_cells(a.x(), lo.c, hi.c, &out, mask.data()));
    std::vector<std::string> obs, var;
    for (std::size_t i = 0; i < a.obs_names().size(); ++i)
        if (genes || mask[i]) obs.push_back(a.obs_names()[i]);
    for (std::size_t j = 0; j < a.var_names().size(); ++j)
        if (!genes || mask[j]) var.push_back(a.var_names()[j]);
    return IMAnnData::from_device(a.ctx(), out, std::move(obs), std::move(var));
}
inline IMAnnData filter_cells(const IMAnnData& a, const FlexValue& lo, const FlexValue& hi) { return filter_impl(a, lo, hi, false); }
inline IMAnnData filter_genes(const IMAnnData& a, const FlexValue& lo, const FlexValue& hi) { return filter_impl(a, lo, hi, true); }
inline void filter_cells_inplace(IMAnnData& a, const FlexValue& lo, const FlexValue& hi) { a = filter_impl(a, lo, hi, false); }
inline void filter_genes_inplace(IMAnnData& a, const FlexValue& lo, const FlexValue& hi) { a = filter_impl(a, lo, hi, true); }

namespace dim_red {                                              // src/memory/processing/dim_red/mod.rs
inline std::vector<std::uint64_t> select_features(const IMAnnData& a, const FeatureSelection& fs) {   // :123-156
    switch (fs.kind) {
        case FeatureSelection::HighlyVariableK: {
            std::vector<std::uint64_t> idx(std::min<std::size_t>(fs.n, a.n_vars()));
            std::uint64_t n_out = 0;
            a.ctx().check(srx_select_hvg(a.x(), fs.n, idx.data(), &n_out));
            idx.resize(n_out);
            return idx;
        }
        case FeatureSelection::VarianceThresholdK: {
            auto var = statistics::compute_variance(a, Direction::Column);
            std::vector<std::uint64_t> idx;
            for (std::size_t i = 0; i < var.size(); ++i)
                if (var[i] > fs.threshold) idx.push_back(i);
            return idx;
        }
        case FeatureSelection::NoneK: {
            std::vector<std::uint64_t> idx(a.n_vars());
            for (std::size_t i = 0; i < idx.size(); ++i) idx[i] = i;
            return idx;
        }
        case FeatureSelection::HighlyVariableColK: {                       // :125-134: indices where the bool column is true
            auto it = a.var_bool().find(fs.col);
            if (it == a.var_bool().end()) throw Error(SRX_E_ARG, "Error accessing column '" + fs.col + "' : not found");
            std::vector<std::uint64_t> idx;
            for (std::size_t i = 0; i < it->second.size(); ++i)
                if (it->second[i]) idx.push_back(i);
            return idx;
        }
        case FeatureSelection::RandomizedK: {                              // :141-146: shuffle of 0..n_vars, first n (thread_rng)
            std::vector<std::uint64_t> idx(a.n_vars());
            for (std::size_t i = 0; i < idx.size(); ++i) idx[i] = i;
            std::random_device rd;
            std::mt19937_64 rng(rd());
            std::shuffle(idx.begin(), idx.end(), rng);
            idx.resize(std::min<std::size_t>(fs.n, idx.size()));
            return idx;
        }
    }
    throw Error(SRX_E_ARG, "unknown FeatureSelection");
}
// single_algebra's SVDImplementation markers (dim_red/mod.rs:12,24): accepted, as in the reference's signature, and
// ignored — the device solver has no SVD backend to choose.
struct FaerSVD {};
struct LapackSVD {};
// pca_inplace<S: SVDImplementation>(anndata, n_components, center, scale, n_threads, feature_selection, svd_mode)
// (:24-94), argument for argument.  Stores obsm["X_pca"] (:105-106) like attach_pca_results; returns the solver's info
// (the reference returns Ok(())).
template <typename S = FaerSVD>
inline srx_pca_info pca_inplace(IMAnnData& a, std::optional<std::size_t> n_components, std::optional<bool> center,
                                std::optional<bool> scale, std::optional<std::size_t> n_threads,
                                const FeatureSelection& fs, S /*svd_mode*/ = S{}) {
    auto sel = select_features(a, fs);
    srx_pca_opts o{};
    o.n_components = n_components ? (int)*n_components : -1;
    o.center = center ? (int)*center : -1;
    o.scale = scale ? (int)*scale : -1;
    o.n_threads = n_threads ? (int)*n_threads : -1;
    const std::size_t n_pc = std::min<std::size_t>(n_components.value_or(2), sel.size());    // :52
    Array2 scores;
    scores.nrows = a.n_obs();
    scores.ncols = n_pc;
    scores.data.resize(scores.nrows * n_pc);
    srx_pca_info info{};
    a.ctx().check(srx_pca(a.x(), sel.data(), sel.size(), &o, scores.data.data(), nullptr, nullptr, nullptr, nullptr,
                          &info));
    a.obsm()["X_pca"] = std::move(scores);
    return info;
}
}  // namespace dim_red
}  // namespace processing
}  // namespace memory

// ---- single_rust::backed (src/backed/mod.rs): the matrix is not resident, it is visited as row chunks ----------
struct ComputationMode {                                          // src/shared/mod.rs:25-28
    enum Kind { ChunkedK, WholeK } kind = WholeK;
    std::size_t size = 0;
    static ComputationMode Chunked(std::size_t n) { ComputationMode m; m.kind = ChunkedK; m.size = n; return m; }
    static ComputationMode Whole() { return ComputationMode(); }
};

namespace backed {

// The CSR group of an h5ad `X` (indptr / indices / data) held by the caller: borrowed host arrays (a memory-mapped
// flat store, or the in-memory arrays).  `iter(chunk)` visits consecutive row ranges like ArrayElemOp::iter.
class BackedAnnData {
public:
    BackedAnnData(Context& ctx, std::size_t n_obs, std::size_t n_vars, const std::uint64_t* row_offsets,
                  const std::uint64_t* col_indices, const void* values, int dtype)
        : ctx_(&ctx), n_obs_(n_obs), n_vars_(n_vars), ip_(row_offsets), ix_(col_indices), v_(values), dtype_(dtype) {}
    template <typename T>
    static BackedAnnData of(Context& ctx, const CsrMatrix<T>& x) {
        return BackedAnnData(ctx, x.nrows, x.ncols, x.row_offsets.data(), x.col_indices.data(), x.values.data(), DTypeOf<T>::v);
    }
    std::size_t n_obs() const { return n_obs_; }
    std::size_t n_vars() const { return n_vars_; }
    Context& ctx() const { return *ctx_; }
    // rows [start, end) as an srx_csr tile: a window of the row offsets, nothing copied
    srx_csr tile(std::size_t start, std::size_t end) const {
        static const std::size_t width[] = {1, 2, 4, 1, 2, 4, 4, 8};      // SRX_I8 .. SRX_F64
        const std::uint64_t lo = ip_[start];
        return srx_csr{end - start, n_vars_, ip_[end] - lo, ip_ + start, ix_ + lo,
                       const_cast<char*>(static_cast<const char*>(v_)) + lo * width[dtype_], dtype_};
    }

private:
    Context* ctx_;
    std::size_t n_obs_, n_vars_;
    const std::uint64_t *ip_, *ix_;
    const void* v_;
    int dtype_;
};

class Session {                                                   // owning wrapper of an srx_backed
public:
    Session(Context& ctx, std::size_t n_cols, int store = SRX_STORE_AUTO) : ctx_(&ctx) {
        ctx.check(srx_backed_create(ctx.handle(), n_cols, store, &h_));
    }
    ~Session() { srx_backed_destroy(h_); }
    Session(const Session&) = delete;
    Session& operator=(const Session&) = delete;
    srx_backed* handle() const { return h_; }

private:
    Context* ctx_;
    srx_backed* h_ = nullptr;
};

namespace statistics {                                            // src/backed/statistics/mod.rs:5-45
namespace detail {
inline void sweep(const BackedAnnData& a, std::size_t size, Session& s, std::uint32_t* row_num, double* row_sum) {
    if (size == 0) throw Error(SRX_E_ARG, "ComputationMode::Chunked(0)");
    for (std::size_t start = 0; start < a.n_obs(); start += size) {
        const std::size_t end = std::min(start + size, a.n_obs());
        srx_csr t = a.tile(start, end);
        // every chunk writes at ITS rows (the reference's chunk helpers use the chunk-local index, csr.rs:57-62)
        a.ctx().check(srx_backed_stats_tile(s.handle(), &t, 0.0, 0, row_num ? row_num + start : nullptr,
                                            row_sum ? row_sum + start : nullptr));
    }
}
}  // namespace detail

inline std::vector<std::uint32_t> compute_number(const BackedAnnData& a, Direction d, const ComputationMode& mode) {
    const std::size_t chunk = mode.kind == ComputationMode::ChunkedK ? mode.size : std::max<std::size_t>(a.n_obs(), 1);
    Session s(a.ctx(), a.n_vars());
    std::vector<std::uint32_t> out(d == Direction::Row ? a.n_obs() : a.n_vars());
    detail::sweep(a, chunk, s, d == Direction::Row ? out.data() : nullptr, nullptr);
    if (d == Direction::Column) {
        std::vector<std::uint64_t> cnt(a.n_vars());
        a.ctx().check(srx_backed_moments(s.handle(), cnt.data(), nullptr, nullptr, nullptr));
        for (std::size_t j = 0; j < cnt.size(); ++j) out[j] = (std::uint32_t)cnt[j];
    }
    return out;
}

inline std::vector<double> compute_sum(const BackedAnnData& a, Direction d, const ComputationMode& mode) {
    const std::size_t chunk = mode.kind == ComputationMode::ChunkedK ? mode.size : std::max<std::size_t>(a.n_obs(), 1);
    Session s(a.ctx(), a.n_vars());
    std::vector<double> out(d == Direction::Row ? a.n_obs() : a.n_vars());
    detail::sweep(a, chunk, s, nullptr, d == Direction::Row ? out.data() : nullptr);
    if (d == Direction::Column) a.ctx().check(srx_backed_moments(s.handle(), nullptr, out.data(), nullptr, nullptr));
    return out;
}
}  // namespace statistics

namespace processing {
// The whole path over row chunks (the reference's backed::processing is empty; include/srx.h "backed mode").
struct PcaResult {
    Array2 x_pca;                      // n_obs x n_pc
    Array2 components;                 // k x n_pc, rows in selection order
    std::vector<double> explained_variance_ratio, mean, std_;
    std::vector<std::uint64_t> selected;
    srx_pca_info info{};
};
inline PcaResult pca_pipeline(const BackedAnnData& a, std::size_t chunk, double target_sum, std::size_t n_hvg,
                              int n_components, int store = SRX_STORE_AUTO) {
    if (chunk == 0) throw Error(SRX_E_ARG, "chunk size 0");
    Context& ctx = a.ctx();
    Session s(ctx, a.n_vars(), store);
    const int tf = SRX_BACKED_NORMALIZE | SRX_BACKED_LOG1P;
    for (std::size_t start = 0; start < a.n_obs(); start += chunk) {
        srx_csr t = a.tile(start, std::min(start + chunk, a.n_obs()));
        ctx.check(srx_backed_stats_tile(s.handle(), &t, target_sum, tf, nullptr, nullptr));
    }
    srx_pca_opts o{};
    o.n_components = n_components;
    o.center = o.scale = -1;
    std::uint64_t k = 0;
    ctx.check(srx_backed_select(s.handle(), n_hvg, nullptr, 0, &o, nullptr, &k));
    for (std::size_t start = 0; start < a.n_obs(); start += chunk) {
        srx_csr t = a.tile(start, std::min(start + chunk, a.n_obs()));
        ctx.check(srx_backed_gram_tile(s.handle(), &t, target_sum, tf));
    }
    PcaResult r;
    ctx.check(srx_backed_solve(s.handle(), &r.info));
    const std::size_t n_pc = r.info.n_pc;
    r.x_pca = Array2{a.n_obs(), n_pc, std::vector<double>(a.n_obs() * n_pc)};
    r.components = Array2{(std::size_t)k, n_pc, std::vector<double>(k * n_pc)};
    r.explained_variance_ratio.resize(n_pc);
    r.mean.resize(k);
    r.std_.resize(k);
    r.selected.resize(k);
    ctx.check(srx_backed_fetch(s.handle(), r.x_pca.data.data(), r.components.data.data(), r.explained_variance_ratio.data(),
                               r.mean.data(), r.std_.data(), r.selected.data()));
    return r;
}
}  // namespace processing
}  // namespace backed
}  // namespace single_rust
