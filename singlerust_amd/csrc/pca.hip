// pca.hip — placeholder while the SpMM path is being written.
#include "common.hpp"
using namespace srx;
extern "C" {
int32_t srx_pca(srx_mat* m, const uint64_t*, uint64_t, const srx_pca_opts*, double*, double*, double*, double*, double*, srx_pca_info*) { return fail(m ? m->ctx : nullptr, SRX_E_ARG, "srx_pca: not built yet"); }
int32_t srx_pca_loadings(const double*, const double*, const uint64_t*, uint64_t, uint64_t, uint64_t, double*) { return fail(nullptr, SRX_E_ARG, "not built yet"); }
int32_t srx_pipeline(srx_mat* m, double, uint64_t, const srx_pca_opts*, srx_pipeline_result*) { return fail(m ? m->ctx : nullptr, SRX_E_ARG, "not built yet"); }
int32_t srx_result_fetch(srx_mat* m, double*, double*, double*, double*, double*, uint64_t*) { return fail(m ? m->ctx : nullptr, SRX_E_ARG, "not built yet"); }
}
