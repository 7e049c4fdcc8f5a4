// Test harness: exposes singlerust_amd/csrc/smallmat.hpp (host-only dense algebra of the PCA
// driver) to the CPU test-suite through a tiny C ABI.  Built with g++ by tests/test_smallmat_cpu.py.
#include "../../singlerust_amd/csrc/smallmat.hpp"
extern "C" {
int t_sym_eig_desc(int n, const double* A, double* evals, double* evecs) {
    return srx::smallmat::sym_eig_desc(n, A, evals, evecs) ? 0 : 1;
}
int t_chol_upper_inverse(int n, int ld, const double* G, double* Rinv) {
    return srx::smallmat::chol_upper_inverse(n, ld, G, Rinv) ? 0 : 1;
}
}
