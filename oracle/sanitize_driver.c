/*
 * sanitize_driver.c — walks every entry point of the oracle (srx_oracle.c, TEST INFRASTRUCTURE ONLY) under
 * AddressSanitizer + UndefinedBehaviorSanitizer: the 4 x 5 known-answer matrix of SURVEY.md 8(c), empty rows / empty
 * genes, and random matrices of every dtype.  `make -C oracle sanitize`; exits non-zero on a finding or a wrong KAT value.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "srx_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

static size_t dsize(int dt) {
    switch (dt) {
        case ORC_I8: case ORC_U8: return 1;
        case ORC_I16: case ORC_U16: return 2;
        case ORC_I32: case ORC_U32: case ORC_F32: return 4;
        default: return 8;
    }
}
static void put(void* vals, int dt, uint64_t i, int v) {
    switch (dt) {
        case ORC_I8: ((int8_t*)vals)[i] = (int8_t)v; break;
        case ORC_I16: ((int16_t*)vals)[i] = (int16_t)v; break;
        case ORC_I32: ((int32_t*)vals)[i] = v; break;
        case ORC_U8: ((uint8_t*)vals)[i] = (uint8_t)v; break;
        case ORC_U16: ((uint16_t*)vals)[i] = (uint16_t)v; break;
        case ORC_U32: ((uint32_t*)vals)[i] = (uint32_t)v; break;
        case ORC_F32: ((float*)vals)[i] = (float)v; break;
        default: ((double*)vals)[i] = (double)v; break;
    }
}

static int walk(const orc_csr* m) {
    const uint64_t N = m->n_rows, G = m->n_cols, nnz = m->nnz;
    uint32_t* nr = malloc((N + 1) * sizeof *nr);
    uint32_t* nc = malloc((G + 1) * sizeof *nc);
    double* fr = malloc((N + 1) * sizeof *fr);
    double* fc = malloc((G + 1) * sizeof *fc);
    double* fr2 = malloc((N + 1) * sizeof *fr2);
    double* fc2 = malloc((G + 1) * sizeof *fc2);
    double* scaled = malloc((nnz + 1) * sizeof *scaled);
    double* logged = malloc((nnz + 1) * sizeof *logged);
    uint64_t* cnt = malloc((G + 1) * sizeof *cnt);
    uint64_t* sel = malloc((G + 1) * sizeof *sel);
    int rc = 0;
    rc |= orc_number(m, ORC_ROW, nr) | orc_number(m, ORC_COLUMN, nc);
    rc |= orc_sum(m, ORC_ROW, fr) | orc_sum(m, ORC_COLUMN, fc);
    rc |= orc_variance(m, ORC_ROW, fr) | orc_variance(m, ORC_COLUMN, fc);
    rc |= orc_std_dev(m, ORC_ROW, fr) | orc_std_dev(m, ORC_COLUMN, fc);
    rc |= orc_min_max(m, ORC_ROW, fr, fr2) | orc_min_max(m, ORC_COLUMN, fc, fc2);
    rc |= orc_gene_moments(m, cnt, fc, fc2);
    rc |= orc_normalize_total(m, 1e4, ORC_COLUMN, scaled);
    rc |= orc_normalize_total(m, 1e4, ORC_ROW, scaled);
    orc_csr n64 = *m;
    n64.values = scaled;
    n64.dtype = ORC_F64;
    rc |= orc_log1p(&n64, logged);
    n64.values = logged;
    rc |= orc_variance(&n64, ORC_COLUMN, fc);
    uint64_t n_out = 0;
    const uint64_t want = G < 7 ? G : 7;
    rc |= orc_select_hvg(fc, G, want, sel, &n_out);
    if (n_out != want) rc |= 64;
    double* dense = malloc((N * n_out + 1) * sizeof *dense);
    rc |= orc_densify_selected(&n64, sel, n_out, dense);
    if (m->dtype == ORC_F32) {                      /* the F32-stays-F32 arm of log1p */
        float* lf = malloc((nnz + 1) * sizeof *lf);
        rc |= orc_log1p(m, lf);
        free(lf);
    }
    free(dense); free(nr); free(nc); free(fr); free(fc); free(fr2); free(fc2); free(scaled); free(logged); free(cnt); free(sel);
    return rc;
}

int main(void) {
    /* the known-answer matrix */
    const uint64_t ip[5] = {0, 2, 3, 3, 6}, ix[6] = {1, 3, 0, 0, 1, 2};
    double v[6] = {3, 1, 2, 1, 1, 4};
    orc_csr kat = {4, 5, 6, ip, ix, v, ORC_F64};
    double rs[4], out[6];
    if (orc_sum(&kat, ORC_ROW, rs) || rs[0] != 4 || rs[1] != 2 || rs[2] != 0 || rs[3] != 6) return 2;
    if (orc_normalize_total(&kat, 1e4, ORC_ROW, out) || out[0] != 7500.0 || out[2] != 10000.0) return 3;
    if (walk(&kat)) return 4;
    /* random matrices of every dtype, with empty rows and empty genes */
    for (int dt = ORC_I8; dt <= ORC_F64; ++dt) {
        for (int rep = 0; rep < 6; ++rep) {
            const uint64_t N = 1 + rnd() % 60, G = 1 + rnd() % 50;
            uint64_t* indptr = calloc(N + 1, sizeof *indptr);
            uint64_t* indices = malloc((N * G + 1) * sizeof *indices);
            void* vals = malloc((N * G + 1) * dsize(dt));
            uint64_t nnz = 0;
            for (uint64_t r = 0; r < N; ++r) {
                if (rnd() % 5) {
                    for (uint64_t c = 0; c < G; ++c) {
                        if (c % 7 == 3) continue;                    /* genes that stay empty */
                        if (rnd() % 4 == 0) {
                            indices[nnz] = c;
                            put(vals, dt, nnz, 1 + (int)(rnd() % 50));
                            ++nnz;
                        }
                    }
                }
                indptr[r + 1] = nnz;
            }
            orc_csr m = {N, G, nnz, indptr, indices, vals, dt};
            const int rc = walk(&m);
            free(indptr); free(indices); free(vals);
            if (rc) {
                fprintf(stderr, "oracle walk failed: dtype %d rc %d\n", dt, rc);
                return 5;
            }
        }
    }
    puts("oracle sanitize walk: ok");
    return 0;
}
