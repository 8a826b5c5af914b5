/*
 * oracle/npb_makea_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Literal CPU restatement of the NPB CG matrix generator, kept as the checker
 * for the fast generator in lilac-benchmarks_amd/host/npb_gen.c (which buckets
 * and sorts instead of inserting).  It follows the reference statement by
 * statement, including the O(nnz * rowlen) ordered insertion, so it is only
 * usable for classes S, W and A; that is its point: no cleverness to get wrong.
 *
 *   randlc   /root/reference/NPB3.3.1/common/randi8.f:1-35
 *   makea    /root/reference/NPB3.3.1/CG/cg.f:650-735
 *   sparse   /root/reference/NPB3.3.1/CG/cg.f:739-905
 *   sprnvc   /root/reference/NPB3.3.1/CG/cg.f:910-968
 *   icnvrt   /root/reference/NPB3.3.1/CG/cg.f:973-990
 *   vecset   /root/reference/NPB3.3.1/CG/cg.f:995-1019
 *   seeds    /root/reference/NPB3.3.1/CG/cg.f:186-188 (tran, amult and the one
 *            randlc call made before makea)
 *
 * Pinning: SURVEY.md section 8(a) row a7 records FNV-1a-64 checksums of
 * rowstr / colidx / a for classes S and A taken from the reference binary
 * (cg.f built with amdflang, reference native.so interposed); tests/
 * test_npb_gen.py asserts this restatement reproduces them, and that NPB's own
 * verification values (cg.f:122-163) are met when the CG driver runs on these
 * matrices.  The values of `a` depend on libm pow() (cg.f:819).
 *
 * Arrays are kept 1-based like the Fortran (index 0 unused) to keep the
 * correspondence with cg.f readable.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#pragma STDC FP_CONTRACT OFF

/* randi8.f:1-35 : x_{k+1} = a * x_k mod 2^46, returned scaled by 2^-46 */
static double o_randlc(double *x, double a)
{
    uint64_t lx = (uint64_t)(int64_t)*x;
    uint64_t la = (uint64_t)(int64_t)a;
    lx = (lx * la) & 0x00003FFFFFFFFFFFull;
    *x = (double)(int64_t)lx;
    return 0x1p-46 * (double)(int64_t)lx;
}

/* cg.f:910-968 */
static void o_sprnvc(int n, int nz, int nn1, double *v, int *iv, double *tran, double amult)
{
    int nzv = 0;
    while (nzv < nz) {
        double vecelt = o_randlc(tran, amult);
        double vecloc = o_randlc(tran, amult);
        int i = (int)(nn1 * vecloc) + 1; /* icnvrt, cg.f:986 */
        if (i > n)
            continue;
        int dup = 0;
        for (int ii = 1; ii <= nzv; ++ii)
            if (iv[ii] == i) {
                dup = 1;
                break;
            }
        if (dup)
            continue;
        ++nzv;
        v[nzv] = vecelt;
        iv[nzv] = i;
    }
}

/* cg.f:995-1019 */
static void o_vecset(double *v, int *iv, int *nzv, int i, double val)
{
    int set = 0;
    for (int k = 1; k <= *nzv; ++k)
        if (iv[k] == i) {
            v[k] = val;
            set = 1;
        }
    if (!set) {
        ++*nzv;
        v[*nzv] = val;
        iv[*nzv] = i;
    }
}

/*
 * Generates the class matrix into caller arrays (0-based storage of the
 * 1-based CSR the ABI wants: rowstr[0] == 1).  nzmax = na*(nonzer+1)^2
 * (cg.f/globals.h:80-82).  Returns nnz, or -1 on "space exceeded"
 * (cg.f:791-795), -2 on the internal error of cg.f:862.
 */
long npb_oracle_makea(int n, int nonzer, double rcond, double shift, long nzmax, double *a_out,
                      int *colidx_out, int *rowstr_out)
{
    double tran = 314159265.0, amult = 1220703125.0; /* cg.f:186-187 */
    (void)o_randlc(&tran, amult);                    /* cg.f:188 */

    const int w = nonzer + 1;
    int *arow = calloc((size_t)n + 1, sizeof *arow);
    int *acol = calloc(((size_t)n + 1) * (w + 1), sizeof *acol);
    double *aelt = calloc(((size_t)n + 1) * (w + 1), sizeof *aelt);
    int *ivc = calloc((size_t)w + 2, sizeof *ivc);
    double *vc = calloc((size_t)w + 2, sizeof *vc);
#define ACOL(e, i) acol[(size_t)(i) * (w + 1) + (e)]
#define AELT(e, i) aelt[(size_t)(i) * (w + 1) + (e)]

    int nn1 = 1; /* cg.f:701-704 */
    do {
        nn1 *= 2;
    } while (nn1 < n);

    for (int iouter = 1; iouter <= n; ++iouter) { /* cg.f:710-719 */
        int nzv = nonzer;
        o_sprnvc(n, nzv, nn1, vc, ivc, &tran, amult);
        o_vecset(vc, ivc, &nzv, iouter, 0.5);
        arow[iouter] = nzv;
        for (int e = 1; e <= nzv; ++e) {
            ACOL(e, iouter) = ivc[e];
            AELT(e, iouter) = vc[e];
        }
    }

    /* sparse, cg.f:739-905, nrows = n */
    long *rowstr = calloc((size_t)n + 3, sizeof *rowstr);
    for (int i = 1; i <= n; ++i)
        for (int nza = 1; nza <= arow[i]; ++nza) {
            int j = ACOL(nza, i) + 1;
            rowstr[j] += arow[i];
        }
    rowstr[1] = 1;
    for (int j = 2; j <= n + 1; ++j)
        rowstr[j] += rowstr[j - 1];
    long nza_total = rowstr[n + 1] - 1;
    if (nza_total > nzmax) {
        free(arow), free(acol), free(aelt), free(ivc), free(vc), free(rowstr);
        return -1;
    }

    double *a = calloc((size_t)nza_total + 2, sizeof *a);
    int *colidx = calloc((size_t)nza_total + 2, sizeof *colidx);
    long *nzloc = calloc((size_t)n + 2, sizeof *nzloc);

    double size = 1.0;
    double ratio = pow(rcond, 1.0 / (double)n); /* cg.f:819 */

    for (int i = 1; i <= n; ++i) {
        for (int nza = 1; nza <= arow[i]; ++nza) {
            int j = ACOL(nza, i);
            double scale = size * AELT(nza, i);
            for (int nzrow = 1; nzrow <= arow[i]; ++nzrow) {
                int jcol = ACOL(nzrow, i);
                double va = AELT(nzrow, i) * scale;
                if (jcol == j && j == i)
                    va = va + rcond - shift; /* cg.f:835-837 */
                long k;
                int found = 0;
                for (k = rowstr[j]; k <= rowstr[j + 1] - 1; ++k) {
                    if (colidx[k] > jcol) { /* ordered insert, cg.f:840-851 */
                        for (long kk = rowstr[j + 1] - 2; kk >= k; --kk)
                            if (colidx[kk] > 0) {
                                a[kk + 1] = a[kk];
                                colidx[kk + 1] = colidx[kk];
                            }
                        colidx[k] = jcol;
                        a[k] = 0.0;
                        found = 1;
                        break;
                    } else if (colidx[k] == 0) { /* cg.f:852-854 */
                        colidx[k] = jcol;
                        found = 1;
                        break;
                    } else if (colidx[k] == jcol) { /* duplicate, cg.f:855-860 */
                        nzloc[j] += 1;
                        found = 1;
                        break;
                    }
                }
                if (!found) {
                    free(arow), free(acol), free(aelt), free(ivc), free(vc), free(rowstr);
                    free(a), free(colidx), free(nzloc);
                    return -2;
                }
                a[k] = a[k] + va; /* cg.f:865 */
            }
        }
        size = size * ratio;
    }

    /* compaction, cg.f:875-897 */
    for (int j = 2; j <= n; ++j)
        nzloc[j] += nzloc[j - 1];
    for (int j = 1; j <= n; ++j) {
        long j1 = (j > 1) ? rowstr[j] - nzloc[j - 1] : 1;
        long j2 = rowstr[j + 1] - nzloc[j] - 1;
        long src = rowstr[j];
        for (long k = j1; k <= j2; ++k) {
            a[k] = a[src];
            colidx[k] = colidx[src];
            ++src;
        }
    }
    for (int j = 2; j <= n + 1; ++j)
        rowstr[j] -= nzloc[j - 1];
    long nnz = rowstr[n + 1] - 1;

    for (int j = 0; j <= n; ++j)
        rowstr_out[j] = (int)rowstr[j + 1];
    memcpy(a_out, a + 1, (size_t)nnz * sizeof *a);
    memcpy(colidx_out, colidx + 1, (size_t)nnz * sizeof *colidx); /* firstcol = 1: cg.f:207-211 is the identity */

    free(arow), free(acol), free(aelt), free(ivc), free(vc), free(rowstr);
    free(a), free(colidx), free(nzloc);
    return nnz;
#undef ACOL
#undef AELT
}
