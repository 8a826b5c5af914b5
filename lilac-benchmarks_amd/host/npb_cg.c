/*
 * host/npb_cg.c -- NPB CG caller counterpart: matrix generator + benchmark loop.
 * See npb_cg.h for scope; reference lines are cited next to each piece.
 *
 * Bit-compatibility notes (what makes the generated matrix identical to the
 * Fortran one, value for value):
 *   - randlc is exact integer arithmetic mod 2^46 (common/randi8.f:1-35);
 *   - every generated value is  aelt(nzrow,i) * (size * aelt(nza,i))  with the
 *     same association as cg.f:828-831, plus "(va + rcond) - shift" on the
 *     diagonal of the generating vector (cg.f:835-837);
 *   - duplicates are summed in generation order (ascending i), starting from
 *     0.0 (cg.f:848-865).  We bucket the triples by row in generation order,
 *     sort each row by (column, sequence number) -- a stable order by
 *     construction -- and add runs of equal columns front to back, which is the
 *     same sequence of additions;
 *   - size = ratio**(i-1) is accumulated by repeated multiplication
 *     (cg.f:818-819,870), ratio = rcond ** (1/n) through libm pow().
 * Compile with -ffp-contract=off (the Makefile does).
 */
#include "npb_cg.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* NPB3.3.1/sys/setparams.c:551-575 and the verification table cg.f:122-163 */
static const npb_class k_classes[] = {
    {'S', 1400, 7, 15, 10.0, 0.1, 8.5971775078648},
    {'W', 7000, 8, 15, 12.0, 0.1, 10.362595087124},
    {'A', 14000, 11, 15, 20.0, 0.1, 17.130235054029},
    {'B', 75000, 13, 75, 60.0, 0.1, 22.712745482631},
    {'C', 150000, 15, 75, 110.0, 0.1, 28.973605592845},
    {'D', 1500000, 21, 100, 500.0, 0.1, 52.514532105794},
    {'E', 9000000, 26, 100, 1500.0, 0.1, 77.522164599383},
};

int npb_class_lookup(char name, npb_class *out)
{
    for (size_t k = 0; k < sizeof k_classes / sizeof k_classes[0]; ++k)
        if (k_classes[k].name == name) {
            *out = k_classes[k];
            return 0;
        }
    return -1;
}

/* common/randi8.f:1-35 */
static inline double lcg_next(uint64_t *state, uint64_t mult)
{
    *state = (*state * mult) & ((1ull << 46) - 1);
    return 0x1p-46 * (double)*state;
}

static int cmp_u64(const void *pa, const void *pb)
{
    uint64_t a = *(const uint64_t *)pa, b = *(const uint64_t *)pb;
    return a < b ? -1 : (a > b);
}

void npb_matrix_free(npb_matrix *m)
{
    free(m->rowstr);
    free(m->colidx);
    free(m->a);
    memset(m, 0, sizeof *m);
}

/*
 * Generates rows [*rb,*re) of the class matrix (0-based, half open).  With nparts > 0 the range
 * is chosen here: part `part` of `nparts` contiguous row blocks of about equal generated-triple
 * counts (the pre-deduplication proxy for nnz), and returned through rb/re.  Only the triples of
 * the wanted rows are materialised, so a row block of class D costs 1/nparts of the memory and
 * sort time; the random stream itself is always run in full (it is O(na * nonzer)).
 */
static int makea_rows(char cls, int part, int nparts, int *rb, int *re, npb_matrix *out)
{
    memset(out, 0, sizeof *out);
    if (npb_class_lookup(cls, &out->cls))
        return -1;
    const double t0 = now_s();
    const int n = out->cls.na, nonzer = out->cls.nonzer;
    const double rcond = out->cls.rcond, shift = out->cls.shift;
    const int w = nonzer + 1;
    if ((int64_t)n * w * w > 0x7fffffffLL)
        return -3; /* nz = na*(nonzer+1)^2 overflows the ABI's int (globals.h:80-82) */

    /* ---- the n sparse generating vectors (makea, cg.f:697-719) ---- */
    int *arow = malloc((size_t)n * sizeof *arow);
    int *acol = malloc((size_t)n * w * sizeof *acol);
    double *aelt = malloc((size_t)n * w * sizeof *aelt);
    int64_t *start = calloc((size_t)n + 1, sizeof *start);
    if (!arow || !acol || !aelt || !start)
        goto oom0;

    uint64_t seed = 314159265ull;         /* tran,  cg.f:186 */
    const uint64_t mult = 1220703125ull;  /* amult, cg.f:187 */
    (void)lcg_next(&seed, mult);          /* zeta = randlc(tran, amult), cg.f:188 */

    int nn1 = 1; /* smallest power of two >= n, at least 2 (cg.f:701-704) */
    do
        nn1 *= 2;
    while (nn1 < n);

    for (int i = 0; i < n; ++i) {
        int *ic = acol + (size_t)i * w;
        double *vc = aelt + (size_t)i * w;
        int nzv = 0;
        while (nzv < nonzer) { /* sprnvc, cg.f:936-962 */
            double vecelt = lcg_next(&seed, mult);
            double vecloc = lcg_next(&seed, mult);
            int col = (int)(nn1 * vecloc) + 1; /* icnvrt, cg.f:986 */
            if (col > n)
                continue;
            int seen = 0;
            for (int k = 0; k < nzv; ++k)
                seen |= (ic[k] == col);
            if (seen)
                continue;
            vc[nzv] = vecelt;
            ic[nzv] = col;
            ++nzv;
        }
        /* vecset(n, vc, ivc, nzv, iouter, .5), cg.f:1008-1018 */
        int set = 0;
        for (int k = 0; k < nzv; ++k)
            if (ic[k] == i + 1) {
                vc[k] = 0.5;
                set = 1;
            }
        if (!set) {
            vc[nzv] = 0.5;
            ic[nzv] = i + 1;
            ++nzv;
        }
        arow[i] = nzv;
    }

    /* ---- triples per row (sparse, cg.f:772-784) ---- */
    for (int i = 0; i < n; ++i)
        for (int e = 0; e < arow[i]; ++e)
            start[acol[(size_t)i * w + e]] += arow[i]; /* row j (1-based) counted at start[j] */
    /* start[j] currently = count of row j (1-based) ; turn into offsets of 0-based rows */
    {
        int64_t run = 0;
        for (int j = 1; j <= n; ++j) {
            int64_t c = start[j];
            start[j - 1] = run;
            run += c;
        }
        start[n] = run;
    }
    /* row range */
    int r0 = 0, r1 = n;
    if (nparts > 0) {
        const int64_t all = start[n];
        int lo = 0;
        for (int p = 0; p <= part; ++p) {
            r0 = lo;
            if (p == nparts - 1) {
                lo = n;
            } else {
                const int64_t target = all / nparts * (p + 1);
                while (lo < n && start[lo] < target)
                    ++lo;
            }
            r1 = lo;
        }
    } else if (rb && re) {
        r0 = *rb;
        r1 = *re;
        if (r0 < 0 || r1 > n || r0 > r1) {
            free(arow), free(acol), free(aelt), free(start);
            return -1;
        }
    }
    if (rb)
        *rb = r0;
    if (re)
        *re = r1;
    /* offsets relative to the first wanted row */
    const int64_t trip0 = start[r0];
    const int64_t ntrip = start[r1] - trip0;

    int *tcol = malloc((size_t)(ntrip ? ntrip : 1) * sizeof *tcol);
    double *tval = malloc((size_t)(ntrip ? ntrip : 1) * sizeof *tval);
    int64_t *pos = malloc((size_t)n * sizeof *pos);
    int *ucnt = malloc((size_t)n * sizeof *ucnt);
    if (!tcol || !tval || !pos || !ucnt)
        goto oom1;
    for (int j = 0; j < n; ++j)
        pos[j] = start[j] - trip0;

    /* ---- values in generation order (cg.f:818-871) ---- */
    {
        double size = 1.0;
        const double ratio = pow(rcond, 1.0 / (double)n);
        for (int i = 0; i < n; ++i) {
            const int *ic = acol + (size_t)i * w;
            const double *vc = aelt + (size_t)i * w;
            const int cnt = arow[i];
            for (int e = 0; e < cnt; ++e) {
                const int j = ic[e]; /* 1-based row */
                if (j - 1 < r0 || j - 1 >= r1)
                    continue;
                const double scale = size * vc[e];
                int64_t p = pos[j - 1];
                for (int f = 0; f < cnt; ++f) {
                    const int jcol = ic[f];
                    double va = vc[f] * scale;
                    if (jcol == j && j == i + 1)
                        va = va + rcond - shift;
                    tcol[p] = jcol;
                    tval[p] = va;
                    ++p;
                }
                pos[j - 1] = p;
            }
            size = size * ratio;
        }
    }
    free(arow), free(acol), free(aelt), free(pos);
    arow = NULL, acol = NULL, aelt = NULL, pos = NULL;

    /* ---- per row: order by (column, sequence), add duplicates front to back ---- */
    int oom = 0;
#pragma omp parallel
    {
        uint64_t *key = NULL;
        double *vtmp = NULL;
        size_t cap = 0;
#pragma omp for schedule(dynamic, 256)
        for (int j = r0; j < r1; ++j) {
            const int64_t s = start[j] - trip0;
            const size_t len = (size_t)(start[j + 1] - start[j]);
            if (len > cap) {
                free(key), free(vtmp);
                cap = len * 2 + 64;
                key = malloc(cap * sizeof *key);
                vtmp = malloc(cap * sizeof *vtmp);
                if (!key || !vtmp) {
                    cap = 0;
#pragma omp atomic write
                    oom = 1;
                }
            }
            if (cap == 0) {
                ucnt[j] = 0;
                continue;
            }
            for (size_t k = 0; k < len; ++k) {
                key[k] = ((uint64_t)(uint32_t)tcol[s + k] << 32) | (uint64_t)k;
                vtmp[k] = tval[s + k];
            }
            qsort(key, len, sizeof *key, cmp_u64);
            int u = 0;
            size_t k = 0;
            while (k < len) {
                const int col = (int)(key[k] >> 32);
                double acc = 0.0;
                while (k < len && (int)(key[k] >> 32) == col) {
                    acc = acc + vtmp[(uint32_t)key[k]];
                    ++k;
                }
                tcol[s + u] = col;
                tval[s + u] = acc;
                ++u;
            }
            ucnt[j] = u;
        }
        free(key), free(vtmp);
    }
    if (oom)
        goto oom1;

    /* ---- compact into CSR (cg.f:875-897) ---- */
    const int nloc = r1 - r0;
    out->rowstr = malloc(((size_t)nloc + 1) * sizeof *out->rowstr);
    if (!out->rowstr)
        goto oom1;
    {
        int64_t run = 1;
        for (int j = r0; j < r1; ++j) {
            out->rowstr[j - r0] = (int)run;
            run += ucnt[j];
        }
        out->rowstr[nloc] = (int)run;
        out->nnz = run - 1;
    }
    out->row_begin = r0;
    out->row_end = r1;
    out->colidx = malloc((size_t)(out->nnz ? out->nnz : 1) * sizeof *out->colidx);
    out->a = malloc((size_t)(out->nnz ? out->nnz : 1) * sizeof *out->a);
    if (!out->colidx || !out->a) {
        npb_matrix_free(out);
        goto oom1;
    }
#pragma omp parallel for schedule(static)
    for (int j = r0; j < r1; ++j) {
        const size_t dst = (size_t)out->rowstr[j - r0] - 1;
        memcpy(out->colidx + dst, tcol + (start[j] - trip0), (size_t)ucnt[j] * sizeof *tcol);
        memcpy(out->a + dst, tval + (start[j] - trip0), (size_t)ucnt[j] * sizeof *tval);
    }
    free(tcol), free(tval), free(ucnt), free(start);
    out->t_gen = now_s() - t0;
    return 0;

oom1:
    free(tcol), free(tval), free(pos), free(ucnt);
oom0:
    free(arow), free(acol), free(aelt), free(start);
    return -2;
}

int npb_makea(char cls, npb_matrix *out) { return makea_rows(cls, 0, 0, NULL, NULL, out); }

int npb_makea_part(char cls, int part, int nparts, npb_matrix *out)
{
    if (nparts < 1 || part < 0 || part >= nparts)
        return -1;
    int rb = 0, re = 0;
    return makea_rows(cls, part, nparts, &rb, &re, out);
}

int npb_makea_rows(char cls, int row_begin, int row_end, npb_matrix *out)
{
    return makea_rows(cls, 0, 0, &row_begin, &row_end, out);
}

/* ================================ the benchmark ================================== */

typedef struct cg_ctx {
    const npb_matrix *m;
    spmv_harness_fn spmv;
    double t_spmv;
    int64_t calls;
} cg_ctx;

static inline void do_spmv(cg_ctx *c, double *y, double *x)
{
    int rows = c->m->cls.na; /* rows = lastrow - firstrow + 1, cg.f:531 */
    double t0 = now_s();
    c->spmv(y, c->m->a, x, c->m->rowstr, c->m->colidx, &rows);
    c->t_spmv += now_s() - t0;
    c->calls++;
}

/* conj_grad, cg.f:447-644.  Vectors have na+2 entries like the Fortran COMMON (cg.f:75-80). */
static void conj_grad(cg_ctx *c, double *x, double *z, double *p, double *q, double *r, double *rnorm)
{
    const int n = c->m->cls.na;
    double rho = 0.0;
    for (int j = 0; j < n + 1; ++j) { /* cg.f:492-497 */
        q[j] = 0.0;
        z[j] = 0.0;
        r[j] = x[j];
        p[j] = r[j];
    }
    for (int j = 0; j < n; ++j) /* cg.f:504-506 */
        rho = rho + r[j] * r[j];

    for (int cgit = 0; cgit < 25; ++cgit) { /* cgitmax = 25, cg.f:489,513 */
        do_spmv(c, q, p);                   /* cg.f:532 */
        double d = 0.0;
        for (int j = 0; j < n; ++j) /* cg.f:573-576 */
            d = d + p[j] * q[j];
        const double alpha = rho / d;
        const double rho0 = rho;
        rho = 0.0;
        for (int j = 0; j < n; ++j) { /* cg.f:594-597 */
            z[j] = z[j] + alpha * p[j];
            r[j] = r[j] - alpha * q[j];
        }
        for (int j = 0; j < n; ++j) /* cg.f:603-605 */
            rho = rho + r[j] * r[j];
        const double beta = rho / rho0;
        for (int j = 0; j < n; ++j) /* cg.f:615-617 */
            p[j] = r[j] + beta * p[j];
    }
    do_spmv(c, r, z); /* cg.f:628 */
    double sum = 0.0;
    for (int j = 0; j < n; ++j) { /* cg.f:634-637 */
        const double d = x[j] - r[j];
        sum = sum + d * d;
    }
    *rnorm = sqrt(sum);
}

/* The CG vectors have static storage duration like the Fortran COMMON block /main_flt_mem/
 * (cg.f:69-80), dimensioned for the largest class behind the ABI (D: na = 1500000). */
#define NPB_NA_MAX 1500000
static double g_x[NPB_NA_MAX + 2], g_z[NPB_NA_MAX + 2], g_p[NPB_NA_MAX + 2], g_q[NPB_NA_MAX + 2],
    g_r[NPB_NA_MAX + 2];

int npb_cg_run(const npb_matrix *m, spmv_harness_fn spmv, int niter_override, int verbose,
               npb_result *out)
{
    memset(out, 0, sizeof *out);
    const int n = m->cls.na;
    const int niter = niter_override > 0 ? niter_override : m->cls.niter;
    if (niter > NPB_MAX_ITER || m->row_begin != 0 || m->row_end != n || n > NPB_NA_MAX)
        return -1; /* the benchmark loop needs the whole matrix */
    double *x = g_x, *z = g_z, *p = g_p, *q = g_q, *r = g_r;
    memset(x, 0, ((size_t)n + 2) * sizeof *x), memset(z, 0, ((size_t)n + 2) * sizeof *z);
    memset(p, 0, ((size_t)n + 2) * sizeof *p), memset(q, 0, ((size_t)n + 2) * sizeof *q);
    memset(r, 0, ((size_t)n + 2) * sizeof *r);
    cg_ctx c = {m, spmv, 0.0, 0};
    double rnorm = 0.0, zeta = 0.0;

    if (verbose) {
        printf("\n\n NAS Parallel Benchmarks (NPB3.3-SER) - CG Benchmark\n\n");
        printf(" Size: %11d\n", n);
        printf(" Iterations: %5d\n\n", niter);
    }

    const double t_init0 = now_s();
    for (int i = 0; i < n + 1; ++i) /* cg.f:216-218 */
        x[i] = 1.0;
    /* one untimed iteration (cg.f:233-272): this is where a caching backend uploads */
    conj_grad(&c, x, z, p, q, r, &rnorm);
    {
        double t1 = 0.0, t2 = 0.0;
        for (int j = 0; j < n; ++j) {
            t1 = t1 + x[j] * z[j];
            t2 = t2 + z[j] * z[j];
        }
        t2 = 1.0 / sqrt(t2);
        for (int j = 0; j < n; ++j)
            x[j] = t2 * z[j];
        (void)t1;
    }
    for (int i = 0; i < n + 1; ++i) /* cg.f:281-283 */
        x[i] = 1.0;
    zeta = 0.0;
    out->t_init = now_s() - t_init0;
    if (verbose)
        printf(" Initialization time = %15.3f seconds\n", out->t_init);

    c.t_spmv = 0.0;
    const double t0 = now_s();
    for (int it = 1; it <= niter; ++it) { /* cg.f:299-349 */
        conj_grad(&c, x, z, p, q, r, &rnorm);
        double t1 = 0.0, t2 = 0.0;
        for (int j = 0; j < n; ++j) {
            t1 = t1 + x[j] * z[j];
            t2 = t2 + z[j] * z[j];
        }
        t2 = 1.0 / sqrt(t2);
        zeta = m->cls.shift + 1.0 / t1;
        if (verbose) {
            if (it == 1)
                printf("\n   iteration           ||r||                 zeta\n");
            printf("    %5d       %20.14E%20.13f\n", it, rnorm, zeta);
        }
        out->zeta_hist[it - 1] = zeta;
        out->rnorm_hist[it - 1] = rnorm;
        for (int j = 0; j < n; ++j)
            x[j] = t2 * z[j];
    }
    out->t_bench = now_s() - t0;

    out->cls = m->cls.name;
    out->na = n;
    out->niter = niter;
    out->nnz = m->nnz;
    out->spmv_calls = c.calls;
    out->t_spmv = c.t_spmv;
    out->zeta = zeta;
    out->rnorm = rnorm;
    out->err = fabs(zeta - m->cls.zeta_verify) / m->cls.zeta_verify;           /* cg.f:366 */
    out->verified = (niter == m->cls.niter) && (out->err <= 1.0e-10);           /* cg.f:363-368 */
    if (out->t_bench > 0.0) {                                                   /* cg.f:395-402 */
        const double nz = (double)m->cls.nonzer * (double)(m->cls.nonzer + 1);
        out->mops = (double)(2.0 * niter * (double)n) * (3.0 + nz + 25.0 * (5.0 + nz) + 3.0) /
                    out->t_bench / 1000000.0;
    }
    return 0;
}

/* cg.f:193-402 with conj_grad on the device (see npb_cg.h): nparts = 0 one matrix on device 0, else row blocks */
static int run_device(const npb_matrix *m, const npb_cg_device_fns *f, int nparts, int ndevices, int niter_override, int verbose,
                      npb_result *out)
{
    memset(out, 0, sizeof *out);
    const int n = m->cls.na;
    const int niter = niter_override > 0 ? niter_override : m->cls.niter;
    if (niter > NPB_MAX_ITER || m->row_begin != 0 || m->row_end != n || nparts < 0 || nparts > 64)
        return -1;
    const double t_init0 = now_s();
    void *dms[64] = {0};
    void *cg = NULL;
    int ndm = 0;
    if (nparts == 0) {
        dms[0] = f->matrix_create(m->a, m->rowstr, m->colidx, n, 8, 0, 0, n, n);
        ndm = dms[0] ? 1 : 0;
        cg = dms[0] ? f->cg_create(dms[0]) : NULL;
    } else {
        int bounds[65];
        if (!f->partition_rows || !f->cg_create_multi || ndevices < 1)
            return -1;
        f->partition_rows(m->rowstr, n, nparts, bounds);
        for (int k = 0; k < nparts; ++k) {
            dms[k] = f->matrix_create(m->a, m->rowstr, m->colidx, n, 8, k % ndevices, bounds[k], bounds[k + 1], n);
            if (!dms[k])
                break;
            ++ndm;
        }
        cg = ndm == nparts ? f->cg_create_multi(dms, nparts) : NULL;
    }
    if (!cg) {
        for (int k = 0; k < ndm; ++k)
            f->matrix_destroy(dms[k]);
        return -3;
    }
    double rnorm = 0.0, zeta = 0.0, t1 = 0.0, t2 = 0.0;
    int rc = 0;
    /* one untimed iteration (cg.f:233-272) */
    rc |= f->cg_fill_x(cg, 1.0);
    rc |= f->cg_conj_grad(cg, 25, &rnorm);
    rc |= f->cg_norms(cg, &t1, &t2);
    rc |= f->cg_x_from_z(cg, 1.0 / sqrt(t2));
    rc |= f->cg_fill_x(cg, 1.0); /* cg.f:281-283 */
    out->t_init = now_s() - t_init0;
    if (verbose)
        printf(" Initialization time = %15.3f seconds\n", out->t_init);
    const double t0 = now_s();
    for (int it = 1; it <= niter && !rc; ++it) {
        rc |= f->cg_conj_grad(cg, 25, &rnorm);
        rc |= f->cg_norms(cg, &t1, &t2);
        t2 = 1.0 / sqrt(t2);
        zeta = m->cls.shift + 1.0 / t1;
        if (verbose) {
            if (it == 1)
                printf("\n   iteration           ||r||                 zeta\n");
            printf("    %5d       %20.14E%20.13f\n", it, rnorm, zeta);
        }
        out->zeta_hist[it - 1] = zeta;
        out->rnorm_hist[it - 1] = rnorm;
        rc |= f->cg_x_from_z(cg, t2);
    }
    out->t_bench = now_s() - t0;
    f->cg_destroy(cg);
    for (int k = 0; k < ndm; ++k)
        f->matrix_destroy(dms[k]);
    if (rc)
        return -4;
    out->cls = m->cls.name;
    out->na = n;
    out->niter = niter;
    out->nnz = m->nnz;
    out->spmv_calls = 26 * (int64_t)(niter + 1);
    out->t_spmv = 0.0;
    out->zeta = zeta;
    out->rnorm = rnorm;
    out->err = fabs(zeta - m->cls.zeta_verify) / m->cls.zeta_verify;
    out->verified = (niter == m->cls.niter) && (out->err <= 1.0e-10);
    if (out->t_bench > 0.0) {
        const double nz = (double)m->cls.nonzer * (double)(m->cls.nonzer + 1);
        out->mops = (double)(2.0 * niter * (double)n) * (3.0 + nz + 25.0 * (5.0 + nz) + 3.0) / out->t_bench /
                    1000000.0;
    }
    return 0;
}

int npb_cg_run_device(const npb_matrix *m, const npb_cg_device_fns *f, int niter_override, int verbose, npb_result *out)
{
    return run_device(m, f, 0, 1, niter_override, verbose, out);
}

int npb_cg_run_device_parts(const npb_matrix *m, const npb_cg_device_fns *f, int nparts, int ndevices, int niter_override,
                            int verbose, npb_result *out)
{
    if (nparts < 1)
        return -1;
    return run_device(m, f, nparts, ndevices, niter_override, verbose, out);
}

void npb_print_result(const npb_result *r)
{
    printf(" Benchmark completed \n");
    if (r->verified) {
        printf(" VERIFICATION SUCCESSFUL \n");
        printf(" Zeta is    %20.13E\n", r->zeta);
        printf(" Error is   %20.13E\n", r->err);
    } else {
        printf(" VERIFICATION FAILED\n");
        printf(" Zeta                %20.13E\n", r->zeta);
    }
    printf("\n\n CG Benchmark Completed.\n");
    printf(" Class           =             %12c\n", r->cls);
    printf(" Size            =             %12d\n", r->na);
    printf(" Iterations      =             %12d\n", r->niter);
    printf(" Time in seconds =             %12.2f\n", r->t_bench);
    printf(" Mop/s total     =             %12.2f\n", r->mops);
    printf(" Operation type  =           floating point\n");
    printf(" Verification    =             %s\n", r->verified ? "  SUCCESSFUL" : "UNSUCCESSFUL");
    printf(" SpMV calls      =             %12lld\n", (long long)r->spmv_calls);
    printf(" Time in SpMV    =             %12.4f\n", r->t_spmv);
}
