/*
 * host/callers.c -- see callers.h.  Reference lines are cited next to each piece.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include "callers.h"

#include <ctype.h>
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

void csr_d_free(csr_d *m)
{
    free(m->rowstr), free(m->colidx), free(m->a);
    memset(m, 0, sizeof *m);
}
void csr_f_free(csr_f *m)
{
    free(m->rowstr), free(m->colidx), free(m->a);
    memset(m, 0, sizeof *m);
}

/* ---- small seeded PRNG (xorshift64*) + Box-Muller ------------------------------------------- */
typedef struct rng {
    uint64_t s;
    int have;
    double spare;
} rng;
static uint64_t rng_u64(rng *r)
{
    r->s ^= r->s >> 12;
    r->s ^= r->s << 25;
    r->s ^= r->s >> 27;
    return r->s * 0x2545F4914F6CDD1Dull;
}
static double rng_uniform(rng *r) { return (double)(rng_u64(r) >> 11) * (1.0 / 9007199254740992.0); }
static double rng_gauss(rng *r, double mu, double sigma)
{
    if (r->have) {
        r->have = 0;
        return mu + sigma * r->spare;
    }
    double u, v, s;
    do {
        u = 2.0 * rng_uniform(r) - 1.0;
        v = 2.0 * rng_uniform(r) - 1.0;
        s = u * u + v * v;
    } while (s >= 1.0 || s == 0.0);
    const double f = sqrt(-2.0 * log(s) / s);
    r->spare = v * f;
    r->have = 1;
    return mu + sigma * u * f;
}

static int cmp_int(const void *a, const void *b)
{
    const int x = *(const int *)a, y = *(const int *)b;
    return x < y ? -1 : x > y;
}

/* ================================= SparseBench ============================================== */

int sb_read_crs(const char *path, csr_d *out)
{
    memset(out, 0, sizeof *out);
    FILE *f = fopen(path, "r");
    if (!f)
        return -1;
    long n = 0, nnz = 0;
    if (fscanf(f, "%ld %ld", &n, &nnz) != 2 || n <= 0 || nnz < 0) {
        fclose(f);
        return -2;
    }
    out->rows = out->cols = (int)n;
    out->nnz = nnz;
    out->rowstr = malloc(((size_t)n + 1) * sizeof(int));
    out->colidx = malloc((size_t)(nnz ? nnz : 1) * sizeof(int));
    out->a = malloc((size_t)(nnz ? nnz : 1) * sizeof(double));
    if (!out->rowstr || !out->colidx || !out->a) {
        fclose(f);
        csr_d_free(out);
        return -3;
    }
    for (long i = 0; i <= n; ++i)
        if (fscanf(f, "%d", &out->rowstr[i]) != 1) {
            fclose(f);
            csr_d_free(out);
            return -2;
        }
    if (out->rowstr[0] != 1 || out->rowstr[n] != nnz + 1) {
        fclose(f);
        csr_d_free(out);
        return -2; /* the header and the row pointers disagree (gen_crs.f:768-772 stops with "Strange crs read error") */
    }
    for (long i = 0; i < n; ++i)
        if (out->rowstr[i + 1] < out->rowstr[i]) {
            fclose(f);
            csr_d_free(out);
            return -2;
        }
    /* gen_crs.f:775-785: the entries are read in stream order, exactly nnz of them */
    for (long j = 0; j < nnz; ++j) {
        if (fscanf(f, "%d %lf", &out->colidx[j], &out->a[j]) != 2 || out->colidx[j] < 1 || out->colidx[j] > n) {
            fclose(f);
            csr_d_free(out);
            return -2; /* "Illegal matrix element", gen_crs.f:778-782 */
        }
    }
    fclose(f);
    return 0;
}

int sb_write_crs(const char *path, const csr_d *m, const int *extra_cols, const double *extra_vals,
                 int64_t n_extra)
{
    FILE *f = fopen(path, "w");
    if (!f)
        return -1;
    fprintf(f, "%12d%12lld\n", m->rows, (long long)m->nnz); /* big_gen.py:53 */
    for (int i = 0; i <= m->rows; ++i)
        fprintf(f, "%12d\n", m->rowstr[i]); /* :54-55 */
    for (int64_t j = 0; j < m->nnz; ++j)
        fprintf(f, "%12d %20.17f\n", m->colidx[j], m->a[j]); /* :56-57 */
    for (int64_t j = 0; j < n_extra; ++j) /* the surplus lines big_gen.py leaves behind */
        fprintf(f, "%12d %20.17f\n", extra_cols[j], extra_vals[j]);
    fclose(f);
    return 0;
}

int sb_generate(int size, uint64_t seed, csr_d *out)
{
    memset(out, 0, sizeof *out);
    const int64_t n64 = (int64_t)size * size * size;
    if (size < 1 || n64 > 0x7fffffff / 16)
        return -2;
    const int n = (int)n64;
    rng r = {seed * 0x9E3779B97F4A7C15ull + 0x1234567ull, 0, 0.0};
    out->rows = out->cols = n;
    out->rowstr = malloc(((size_t)n + 1) * sizeof(int));
    int *count = malloc((size_t)n * sizeof(int));
    if (!out->rowstr || !count) {
        free(count);
        csr_d_free(out);
        return -3;
    }
    out->rowstr[0] = 1;
    for (int i = 0; i < n; ++i) { /* big_gen.py:63-68 */
        int c = (int)rng_gauss(&r, 5.0, 4.0);
        if (c < 1)
            c = 1;
        if (c > n)
            c = n;
        count[i] = c;
        out->rowstr[i + 1] = out->rowstr[i] + c;
    }
    const int64_t nnz = out->rowstr[n] - 1;
    out->nnz = nnz;
    out->colidx = malloc((size_t)nnz * sizeof(int));
    out->a = malloc((size_t)nnz * sizeof(double));
    int *sample = malloc(((size_t)64 + 1) * sizeof(int));
    size_t cap = 64;
    if (!out->colidx || !out->a || !sample) {
        free(count), free(sample);
        csr_d_free(out);
        return -3;
    }
    int64_t w = 0; /* stream position; only the first nnz entries are what the reader keeps */
    for (int i = 0; i < n && w < nnz; ++i) { /* big_gen.py:71-82 */
        const int c = count[i];
        if ((size_t)c + 1 > cap) {
            cap = (size_t)c * 2 + 1;
            int *ns = realloc(sample, cap * sizeof(int));
            if (!ns) {
                free(count), free(sample);
                csr_d_free(out);
                return -3;
            }
            sample = ns;
        }
        int k = 0, has_diag = 0;
        while (k < c) { /* random.sample(range(1, n+1), c): distinct */
            int col = 1 + (int)(rng_uniform(&r) * n);
            if (col > n)
                col = n;
            int dup = 0;
            for (int t = 0; t < k; ++t)
                dup |= sample[t] == col;
            if (dup)
                continue;
            has_diag |= col == i + 1;
            sample[k++] = col;
        }
        if (!has_diag)
            sample[k++] = i + 1; /* appended after row_ptr was fixed: quirk N1 */
        qsort(sample, (size_t)k, sizeof(int), cmp_int);
        for (int t = 0; t < k && w < nnz; ++t, ++w) {
            out->colidx[w] = sample[t];
            out->a[w] = fabs(rng_gauss(&r, 0.0, 2.0));
        }
    }
    free(count), free(sample);
    return 0;
}

int sb_bicg(const csr_d *m, d_harness_fn spmv, int maxit, double rtol, double *hist, double *x,
            int64_t *calls, double *t_mult)
{
    const int n = m->rows;
    int rows = n;
    double *blk = calloc((size_t)9 * ((size_t)n + 2), sizeof(double));
    double *rhs = malloc(((size_t)n + 2) * sizeof(double));
    if (!blk || !rhs) {
        free(blk), free(rhs);
        return 0;
    }
    double *p = blk, *ap = p + n + 2, *pl = ap + n + 2, *apl = pl + n + 2, *r = apl + n + 2, *rl = r + n + 2,
           *tmp = rl + n + 2;
    for (int i = 0; i < n; ++i) { /* main.f:343-346 */
        x[i] = 0.0;
        rhs[i] = 1.0;
    }
    int64_t nc = 0;
    double tm = 0.0, t0;
#define MATPROD(y_, x_)                                            \
    do {                                                           \
        t0 = now_s();                                              \
        spmv((y_), m->a, (x_), m->rowstr, m->colidx, &rows);       \
        tm += now_s() - t0;                                        \
        ++nc;                                                      \
    } while (0)
    MATPROD(tmp, x); /* iter.f:46-47 */
    for (int i = 0; i < n; ++i) {
        r[i] = tmp[i] - rhs[i];
        rl[i] = r[i];
    }
    double rr = 0.0, rrp = 0.0, rn0 = 0.0;
    int its = maxit, converged = 0;
    for (int it = 1; it <= maxit; ++it) {
        double rn = 0.0; /* vecnorm, vec.f:37-41 */
        for (int i = 0; i < n; ++i)
            rn = rn + r[i] * r[i];
        rn = sqrt(rn);
        hist[it - 1] = rn;
        if (it == 1)
            rn0 = rn;
        if (rn < rtol * rn0) { /* iter.f:58-61 */
            its = it;
            converged = 1;
            break;
        }
        if (it > 1)
            rrp = rr;
        rr = 0.0; /* dotprod(r, rl) */
        for (int i = 0; i < n; ++i)
            rr = rr + r[i] * rl[i];
        if (it == 1) {
            memcpy(p, r, (size_t)n * sizeof(double));
            memcpy(pl, rl, (size_t)n * sizeof(double));
        } else {
            const double beta = rr / rrp;
            for (int i = 0; i < n; ++i)
                p[i] = beta * p[i] + r[i];
            for (int i = 0; i < n; ++i)
                pl[i] = beta * pl[i] + rl[i];
        }
        MATPROD(ap, p);   /* 'n' */
        MATPROD(apl, pl); /* 't' -- the same product in the reference, random.f:66-72 */
        double d = 0.0;
        for (int i = 0; i < n; ++i)
            d = d + pl[i] * ap[i];
        const double alpha = rr / d;
        for (int i = 0; i < n; ++i)
            x[i] = x[i] + (-alpha) * p[i];
        for (int i = 0; i < n; ++i)
            r[i] = r[i] + (-alpha) * ap[i];
        for (int i = 0; i < n; ++i)
            rl[i] = rl[i] + (-alpha) * apl[i];
    }
#undef MATPROD
    free(blk), free(rhs);
    if (calls)
        *calls = nc;
    if (t_mult)
        *t_mult = tm;
    return converged ? its : -its; /* iter.f:97 */
}

/* ================================= bfs ======================================================= */

int bfs_from_pairs(int rows, int cols, int64_t n, const int *x, const int *y, csr_d *out)
{
    memset(out, 0, sizeof *out);
    out->rows = rows;
    out->cols = cols;
    out->rowstr = malloc(((size_t)rows + 2) * sizeof(int));
    out->colidx = malloc((size_t)(n ? n : 1) * sizeof(int));
    out->a = malloc((size_t)(n ? n : 1) * sizeof(double));
    if (!out->rowstr || !out->colidx || !out->a) {
        csr_d_free(out);
        return -3;
    }
    /* COOMatrix::convert, library.cc:56-80 -- including its off-by-one: the entries of input row x
     * land behind row pointer x, and colidx = y + 1 */
    size_t np = 0;
    out->rowstr[np++] = 1;
    long current_row = 0;
    int64_t running_total = 0;
    for (int64_t i = 0; i < n; ++i) {
        for (; current_row < x[i] && np <= (size_t)rows; ++current_row)
            out->rowstr[np++] = (int)(running_total + 1);
        out->colidx[running_total] = y[i] + 1;
        out->a[running_total] = 1.0;
        running_total++;
    }
    for (; current_row < rows; ++current_row)
        out->rowstr[np++] = (int)(running_total + 1);
    while (np <= (size_t)rows)
        out->rowstr[np++] = (int)(running_total + 1);
    out->nnz = (int64_t)out->rowstr[rows] - 1; /* what the harness sees (native.c: rowstr[rows]-rowstr[0]) */
    return 0;
}

int bfs_read_pairs(const char *path, csr_d *out)
{
    FILE *f = fopen(path, "r");
    if (!f)
        return -1;
    long rows, cols, n;
    if (fscanf(f, "%ld %ld %ld", &rows, &cols, &n) != 3 || rows <= 0 || cols <= 0 || n < 0 || rows > 0x7ffffff0L ||
        cols > 0x7ffffff0L || n > 0x7ffffff0L) {
        fclose(f);
        return -2;
    }
    int *x = malloc((size_t)(n ? n : 1) * sizeof(int)), *y = malloc((size_t)(n ? n : 1) * sizeof(int));
    if (!x || !y) {
        fclose(f);
        free(x), free(y);
        return -3;
    }
    for (long i = 0; i < n; ++i)
        if (fscanf(f, "%d %d", &x[i], &y[i]) != 2 || x[i] < 0 || x[i] > rows || y[i] < 0 || y[i] > cols) {
            fclose(f); /* truncated, or an index outside the shape (library.cc:169-184 trusts the file) */
            free(x), free(y);
            return -2;
        }
    fclose(f);
    int rc = bfs_from_pairs((int)rows, (int)cols, n, x, y, out);
    free(x), free(y);
    return rc;
}

int bfs_run(const csr_d *m, d_harness_fn spmv, int source, double *distances)
{
    const int n = m->rows;
    int rows = n;
    /* the reference's vectors have N = rows entries while colidx reaches cols + 1:
     * pad with zeros so the product is defined */
    const size_t len = (size_t)(m->cols > n ? m->cols : n) + 2;
    double *front = calloc(len, sizeof(double)), *prod = calloc(len, sizeof(double));
    if (!front || !prod) {
        free(front), free(prod);
        return -1;
    }
    for (int i = 0; i < n; ++i)
        distances[i] = 0.0;
    front[source] = 1.0;
    distances[source] = 1.0;
    int levels = 0;
    for (int i = 1;; ++i) { /* bfs.cc:50-65 */
        spmv(prod, m->a, front, m->rowstr, m->colidx, &rows);
        ++levels;
        int cont = 0;
        for (int j = 0; j < n; ++j) {
            front[j] = (prod[j] != 0.0) && (distances[j] == 0.0);
            if (front[j] != 0.0) {
                cont = 1;
                distances[j] = i + 1;
            }
        }
        if (!cont)
            break;
    }
    free(front), free(prod);
    return levels;
}

/* ================================= pagerank ================================================== */

void pagerank_prepare(csr_d *m, double d)
{
    double *colsum = calloc((size_t)m->cols + 2, sizeof(double));
    if (!colsum)
        return;
    for (int64_t j = 0; j < m->nnz; ++j)
        colsum[m->colidx[j]] += m->a[j];
    for (int64_t j = 0; j < m->nnz; ++j) {
        const double s = colsum[m->colidx[j]];
        m->a[j] = (s != 0.0 ? m->a[j] / s : m->a[j]) * d; /* normalise(), then scale(d): main.cpp:105-111 */
    }
    free(colsum);
}

double pagerank_run(const csr_d *m, d_harness_fn spmv, double d, int iters, double *x)
{
    const int n = m->rows;
    int rows = n;
    double *y = calloc((size_t)n + 2, sizeof(double)), *last = malloc(((size_t)n + 2) * sizeof(double));
    double error = 0.0;
    if (!y || !last) {
        free(y), free(last);
        return -1.0;
    }
    for (int it = 0; it < iters; ++it) { /* main.cpp:128-149 */
        memcpy(last, x, (size_t)n * sizeof(double));
        double sum = 0.0;
        for (int i = 0; i < n; ++i)
            sum += x[i];
        const double add_term = (1.0 - d) * (sum / n);
        spmv(y, m->a, x, m->rowstr, m->colidx, &rows);
        for (int i = 0; i < n; ++i)
            y[i] += add_term;
        memcpy(x, y, (size_t)n * sizeof(double));
        error = 0.0;
        for (int i = 0; i < n; ++i)
            error += (x[i] - last[i]) * (x[i] - last[i]);
        error = sqrt(error);
    }
    free(y), free(last);
    return error;
}

/* ================================= Matrix Market / parboil =================================== */

typedef struct mm_info {
    int rows, cols;
    long nz;
    int symmetric, pattern;
} mm_info;

static int mm_header(FILE *f, mm_info *h)
{
    char line[1024];
    if (!fgets(line, sizeof line, f) || strncmp(line, "%%MatrixMarket", 14) != 0)
        return -2;
    for (char *c = line; *c; ++c)
        *c = (char)tolower((unsigned char)*c);
    if (!strstr(line, "coordinate") || strstr(line, "complex"))
        return -2; /* convert_dataset.c:66-73 */
    h->symmetric = strstr(line, "symmetric") != NULL;
    h->pattern = strstr(line, "pattern") != NULL;
    do {
        if (!fgets(line, sizeof line, f))
            return -2;
    } while (line[0] == '%');
    if (sscanf(line, "%d %d %ld", &h->rows, &h->cols, &h->nz) != 3)
        return -2;
    if (h->rows <= 0 || h->cols <= 0 || h->nz < 0 || h->nz > 0x3fffffffL)
        return -2; /* the reference converter trusts the file; a driver handed to maintainers must not */
    return 0;
}

typedef struct entry_f {
    int row, col;
    float val;
} entry_f;

int pb_coo_to_csr(const char *mtx_path, int mirrored, int binary, csr_f *out)
{
    memset(out, 0, sizeof *out);
    FILE *f = fopen(mtx_path, "r");
    if (!f)
        return -1;
    mm_info h;
    if (mm_header(f, &h)) {
        fclose(f);
        return -2;
    }
    entry_f *e = malloc((size_t)(mirrored ? 2 : 1) * (size_t)(h.nz ? h.nz : 1) * sizeof *e);
    if (!e) {
        fclose(f);
        return -3;
    }
    long cur = 0;
    for (long i = 0; i < h.nz; ++i, ++cur) { /* convert_dataset.c:93-112 */
        if (!binary) {
            if (fscanf(f, "%d %d %f", &e[cur].row, &e[cur].col, &e[cur].val) != 3)
                break;
        } else {
            if (fscanf(f, "%d %d", &e[cur].row, &e[cur].col) != 2)
                break;
            e[cur].val = 1.0f;
        }
        e[cur].row--;
        e[cur].col--;
        /* an index outside the announced shape would be written outside rowstr[] below (mirrored: both ways) */
        if (e[cur].row < 0 || e[cur].row >= h.rows || e[cur].col < 0 || e[cur].col >= h.cols ||
            (mirrored && (e[cur].col >= h.rows || e[cur].row >= h.cols))) {
            fclose(f);
            free(e);
            return -2;
        }
        if (mirrored && e[cur].row != e[cur].col) {
            ++cur;
            e[cur].val = e[cur - 1].val;
            e[cur].col = e[cur - 1].row;
            e[cur].row = e[cur - 1].col;
        }
    }
    fclose(f);
    const long nz = cur;
    if (nz == 0) {
        free(e);
        return -2;
    }
    /* qsort by row only (convert_dataset.c:120); glibc's qsort is a merge sort, i.e. stable, and the
     * golden outputs depend on that order -- a counting sort by row reproduces it */
    int maxrow = 0;
    for (long i = 0; i < nz; ++i)
        if (e[i].row > maxrow)
            maxrow = e[i].row;
    const int rows = maxrow + 1; /* :121 */
    out->rows = rows;
    out->cols = h.cols > rows ? h.cols : rows;
    out->nnz = nz;
    out->rowstr = calloc((size_t)rows + 2, sizeof(int));
    out->colidx = malloc((size_t)nz * sizeof(int));
    out->a = malloc((size_t)nz * sizeof(float));
    if (!out->rowstr || !out->colidx || !out->a) {
        free(e);
        csr_f_free(out);
        return -3;
    }
    for (long i = 0; i < nz; ++i)
        out->rowstr[e[i].row + 1]++;
    out->rowstr[0] = 1;
    for (int r = 0; r < rows; ++r)
        out->rowstr[r + 1] += out->rowstr[r];
    int *pos = malloc((size_t)rows * sizeof(int));
    if (!pos) {
        free(e);
        csr_f_free(out);
        return -3;
    }
    for (int r = 0; r < rows; ++r)
        pos[r] = out->rowstr[r] - 1;
    for (long i = 0; i < nz; ++i) {
        const int p = pos[e[i].row]++;
        out->colidx[p] = e[i].col + 1; /* :150 */
        out->a[p] = e[i].val;
    }
    free(pos), free(e);
    return 0;
}

int pb_read_vec(const char *path, float *v, int n)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return -1;
    size_t got = fread(v, sizeof(float), (size_t)n, f);
    fclose(f);
    return got == (size_t)n ? 0 : -2;
}

int pb_write_out(const char *path, const float *v, int n)
{
    FILE *f = fopen(path, "wb");
    if (!f)
        return -1;
    uint32_t n32 = (uint32_t)n;
    fwrite(&n32, sizeof n32, 1, f);
    fwrite(v, sizeof(float), (size_t)n, f);
    fclose(f);
    return 0;
}

int pb_read_out(const char *path, float **v, int *n)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return -1;
    uint32_t n32 = 0;
    if (fread(&n32, sizeof n32, 1, f) != 1) {
        fclose(f);
        return -2;
    }
    *v = malloc((size_t)(n32 ? n32 : 1) * sizeof(float));
    if (!*v) {
        fclose(f);
        return -3;
    }
    size_t got = fread(*v, sizeof(float), n32, f);
    fclose(f);
    *n = (int)n32;
    return got == n32 ? 0 : -2;
}

int pb_compare(const float *ref, const float *cmp, int n)
{
    double mx = 0.0;
    for (int i = 0; i < n; ++i)
        if (fabs((double)ref[i]) > mx)
            mx = fabs((double)ref[i]);
    const double abstol = 1e-4 * mx; /* compare-output:18 */
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const double diff = fabs((double)ref[i] - (double)cmp[i]);
        if (!(diff <= abstol || diff < 0.002 * fabs((double)ref[i])))
            ++bad;
    }
    return bad;
}

void pb_run(const csr_f *m, f_harness_fn spmv, float *x, float *y, int reps)
{
    int rows = m->rows;
    for (int p = 0; p < reps; ++p)
        spmv(y, m->a, x, m->rowstr, m->colidx, &rows);
}

int mm_read_csr(const char *path, csr_d *out)
{
    memset(out, 0, sizeof *out);
    FILE *f = fopen(path, "r");
    if (!f)
        return -1;
    mm_info h;
    if (mm_header(f, &h)) {
        fclose(f);
        return -2;
    }
    const size_t cap = (size_t)(h.symmetric ? 2 : 1) * (size_t)(h.nz ? h.nz : 1);
    int *ri = malloc(cap * sizeof(int)), *ci = malloc(cap * sizeof(int));
    double *va = malloc(cap * sizeof(double));
    if (!ri || !ci || !va) {
        fclose(f);
        free(ri), free(ci), free(va);
        return -3;
    }
    long cur = 0;
    for (long i = 0; i < h.nz; ++i) {
        int r, c;
        double v = 1.0;
        if (h.pattern ? fscanf(f, "%d %d", &r, &c) != 2 : fscanf(f, "%d %d %lf", &r, &c, &v) != 3)
            break;
        if (r < 1 || r > h.rows || c < 1 || c > h.cols || (h.symmetric && (c > h.rows || r > h.cols))) {
            fclose(f);
            free(ri), free(ci), free(va);
            return -2; /* malformed file: indices outside the announced shape */
        }
        ri[cur] = r - 1, ci[cur] = c - 1, va[cur] = v, ++cur;
        if (h.symmetric && r != c)
            ri[cur] = c - 1, ci[cur] = r - 1, va[cur] = v, ++cur;
    }
    fclose(f);
    out->rows = h.rows;
    out->cols = h.cols;
    out->nnz = cur;
    out->rowstr = calloc((size_t)h.rows + 2, sizeof(int));
    out->colidx = malloc((size_t)(cur ? cur : 1) * sizeof(int));
    out->a = malloc((size_t)(cur ? cur : 1) * sizeof(double));
    int *pos = malloc(((size_t)h.rows + 1) * sizeof(int));
    if (!out->rowstr || !out->colidx || !out->a || !pos) {
        free(ri), free(ci), free(va), free(pos);
        csr_d_free(out);
        return -3;
    }
    for (long i = 0; i < cur; ++i)
        out->rowstr[ri[i] + 1]++;
    out->rowstr[0] = 1;
    for (int r = 0; r < h.rows; ++r)
        out->rowstr[r + 1] += out->rowstr[r];
    for (int r = 0; r < h.rows; ++r)
        pos[r] = out->rowstr[r] - 1;
    for (long i = 0; i < cur; ++i) {
        const int p = pos[ri[i]]++;
        out->colidx[p] = ci[i] + 1;
        out->a[p] = va[i];
    }
    free(ri), free(ci), free(va), free(pos);
    return 0;
}

/* ================================= PATHSAMPLE ================================================ */

void ps_free(ps_net *n)
{
    free(n->ncol), free(n->nval), free(n->dmat);
    memset(n, 0, sizeof *n);
}

int ps_generate(int nmin, int maxcol, uint64_t seed, ps_net *out)
{
    memset(out, 0, sizeof *out);
    if (nmin < 2 || maxcol < 1)
        return -2;
    if (maxcol > nmin - 1)
        maxcol = nmin - 1;
    out->nmin = nmin, out->maxcol = maxcol;
    out->ncol = malloc((size_t)nmin * sizeof(int));
    out->nval = malloc((size_t)nmin * maxcol * sizeof(int));
    out->dmat = malloc((size_t)nmin * maxcol * sizeof(double));
    if (!out->ncol || !out->nval || !out->dmat) {
        ps_free(out);
        return -3;
    }
    rng r = {seed * 0x9E3779B97F4A7C15ull + 38, 0, 0.0};
    for (int j1 = 0; j1 < nmin; ++j1) {
        int nc = 1 + (int)(rng_u64(&r) % (uint64_t)maxcol);
        int *v = out->nval + (size_t)j1 * maxcol;
        double *d = out->dmat + (size_t)j1 * maxcol, sum = 0.0;
        for (int j2 = 0; j2 < nc; ++j2) { /* distinct neighbours, not the minimum itself */
            int c;
            int again;
            do {
                c = 1 + (int)(rng_u64(&r) % (uint64_t)nmin);
                again = c == j1 + 1;
                for (int k = 0; k < j2 && !again; ++k)
                    again = v[k] == c;
            } while (again);
            v[j2] = c;
            d[j2] = 0.05 + rng_uniform(&r);
            sum += d[j2];
        }
        for (int j2 = 0; j2 < nc; ++j2)
            d[j2] *= 0.98 / sum; /* branching probabilities: the rest is the self-return */
        out->ncol[j1] = nc;
    }
    return 0;
}

int64_t ps_committor(const ps_net *n, d_harness_fn spmv, int npfold, int variant, double *gpfold, double *t_calls)
{
    const int nmin = n->nmin;
    /* Pfold.f90:193-211 / NGT.f:976-996: compressed row storage from the neighbour lists */
    int64_t nonzero = 0;
    for (int j1 = 0; j1 < nmin; ++j1)
        nonzero += n->ncol[j1];
    const size_t len = (size_t)(nonzero > nmin ? nonzero : nmin) + 2; /* swapped roles: see callers.h */
    double *dvec = calloc(len, sizeof(double)), *newpfold = calloc(len, sizeof(double));
    double *gp = calloc((size_t)nmin + 2, sizeof(double));
    int *col_ind = malloc((size_t)(nonzero ? nonzero : 1) * sizeof(int)), *row_ptr = malloc(((size_t)nmin + 1) * sizeof(int));
    if (!dvec || !newpfold || !gp || !col_ind || !row_ptr) {
        free(dvec), free(newpfold), free(gp), free(col_ind), free(row_ptr);
        return -3;
    }
    int64_t ncount = 0;
    int ncolprev = 0;
    row_ptr[0] = 1;
    for (int j1 = 0; j1 < nmin; ++j1) {
        if (j1 > 0)
            row_ptr[j1] = row_ptr[j1 - 1] + ncolprev;
        ncolprev = n->ncol[j1];
        for (int j2 = 0; j2 < n->ncol[j1]; ++j2) {
            dvec[ncount] = n->dmat[(size_t)j1 * n->maxcol + j2];
            col_ind[ncount] = n->nval[(size_t)j1 * n->maxcol + j2];
            ++ncount;
        }
    }
    row_ptr[nmin] = (int)(nonzero + 1);
    /* initial committor values: 1 for the "A" minima (here: the first tenth), 0 elsewhere */
    for (int j = 0; j < nmin; ++j)
        gp[j] = j < (nmin + 9) / 10 ? 1.0 : 0.0;
    memcpy(newpfold, gp, (size_t)nmin * sizeof(double)); /* NEWPFOLD(1:NMIN)=GPFOLD(1:NMIN) */
    double t = 0.0;
    for (int j1 = 0; j1 < npfold; ++j1) {
        int rows = nmin - 1; /* the temporary NMIN-1 */
        double t0 = now_s();
        spmv(gp, newpfold, dvec, row_ptr, col_ind, &rows); /* Pfold.f90:221, NGT.f:1001 */
        t += now_s() - t0;
        if (variant == 0)
            memcpy(newpfold, gp, (size_t)nmin * sizeof(double)); /* Pfold.f90:222 */
    }
    memcpy(gpfold, gp, (size_t)nmin * sizeof(double));
    if (t_calls)
        *t_calls = t;
    free(dvec), free(newpfold), free(gp), free(col_ind), free(row_ptr);
    return nonzero;
}

/* ================================= seeded graphs ============================================== */

static int cmp_u64(const void *a, const void *b)
{
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

static int csr_from_sorted_edges(int n, const uint64_t *key, int64_t m, csr_d *out)
{
    memset(out, 0, sizeof *out);
    out->rows = out->cols = n;
    out->rowstr = calloc((size_t)n + 2, sizeof(int));
    out->colidx = malloc((size_t)(m ? m : 1) * sizeof(int));
    out->a = malloc((size_t)(m ? m : 1) * sizeof(double));
    if (!out->rowstr || !out->colidx || !out->a) {
        csr_d_free(out);
        return -3;
    }
    int64_t w = 0;
    for (int64_t i = 0; i < m; ++i) {
        if (i && key[i] == key[i - 1])
            continue; /* duplicate edge */
        const int s = (int)(key[i] >> 32), d = (int)(key[i] & 0xffffffffu);
        out->rowstr[s + 1]++;
        out->colidx[w] = d + 1;
        out->a[w] = 1.0;
        ++w;
    }
    out->nnz = w;
    out->rowstr[0] = 1;
    for (int r = 0; r < n; ++r)
        out->rowstr[r + 1] += out->rowstr[r];
    return 0;
}

int graph_rmat(int scale, int edge_factor, uint64_t seed, csr_d *out)
{
    if (scale < 1 || scale > 26)
        return -2;
    const int n = 1 << scale;
    const int64_t m = (int64_t)n * edge_factor;
    uint64_t *key = malloc((size_t)m * sizeof *key);
    if (!key)
        return -3;
    rng r = {seed * 0x9E3779B97F4A7C15ull + 99, 0, 0.0};
    for (int64_t e = 0; e < m; ++e) {
        uint32_t s = 0, d = 0;
        for (int b = 0; b < scale; ++b) { /* (a,b,c,d) = (.57,.19,.19,.05) */
            const double u = rng_uniform(&r);
            s |= (uint32_t)(u >= 0.76) << b;
            d |= (uint32_t)((u >= 0.57 && u < 0.76) || u >= 0.95) << b;
        }
        key[e] = ((uint64_t)s << 32) | d;
    }
    qsort(key, (size_t)m, sizeof *key, cmp_u64);
    int rc = csr_from_sorted_edges(n, key, m, out);
    free(key);
    return rc;
}

int graph_grid2d(int nx, int ny, csr_d *out)
{
    const int64_t n64 = (int64_t)nx * ny;
    if (nx < 1 || ny < 1 || n64 > 0x7fffffff / 8)
        return -2;
    const int n = (int)n64;
    uint64_t *key = malloc((size_t)n * 4 * sizeof *key);
    if (!key)
        return -3;
    int64_t m = 0;
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
            const uint64_t v = (uint64_t)y * nx + x;
            if (y > 0)
                key[m++] = (v << 32) | (v - nx);
            if (x > 0)
                key[m++] = (v << 32) | (v - 1);
            if (x + 1 < nx)
                key[m++] = (v << 32) | (v + 1);
            if (y + 1 < ny)
                key[m++] = (v << 32) | (v + nx);
        }
    int rc = csr_from_sorted_edges(n, key, m, out); /* generated in (row, column) order already */
    free(key);
    return rc;
}
