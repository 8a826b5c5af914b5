/*
 * tests/native/harness_driver.c -- scenarios for spmv_harness.c on the mock device (mock_device.c):
 * cache hits, write detection through page protection and through the compared edge pages, cache policies,
 * the zero-copy path, churn, float/double side by side, degenerate shapes.  Prints "ok <scenario>" lines; any
 * failed expectation aborts.  Run by tests/test_native_sanitizers.py, plain and under ASan/UBSan.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spmv_harness.h"
#include "spmv_hip.h"

extern int mock_creates, mock_relayout_creates, mock_word_copies, mock_registered, mock_value_refreshes;

#define EXPECT(c)                                                                 \
    do {                                                                          \
        if (!(c)) {                                                               \
            fprintf(stderr, "%s:%d: expectation failed: %s\n", __FILE__, __LINE__, #c); \
            abort();                                                              \
        }                                                                         \
    } while (0)

static uint64_t st = 0x9E3779B97F4A7C15ull;
static double rnd(void)
{
    st ^= st << 13, st ^= st >> 7, st ^= st << 17;
    return (double)(st % 2001) / 1000.0 - 1.0;
}

typedef struct csr {
    int rows, cols;
    int64_t nnz;
    int *rowstr, *colidx;
    double *a;
} csr;

static csr make(int rows, int cols, int len)
{
    csr m;
    m.rows = rows, m.cols = cols, m.nnz = (int64_t)rows * len;
    m.rowstr = (int *)malloc(((size_t)rows + 1) * sizeof(int));
    m.colidx = (int *)malloc((size_t)m.nnz * sizeof(int));
    m.a = (double *)malloc((size_t)m.nnz * sizeof(double));
    for (int r = 0; r <= rows; ++r)
        m.rowstr[r] = 1 + r * len;
    for (int64_t j = 0; j < m.nnz; ++j) {
        m.colidx[j] = 1 + (int)((uint64_t)(rnd() * 1e6 + 2e6) % (uint64_t)cols);
        m.a[j] = rnd();
    }
    m.colidx[m.nnz - 1] = cols; /* the inferred column count */
    return m;
}

static void check(const csr *m, const double *x, const double *y)
{
    for (int r = 0; r < m->rows; ++r) {
        double acc = 0.0;
        for (int j = m->rowstr[r] - 1; j < m->rowstr[r + 1] - 1; ++j)
            acc += m->a[j] * x[m->colidx[j] - 1];
        EXPECT(acc == y[r]);
    }
}

/* SPMV_HIP_DISPATCH=auto: tiny products on the host loop, bit-identical to native-impl.c; a mid-size matrix on
 * the host until the process has spent its budget there, then on the device; a large one on the device at once */
static int dispatch_scenario(void)
{
    spmv_hip_stats s;
    {
        csr m = make(300, 300, 3);
        double *x = (double *)malloc(300 * sizeof(double)), *y = (double *)malloc(300 * sizeof(double));
        for (int i = 0; i < 300; ++i)
            x[i] = rnd();
        int rows = m.rows;
        for (int k = 0; k < 5; ++k)
            spmv_harness_(y, m.a, x, m.rowstr, m.colidx, &rows);
        check(&m, x, y);
        spmv_hip_get_stats(&s);
        EXPECT(s.calls == 5 && s.host_calls == 5 && s.uploads == 0 && mock_creates == 0);
        free(m.rowstr), free(m.colidx), free(m.a), free(x), free(y);
    }
    {
        csr m = make(20000, 20000, 25); /* 500 000 nonzeros: ~0.5 ms on the host, more than a PCIe round trip */
        double *x = (double *)malloc(20000 * sizeof(double)), *y = (double *)malloc(20000 * sizeof(double));
        for (int i = 0; i < 20000; ++i)
            x[i] = rnd();
        int rows = m.rows, n = 0;
        do {
            spmv_harness_(y, m.a, x, m.rowstr, m.colidx, &rows);
            check(&m, x, y);
            spmv_hip_get_stats(&s);
            ++n;
        } while (s.uploads == 0 && n < 100000);
        EXPECT(s.uploads == 1 && mock_creates == 1); /* the budget ran out: the context was "created", the matrix uploaded */
        /* the budget is charged in ESTIMATED seconds (0.35 ns per nonzero and row of a matrix that fits the host's caches,
         * + 0.2 us), so the call that moves to the device is the same in every run on every host: 5 tiny products
         * (0.62 us each) + n x 182.2 us < 100 ms */
        EXPECT(s.host_calls == 5 + 548 && n == 549);
        const long long host_before = (long long)s.host_calls;
        spmv_harness_(y, m.a, x, m.rowstr, m.colidx, &rows);
        spmv_hip_get_stats(&s);
        EXPECT((long long)s.host_calls == host_before && s.uploads == 1); /* stays on the device, cached */
        /* a second matrix of the same size, now that the context exists: on the host until it has spent what its upload
         * costs (10 ms + 6 MB at 20 GB/s = 10.3 ms of estimated host time: 56 products of 182.2 us), then on the device */
        {
            csr m2 = make(20000, 20000, 25);
            int r2 = m2.rows, k = 0;
            const long long h0 = (long long)s.host_calls;
            do {
                spmv_harness_(y, m2.a, x, m2.rowstr, m2.colidx, &r2);
                check(&m2, x, y);
                spmv_hip_get_stats(&s);
                ++k;
            } while (s.uploads == 1 && k < 1000);
            EXPECT(k == 57 && (long long)s.host_calls - h0 == 56 && s.uploads == 2);
            spmv_hip_release(); /* both copies, before m2's arrays go */
            free(m2.rowstr), free(m2.colidx), free(m2.a);
            spmv_hip_get_stats(&s);
        }
        const long long host_before2 = (long long)s.host_calls;
        /* a tiny product still goes to the host: the PCIe round trip has not become cheaper */
        int few = 40;
        spmv_harness_(y, m.a, x, m.rowstr, m.colidx, &few);
        spmv_hip_get_stats(&s);
        EXPECT((long long)s.host_calls == host_before2 + 1);
        spmv_hip_release();
        free(m.rowstr), free(m.colidx), free(m.a), free(x), free(y);
    }
    printf("ok host / device dispatch\n");
    return 0;
}

int main(int argc, char **argv)
{
    const char *policy = argc > 1 ? argv[1] : "mprotect";
    spmv_hip_stats s;
    if (argc > 2 && !strcmp(argv[2], "dispatch"))
        return dispatch_scenario();
    csr m = make(6000, 5000, 40); /* a: 1.9 MB, colidx: 960 KB: interior pages get protected */
    double *x = (double *)malloc(5000 * sizeof(double)), *y = (double *)malloc(6000 * sizeof(double));
    for (int i = 0; i < 5000; ++i)
        x[i] = rnd();
    int rows = m.rows;

    spmv_harness_(y, m.a, x, m.rowstr, m.colidx, &rows);
    check(&m, x, y);
    spmv_harness_(y, m.a, x, m.rowstr, m.colidx, &rows);
    check(&m, x, y);
    spmv_hip_get_stats(&s);
    EXPECT(s.calls == 2);
    EXPECT(s.uploads == (strcmp(policy, "off") ? 1 : 2));
    printf("ok cache hit (%s)\n", policy);

    /* edit in the middle of a (an interior, protected page under mprotect) */
    m.a[m.nnz / 2] += 1.0;
    spmv_harness_(y, m.a, x, m.rowstr, m.colidx, &rows);
    spmv_hip_get_stats(&s);
    if (!strcmp(policy, "pointer")) {
        EXPECT(s.uploads == 1); /* gpu-unsafe.c semantics: the edit goes unnoticed */
    } else if (!strcmp(policy, "off")) {
        check(&m, x, y);
        EXPECT(s.uploads == 3);
    } else { /* only the values changed: they cross again, the copy and its plan stay */
        check(&m, x, y);
        EXPECT(s.uploads == 1 && s.value_refreshes == 1 && s.invalidations == 0);
    }
    printf("ok interior edit (%s)\n", policy);

    if (strcmp(policy, "pointer")) {
        /* edit the first element of colidx's neighbour page fragment and the last of rowstr's */
        m.a[0] -= 0.5;
        spmv_harness_(y, m.a, x, m.rowstr, m.colidx, &rows);
        check(&m, x, y);
        m.colidx[1] = m.colidx[1] % 4999 + 1;
        spmv_harness_(y, m.a, x, m.rowstr, m.colidx, &rows);
        check(&m, x, y);
        spmv_hip_get_stats(&s);
        /* a[0] sits on the partial first page (compared, not protected): a second value refresh; the colidx edit is
         * what drops the copy */
        EXPECT(strcmp(policy, "off") == 0 || (s.value_refreshes == 2 && s.invalidations == 1 && s.uploads == 2));
        {
            const int parts = getenv("SPMV_HIP_PARTS") ? atoi(getenv("SPMV_HIP_PARTS")) : 1;
            EXPECT(strcmp(policy, "off") == 0 || mock_relayout_creates == 2 * parts);
        }
        /* a matrix whose PATTERN keeps changing (two copies in a row dropped after fewer than 3 products): its uploads
         * stop asking for the re-laid-out plan */
        if (strcmp(policy, "off")) {
            const int before = mock_relayout_creates;
            for (int k = 0; k < 3; ++k) {
                m.colidx[2 + k] = m.colidx[2 + k] % 4999 + 1;
                spmv_harness_(y, m.a, x, m.rowstr, m.colidx, &rows);
                check(&m, x, y);
            }
            const int parts = getenv("SPMV_HIP_PARTS") ? atoi(getenv("SPMV_HIP_PARTS")) : 1;
            EXPECT(mock_relayout_creates - before == parts); /* two copies died young (the one before the loop and the
                                                                 first of these): the second and third upload here churn */
        }
        printf("ok edge edits and churn (%s)\n", policy);
    }

    /* fewer rows through a temporary, as Pfold.f90:221 passes NMIN-1 */
    int fewer = m.rows - 1;
    spmv_harness_(y, m.a, x, m.rowstr, m.colidx, &fewer);
    for (int r = 0; r < fewer; ++r) {
        double acc = 0.0;
        for (int j = m.rowstr[r] - 1; j < m.rowstr[r + 1] - 1; ++j)
            acc += m.a[j] * x[m.colidx[j] - 1];
        EXPECT(acc == y[r]);
    }
    printf("ok rows as a temporary (%s)\n", policy);

    /* small vectors went through the zero-copy path (copy kernel), unless switched off */
    {
        const char *pin = getenv("SPMV_HIP_PIN");
        const int off = getenv("SPMV_HIP_ZEROCOPY") || getenv("MOCK_NO_DEVPTR") || (pin && !strcmp(pin, "pageable"));
        EXPECT(off ? mock_word_copies == 0 || getenv("MOCK_NO_DEVPTR") != NULL : mock_word_copies > 0);
    }

    /* float twin next to the double matrix */
    {
        const int n = 300;
        int *rs = (int *)malloc((n + 1) * sizeof(int)), *ci = (int *)malloc(n * 3 * sizeof(int));
        float *fa = (float *)malloc(n * 3 * sizeof(float)), *fx = (float *)malloc(n * sizeof(float));
        float *fy = (float *)malloc(n * sizeof(float));
        for (int r = 0; r <= n; ++r)
            rs[r] = 1 + 3 * r;
        for (int j = 0; j < 3 * n; ++j)
            ci[j] = 1 + (j * 7) % n, fa[j] = (float)rnd();
        ci[3 * n - 1] = n;
        for (int i = 0; i < n; ++i)
            fx[i] = (float)rnd();
        int nn = n;
        f_spmv_harness_(fy, fa, fx, rs, ci, &nn);
        for (int r = 0; r < n; ++r) {
            float acc = 0.0f;
            for (int j = rs[r] - 1; j < rs[r + 1] - 1; ++j)
                acc += fa[j] * fx[ci[j] - 1];
            EXPECT(acc == fy[r]);
        }
        spmv_hip_release(); /* before freeing arrays the cache may have protected */
        free(rs), free(ci), free(fa), free(fx), free(fy);
        printf("ok float twin (%s)\n", policy);
    }

    /* Cached matrices that share protected host pages: a double and a float matrix on ONE rowstr / colidx
     * (>= 64 whole pages each, so they are page-protected under the default policy).  A write into the
     * shared colidx must refresh BOTH device copies; evicting one slot must not leave the other one
     * unguarded; the same arrays with another `rows` are a third slot over the same pages. */
    if (strcmp(policy, "pointer")) {
        const int n = 100000, len = 4;
        int *rs = (int *)malloc(((size_t)n + 1) * sizeof(int)), *ci = (int *)malloc((size_t)n * len * sizeof(int));
        double *da = (double *)malloc((size_t)n * len * sizeof(double));
        float *fa = (float *)malloc((size_t)n * len * sizeof(float));
        double *dx = (double *)malloc((size_t)n * sizeof(double)), *dy = (double *)malloc((size_t)n * sizeof(double));
        float *fx = (float *)malloc((size_t)n * sizeof(float)), *fy = (float *)malloc((size_t)n * sizeof(float));
        for (int r = 0; r <= n; ++r)
            rs[r] = 1 + len * r;
        for (int j = 0; j < n * len; ++j) {
            ci[j] = 1 + (int)(((int64_t)j * 7919) % n);
            da[j] = rnd();
            fa[j] = (float)rnd();
        }
        ci[n * len - 1] = n;
        for (int i = 0; i < n; ++i)
            dx[i] = rnd(), fx[i] = (float)rnd();
        int nn = n;
        csr dm = {n, n, (int64_t)n * len, rs, ci, da};
#define CHECK_F()                                                  \
    for (int r = 0; r < nn; ++r) {                                 \
        float acc = 0.0f;                                          \
        for (int j = rs[r] - 1; j < rs[r + 1] - 1; ++j)            \
            acc += fa[j] * fx[ci[j] - 1];                          \
        EXPECT(acc == fy[r]);                                      \
    }
        spmv_harness_(dy, da, dx, rs, ci, &nn);
        f_spmv_harness_(fy, fa, fx, rs, ci, &nn);
        check(&dm, dx, dy);
        CHECK_F();
        /* one edit in the middle of the shared colidx: both copies are stale */
        ci[n * len / 2] = ci[n * len / 2] % (n - 1) + 1;
        spmv_harness_(dy, da, dx, rs, ci, &nn);
        check(&dm, dx, dy);
        f_spmv_harness_(fy, fa, fx, rs, ci, &nn);
        CHECK_F();
        /* an edit of the double values only: the float copy's pages (rowstr/colidx) went writable with the
         * double slot's, so its next call must not trust them -- edit colidx right after, without a fault */
        da[n] += 1.0;
        ci[n * len / 3] = ci[n * len / 3] % (n - 1) + 1;
        f_spmv_harness_(fy, fa, fx, rs, ci, &nn);
        CHECK_F();
        spmv_harness_(dy, da, dx, rs, ci, &nn);
        check(&dm, dx, dy);
        /* the same arrays with fewer rows: a third slot over the same pages */
        int fewer2 = n - 7;
        spmv_harness_(dy, da, dx, rs, ci, &fewer2);
        ci[5 * len] = ci[5 * len] % (n - 1) + 1;
        spmv_harness_(dy, da, dx, rs, ci, &fewer2);
        spmv_harness_(dy, da, dx, rs, ci, &nn);
        check(&dm, dx, dy);
        f_spmv_harness_(fy, fa, fx, rs, ci, &nn);
        CHECK_F();
#undef CHECK_F
        spmv_hip_release();
        free(rs), free(ci), free(da), free(fa), free(dx), free(dy), free(fx), free(fy);
        printf("ok slots sharing protected pages (%s)\n", policy);
    }

    /* PATHSAMPLE's PFOLD (Pfold.f90:219-223): new values before every call, the pattern never changes.  The first change
     * is answered by uploading the values alone; because the copy had served one product, the values are then taken
     * to be volatile and uploaded on every call without a watch: one upload, n - 1 value refreshes, no invalidation.
     * A second matrix on the SAME rowstr / colidx whose values nobody touches is not disturbed by any of it. */
    if (strcmp(policy, "pointer") && strcmp(policy, "off")) {
        const int n = 60000, len = 5, calls = 12;
        int *rs = (int *)malloc(((size_t)n + 1) * sizeof(int)), *ci = (int *)malloc((size_t)n * len * sizeof(int));
        double *av = (double *)malloc((size_t)n * len * sizeof(double)), *as = (double *)malloc((size_t)n * len * sizeof(double));
        double *xx = (double *)malloc((size_t)n * sizeof(double)), *yy = (double *)malloc((size_t)n * sizeof(double));
        for (int r = 0; r <= n; ++r)
            rs[r] = 1 + len * r;
        for (int j = 0; j < n * len; ++j)
            ci[j] = 1 + (int)(((int64_t)j * 104729) % n), av[j] = rnd(), as[j] = rnd();
        ci[n * len - 1] = n;
        for (int i = 0; i < n; ++i)
            xx[i] = rnd();
        int nn = n;
        csr vm = {n, n, (int64_t)n * len, rs, ci, av}, sm = {n, n, (int64_t)n * len, rs, ci, as};
        spmv_hip_reset_stats();
        const int creates0 = mock_creates, refresh0 = mock_value_refreshes;
        spmv_harness_(yy, as, xx, rs, ci, &nn); /* the stable matrix */
        check(&sm, xx, yy);
        for (int k = 0; k < calls; ++k) {
            for (int j = 0; j < n; ++j) /* the first NMIN values change (callers.c ps_committor) */
                av[j] = rnd();
            spmv_harness_(yy, av, xx, rs, ci, &nn);
            check(&vm, xx, yy);
            if (k % 4 == 3) {
                spmv_harness_(yy, as, xx, rs, ci, &nn);
                check(&sm, xx, yy);
            }
        }
        spmv_hip_get_stats(&s);
        const int parts = getenv("SPMV_HIP_PARTS") ? atoi(getenv("SPMV_HIP_PARTS")) : 1;
        EXPECT(s.uploads == 2 && mock_creates - creates0 == 2 * parts); /* one per matrix, ever */
        EXPECT(s.value_refreshes == calls - 1 && mock_value_refreshes - refresh0 == (calls - 1) * parts);
        EXPECT(s.invalidations == 0);
        /* a colidx edit is not a value refresh: both copies that share the array upload again */
        ci[7] = ci[7] % (n - 1) + 1;
        spmv_harness_(yy, av, xx, rs, ci, &nn);
        check(&vm, xx, yy);
        spmv_harness_(yy, as, xx, rs, ci, &nn);
        check(&sm, xx, yy);
        spmv_hip_get_stats(&s);
        EXPECT(s.uploads == 4);
        /* values that stop changing: with a watch again after the look that finds them unchanged */
        for (int k = 0; k < 200; ++k) {
            spmv_harness_(yy, av, xx, rs, ci, &nn);
            if (k % 50 == 0)
                check(&vm, xx, yy);
        }
        spmv_hip_get_stats(&s);
        const long long refreshes_quiet = (long long)s.value_refreshes;
        for (int k = 0; k < 10; ++k)
            spmv_harness_(yy, av, xx, rs, ci, &nn);
        spmv_hip_get_stats(&s);
        EXPECT((long long)s.value_refreshes == refreshes_quiet); /* the volatility ended at the first look (64 calls) */
        av[n * len / 2] += 1.0; /* ... and a later edit is seen again */
        spmv_harness_(yy, av, xx, rs, ci, &nn);
        check(&vm, xx, yy);
        spmv_hip_release();
        free(rs), free(ci), free(av), free(as), free(xx), free(yy);
        printf("ok value refreshes (%s)\n", policy);
    }

    /* an edit of the saved head fragment ALONE (a[0] lies on the partial first page, which is compared,
     * not protected) on a copy younger than 3 products: the values go volatile and the guard record of `a` is dropped --
     * its interior pages must go writable with it, or the caller's next interior write dies on a fault nobody owns. */
    if (strcmp(policy, "pointer") && strcmp(policy, "off")) {
        csr fm = make(6000, 5000, 40);
        int fr = fm.rows;
        spmv_hip_reset_stats();
        spmv_harness_(y, fm.a, x, fm.rowstr, fm.colidx, &fr);
        check(&fm, x, y);
        fm.a[0] += 1.0;
        spmv_harness_(y, fm.a, x, fm.rowstr, fm.colidx, &fr);
        check(&fm, x, y);
        fm.a[fm.nnz / 2] += 1.0; /* an interior page: read-only until the fix */
        fm.a[fm.nnz - 1] += 1.0; /* the tail fragment */
        spmv_harness_(y, fm.a, x, fm.rowstr, fm.colidx, &fr);
        check(&fm, x, y);
        spmv_hip_get_stats(&s);
        EXPECT(s.uploads == 1 && s.value_refreshes == 2 && s.invalidations == 0);
        /* the same on a copy that has lived (>= 3 products): the watch is re-armed, and a later interior write faults
         * into OUR handler and is seen */
        spmv_hip_release();
        spmv_hip_reset_stats();
        for (int k = 0; k < 4; ++k)
            spmv_harness_(y, fm.a, x, fm.rowstr, fm.colidx, &fr);
        fm.a[0] -= 0.25;
        spmv_harness_(y, fm.a, x, fm.rowstr, fm.colidx, &fr);
        check(&fm, x, y);
        for (int k = 0; k < 3; ++k)
            spmv_harness_(y, fm.a, x, fm.rowstr, fm.colidx, &fr);
        fm.a[fm.nnz / 3] += 2.0;
        spmv_harness_(y, fm.a, x, fm.rowstr, fm.colidx, &fr);
        check(&fm, x, y);
        spmv_hip_get_stats(&s);
        EXPECT(s.uploads == 1 && s.value_refreshes == 2 && s.invalidations == 0);
        spmv_hip_release();
        free(fm.rowstr), free(fm.colidx), free(fm.a);
        printf("ok fragment-only edit, then an interior write (%s)\n", policy);
    }

    /* degenerate shapes: no rows, all rows empty */
    {
        int zero = 0, three = 3;
        int rs[4] = {1, 1, 1, 1};
        double yy[3] = {9, 9, 9}, dummy = 0;
        int ci0 = 1;
        spmv_harness_(yy, &dummy, &dummy, rs, &ci0, &zero);
        EXPECT(yy[0] == 9);
        spmv_harness_(yy, &dummy, &dummy, rs, &ci0, &three);
        EXPECT(yy[0] == 0 && yy[1] == 0 && yy[2] == 0);
        printf("ok degenerate shapes (%s)\n", policy);
    }
    spmv_hip_release();
    free(m.rowstr), free(m.colidx), free(m.a), free(x), free(y);
    printf("harness_driver: all scenarios passed (%s), %d uploads seen by the mock device\n", policy, mock_creates);
    return 0;
}
