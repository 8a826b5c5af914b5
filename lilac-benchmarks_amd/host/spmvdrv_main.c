/*
 * host/spmvdrv_main.c -- command-line counterparts of the reference's remaining benchmark
 * programs, each ending with a CSV row in the reference's shape `platform,bench,impl,size,t...`
 * (SparseBench/run_all:62,71, bfs/run_all:33, parboil/run_all:33, pagerank/main.cpp:184-186).
 *
 *   spmvdrv sb       <backend.so> <size | crsmatNNNu file> [seed]
 *   spmvdrv bfs      <backend.so> <pairs file | rmat:SCALE:EF | grid:N> [sources]
 *   spmvdrv pagerank <backend.so> <file.mtx | rmat:SCALE:EF | grid:N> [iters] [runs]
 *   spmvdrv parboil  <backend.so> <matrix.mtx> <vector.bin> [out file] [golden .out]
 *   spmvdrv pathsample <backend.so> <nmin> [npfold] [max neighbours] [seed]
 *   spmvdrv reload   <backend.so> <cycles>      dlopen / multiply / dlclose / write to the arrays, repeated
 *
 * The backend is loaded like pagerank/main.cpp:13-45 does (dlopen RTLD_NOW + dlsym).
 */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "callers.h"

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void impl_label(const char *path, char *out, size_t n)
{
    const char *b = strrchr(path, '/');
    snprintf(out, n, "%s", b ? b + 1 : path);
    if (!strncmp(out, "lib", 3))
        memmove(out, out + 3, strlen(out + 3) + 1);
    char *s = strstr(out, "-spmv.so");
    if (!s)
        s = strstr(out, ".so");
    if (s)
        *s = 0;
}

static uint64_t fnv(const void *p, size_t n)
{
    const unsigned char *c = p;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i)
        h = (h ^ c[i]) * 0x100000001b3ull;
    return h;
}

static int load_graph(const char *spec, csr_d *m, int as_bfs)
{
    int a, b;
    if (sscanf(spec, "rmat:%d:%d", &a, &b) == 2)
        return graph_rmat(a, b, 20, m);
    if (sscanf(spec, "grid:%d", &a) == 1)
        return graph_grid2d(a, a, m);
    if (as_bfs)
        return bfs_read_pairs(spec, m);
    return mm_read_csr(spec, m);
}

/*
 * pagerank/main.cpp:13-45 loads the backend in a constructor and dlclose()s it in the destructor while the caller's
 * arrays are still alive.  A backend that caches them (and write-protects them to notice edits) has to let go of
 * them at dlclose: after it, the caller writes to every array, and then loads the library again.
 */
static int reload_cycles(const char *path, int cycles)
{
    csr_d m;
    if (graph_grid2d(600, 600, &m)) /* 360 000 rows, 1.8 M nonzeros: rowstr/colidx/a are 350-3500 pages each */
        return 4;
    double *x = malloc(((size_t)m.cols + 2) * sizeof(double)), *y = malloc(((size_t)m.rows + 2) * sizeof(double));
    for (int i = 0; i < m.cols + 2; ++i)
        x[i] = 1.0 + (double)(i % 7);
    for (int c = 0; c < cycles; ++c) {
        void *lib = dlopen(path, RTLD_NOW);
        if (!lib) {
            fprintf(stderr, "cycle %d: couldn't load %s\n%s\n", c, path, dlerror());
            return 2;
        }
        d_harness_fn fn = (d_harness_fn)dlsym(lib, "spmv_harness_");
        if (!fn)
            return 3;
        for (int rep = 0; rep < 3; ++rep) /* the second and third call find the cached copy */
            fn(y, m.a, x, m.rowstr, m.colidx, &m.rows);
        int64_t bad = 0;
        for (int r = 0; r < m.rows; ++r) {
            double s = 0.0; /* small integers times c+1: exact in any order */
            for (int j = m.rowstr[r] - 1; j < m.rowstr[r + 1] - 1; ++j)
                s += m.a[j] * x[m.colidx[j] - 1];
            bad += s != y[r];
        }
        dlclose(lib);
        /* the library is gone (or at least told to go): all three arrays must be ordinary writable memory again */
        for (int64_t j = 0; j < m.nnz; j += 512) {
            m.a[j] *= 2.0;
            m.colidx[j] = m.colidx[j];
        }
        for (int r = 0; r <= m.rows; r += 1024)
            m.rowstr[r] = m.rowstr[r];
        printf("cycle %d: %lld rows differ; arrays written after dlclose\n", c, (long long)bad);
        if (bad)
            return 6;
    }
    free(x), free(y);
    csr_d_free(&m);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "Usage: %s sb|bfs|pagerank|parboil|pathsample|reload <backend.so> <input> [...]\n", argv[0]);
        return 1;
    }
    const char *bench = argv[1];
    if (!strcmp(bench, "reload"))
        return reload_cycles(argv[2], atoi(argv[3]));
    void *lib = dlopen(argv[2], RTLD_NOW);
    if (!lib) {
        fprintf(stderr, "Couldn't load %s\n%s\n", argv[2], dlerror());
        return 2;
    }
    d_harness_fn dfn = (d_harness_fn)dlsym(lib, "spmv_harness_");
    f_harness_fn ffn = (f_harness_fn)dlsym(lib, "f_spmv_harness_");
    if (!dfn || !ffn) {
        fprintf(stderr, "Error loading SPMV implementation:\n%s\n", dlerror());
        return 3;
    }
    char impl[128];
    impl_label(argv[2], impl, sizeof impl);
    const char *platform = getenv("SPMV_PLATFORM") ? getenv("SPMV_PLATFORM") : "mi355x";

    if (!strcmp(bench, "sb")) {
        csr_d m;
        int size = atoi(argv[3]);
        int rc = size > 0 ? sb_generate(size, argc > 4 ? strtoull(argv[4], 0, 10) : 170, &m) : sb_read_crs(argv[3], &m);
        if (rc) {
            fprintf(stderr, "cannot get matrix %s: %d\n", argv[3], rc);
            return 4;
        }
        double *hist = calloc(128, sizeof(double)), *x = calloc((size_t)m.rows + 2, sizeof(double));
        int64_t calls = 0;
        double tm = 0.0;
        double t0 = now_s();
        int its = sb_bicg(&m, dfn, 100, 1e-6, hist, x, &calls, &tm); /* maxit main.f:26, rtol main.f:365 */
        double tt = now_s() - t0;
        printf("n=%d nnz=%lld its=%d spmv_calls=%lld ||r0||=%.6e ||r_last||=%.6e\n", m.rows, (long long)m.nnz, its,
               (long long)calls, hist[0], hist[(its < 0 ? -its : its) - 1]);
        printf("Matrix multiply total time: %.6f s   (solver total %.6f s)\n", tm, tt);
        printf("%s,SparseBench,%s,%s,%.6f\n", platform, impl, argv[3], tm);
        free(hist), free(x);
        csr_d_free(&m);
    } else if (!strcmp(bench, "bfs")) {
        csr_d m;
        int rc = load_graph(argv[3], &m, 1);
        if (rc) {
            fprintf(stderr, "cannot get graph %s: %d\n", argv[3], rc);
            return 4;
        }
        const int nsrc = argc > 4 ? atoi(argv[4]) : 16; /* bfs.cc:84 */
        double *dist = malloc(((size_t)m.rows + 2) * sizeof(double));
        uint64_t h = 0;
        int levels = 0;
        double t0 = now_s();
        for (int k = 0; k < nsrc; ++k) { /* fixed sources instead of std::random_device (bfs.cc:80-90) */
            int s = (int)((uint64_t)(k * 2654435761u) % (uint64_t)m.rows);
            levels += bfs_run(&m, dfn, s, dist);
            h ^= fnv(dist, (size_t)m.rows * sizeof(double)) + k;
        }
        double t = now_s() - t0;
        printf("rows=%d nnz=%lld sources=%d levels=%d distances_fnv=%016llx\n", m.rows, (long long)m.nnz, nsrc,
               levels, (unsigned long long)h);
        printf("%f\n", t); /* bfs.cc:95 */
        printf("%s,BFS,%s,%s,%.6f\n", platform, impl, argv[3], t);
        free(dist);
        csr_d_free(&m);
    } else if (!strcmp(bench, "pagerank")) {
        csr_d m;
        int rc = load_graph(argv[3], &m, 0);
        if (rc || m.rows != m.cols) {
            fprintf(stderr, "cannot get a square matrix from %s: %d\n", argv[3], rc);
            return 4;
        }
        const double d = 0.85;
        pagerank_prepare(&m, d);
        const int iters = argc > 4 ? atoi(argv[4]) : 1024, runs = argc > 5 ? atoi(argv[5]) : 5; /* main.cpp:120-121 */
        double *x = malloc(((size_t)m.rows + 2) * sizeof(double));
        double sum = 0.0;
        for (int i = 0; i < m.rows; ++i) { /* fixed start vector instead of std::random_device (main.cpp:82-98) */
            x[i] = (double)((i * 2654435761u) % 1000003u + 1) / 1000003.0;
            sum += x[i];
        }
        for (int i = 0; i < m.rows; ++i)
            x[i] /= sum;
        printf("%s,PageRank,%s,%s", platform, impl, argv[3]);
        double err = 0.0;
        for (int r = 0; r < runs; ++r) {
            double t0 = now_s();
            err = pagerank_run(&m, dfn, d, iters, x);
            printf(",%.6f", now_s() - t0);
        }
        printf("\nrows=%d nnz=%lld iters=%d last_error=%.6e x_fnv=%016llx\n", m.rows, (long long)m.nnz, iters, err,
               (unsigned long long)fnv(x, (size_t)m.rows * sizeof(double)));
        free(x);
        csr_d_free(&m);
    } else if (!strcmp(bench, "parboil")) {
        if (argc < 5) {
            fprintf(stderr, "parboil needs <matrix.mtx> <vector.bin>\n");
            return 1;
        }
        csr_f m;
        int rc = pb_coo_to_csr(argv[3], 1, 0, &m); /* main.c:66-69: mirrored, not binary */
        if (rc) {
            fprintf(stderr, "cannot read %s: %d\n", argv[3], rc);
            return 4;
        }
        float *x = calloc((size_t)m.cols + 2, sizeof(float)), *y = calloc((size_t)m.rows + 2, sizeof(float));
        if (pb_read_vec(argv[4], x, m.rows)) {
            fprintf(stderr, "cannot read %s\n", argv[4]);
            return 4;
        }
        double t0 = now_s();
        pb_run(&m, ffn, x, y, 50); /* main.c:93 */
        double t = now_s() - t0;
        printf("rows=%d nnz=%lld compute=%.6f s (50 calls)\n", m.rows, (long long)m.nnz, t);
        if (argc > 5 && strcmp(argv[5], "-"))
            pb_write_out(argv[5], y, m.rows);
        int bad = -1;
        if (argc > 6) {
            float *g;
            int gn;
            if (!pb_read_out(argv[6], &g, &gn) && gn == m.rows) {
                bad = pb_compare(g, y, gn);
                printf("compare-output: %s (%d mismatches of %d)\n", bad ? "Mismatch" : "Pass", bad, gn);
                free(g);
            }
        }
        printf("%s,parboil-spmv,%s,%s,%.6f\n", platform, impl, argv[3], t);
        free(x), free(y);
        csr_f_free(&m);
        if (bad > 0)
            return 6;
    } else if (!strcmp(bench, "pathsample")) {
        const int nmin = atoi(argv[3]), npfold = argc > 4 ? atoi(argv[4]) : 100;
        ps_net net;
        int rc = ps_generate(nmin, argc > 5 ? atoi(argv[5]) : 12, argc > 6 ? strtoull(argv[6], 0, 10) : 38, &net);
        if (rc) {
            fprintf(stderr, "cannot make a network of %d minima: %d\n", nmin, rc);
            return 4;
        }
        double *gp = malloc(((size_t)nmin + 2) * sizeof(double));
        printf("%s,PATHSAMPLE,%s,%d", platform, impl, nmin);
        for (int variant = 0; variant < 2; ++variant) { /* PFOLD (Pfold.f90:219-223), then NGT (NGT.f:999-1002) */
            double t = 0.0;
            const int64_t nnz = ps_committor(&net, dfn, npfold, variant, gp, &t);
            if (nnz < 0)
                return 4;
            printf(",%.6f", t);
            fprintf(stderr, "%s: nmin=%d nonzero=%lld npfold=%d  %.2f us per call  gpfold_fnv=%016llx\n",
                    variant ? "NGT   (nothing changes between calls)" : "PFOLD (the `a` array rewritten before every call)",
                    nmin, (long long)nnz, npfold, 1e6 * t / (npfold > 0 ? npfold : 1),
                    (unsigned long long)fnv(gp, (size_t)nmin * sizeof(double)));
        }
        printf("\n");
        free(gp);
        ps_free(&net);
    } else {
        fprintf(stderr, "unknown benchmark %s\n", bench);
        return 1;
    }
    dlclose(lib);
    return 0;
}
