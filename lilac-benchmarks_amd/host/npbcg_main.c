/*
 * host/npbcg_main.c -- command-line NPB CG driver:  npbcg <class> <backend.so> [repeats]
 *
 * Loads a backend the way the reference's pagerank does
 * (/root/reference/pagerank/main.cpp:13-45: dlopen(RTLD_NOW) + dlsym of
 * "spmv_harness_"), runs the benchmark `repeats` times and finishes with one CSV
 * row in the reference's shape `platform,NPB,impl,class,t1,...`
 * (/root/reference/NPB3.3.1/run_all:36-41), where impl is derived from
 * lib<impl>-spmv.so like pagerank/main.cpp:49-56 does.
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "npb_cg.h"

static void impl_label(const char *path, char *out, size_t n)
{
    const char *b = strrchr(path, '/');
    b = b ? b + 1 : path;
    snprintf(out, n, "%s", b);
    if (!strncmp(out, "lib", 3))
        memmove(out, out + 3, strlen(out + 3) + 1);
    char *s = strstr(out, "-spmv.so");
    if (!s)
        s = strstr(out, ".so");
    if (s)
        *s = 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "Usage: %s <class S|W|A|B|C|D> <path/to/lib<impl>-spmv.so> [repeats]\n", argv[0]);
        return 1;
    }
    const char cls = argv[1][0];
    const int repeats = argc > 3 ? atoi(argv[3]) : 1;
    void *lib = dlopen(argv[2], RTLD_NOW);
    if (!lib) {
        fprintf(stderr, "Couldn't load %s\n%s\n", argv[2], dlerror());
        return 2;
    }
    spmv_harness_fn fn = (spmv_harness_fn)dlsym(lib, "spmv_harness_");
    if (!fn) {
        fprintf(stderr, "Error loading SPMV implementation:\n%s\n", dlerror());
        return 3;
    }
    npb_matrix m;
    int rc = npb_makea(cls, &m);
    if (rc) {
        fprintf(stderr, "npb_makea(%c) failed: %d\n", cls, rc);
        return 4;
    }
    fprintf(stderr, "class %c: na=%d nnz=%lld generated in %.2f s\n", cls, m.cls.na, (long long)m.nnz,
            m.t_gen);
    char impl[128];
    impl_label(argv[2], impl, sizeof impl);
    double *t = calloc((size_t)repeats, sizeof *t);
    int all_ok = 1;
    for (int k = 0; k < repeats; ++k) {
        npb_result r;
        if (npb_cg_run(&m, fn, 0, k == 0, &r)) {
            fprintf(stderr, "npb_cg_run failed\n");
            return 5;
        }
        if (k == 0)
            npb_print_result(&r);
        t[k] = r.t_bench;
        all_ok &= r.verified;
    }
    const char *platform = getenv("SPMV_PLATFORM") ? getenv("SPMV_PLATFORM") : "mi355x";
    printf("%s,NPB,%s,%c", platform, impl, cls);
    for (int k = 0; k < repeats; ++k)
        printf(",%.4f", t[k]);
    printf("\n");
    free(t);
    npb_matrix_free(&m);
    dlclose(lib);
    return all_ok ? 0 : 6;
}
