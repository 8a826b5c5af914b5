/*
 * oracle/spmv_oracle_mt.c -- TEST / BASELINE INFRASTRUCTURE ONLY (see spmv_oracle.c).
 *
 * The reference's sequential loop (libspmv/native-impl.c:1-12) run on several host threads, each
 * on a contiguous block of rows of about equal nonzero count.  Every row is still added up left
 * to right by one thread, so the result is bit-identical to oracle_spmv / native.c for any thread
 * count; this is the "all host cores" context line of bench.py (an optional
 * row-parallel CPU variant, labelled separately from the faithful single-core baseline).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct mt_task {
    pthread_t th;
    double *ov;
    const double *a, *iv;
    const int *rowstr, *colidx;
    int r0, r1;
} mt_task;

static void *mt_run(void *v)
{
    mt_task *t = (mt_task *)v;
    for (int r = t->r0; r < t->r1; ++r) {
        double acc = 0.0;
        for (int k = t->rowstr[r] - 1; k < t->rowstr[r + 1] - 1; ++k) {
            const double prod = t->a[k] * t->iv[t->colidx[k] - 1];
            acc = acc + prod;
        }
        t->ov[r] = acc;
    }
    return NULL;
}

/* returns the number of threads actually used */
int oracle_spmv_mt(double *ov, const double *a, const double *iv, const int *rowstr, const int *colidx, int rows,
                   int nthreads)
{
    if (nthreads < 1)
        nthreads = 1;
    if (nthreads > 256)
        nthreads = 256;
    if (rows < nthreads)
        nthreads = rows > 0 ? rows : 1;
    mt_task *task = (mt_task *)calloc((size_t)nthreads, sizeof *task);
    if (!task)
        return 0;
    const int64_t base = rows > 0 ? rowstr[0] : 1, nnz = rows > 0 ? (int64_t)rowstr[rows] - base : 0;
    int r = 0;
    for (int i = 0; i < nthreads; ++i) {
        task[i].ov = ov, task[i].a = a, task[i].iv = iv, task[i].rowstr = rowstr, task[i].colidx = colidx;
        task[i].r0 = r;
        const int64_t target = nnz * (i + 1) / nthreads;
        while (r < rows && (i == nthreads - 1 || (int64_t)rowstr[r + 1] - base <= target))
            ++r;
        task[i].r1 = r;
    }
    int started = 0;
    for (int i = 1; i < nthreads; ++i) {
        if (pthread_create(&task[i].th, NULL, mt_run, &task[i]) != 0)
            break;
        started = i;
    }
    mt_run(&task[0]);
    for (int i = 1; i <= started; ++i)
        pthread_join(task[i].th, NULL);
    for (int i = started + 1; i < nthreads; ++i)
        mt_run(&task[i]);
    free(task);
    return nthreads;
}
