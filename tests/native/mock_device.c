/*
 * tests/native/mock_device.c -- a stand-in for spmv_device.hip so that spmv_harness.c (the ABI's cache slots,
 * write detection, transfer modes, statistics) can be exercised, and run under sanitizers, on a machine
 * without a GPU.  TEST INFRASTRUCTURE ONLY: "device memory" is host memory, the "kernel" is the plain CSR
 * loop.  Nothing here is linked into libhip-spmv.so.
 */
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "spmv_internal.h"

static _Thread_local char g_err[256]; /* per thread, like the real layer: parts upload side by side */
void spmv_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *spmv_hip_last_error(void) { return g_err; }
double spmv_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
int spmv_hip_device_count(void)
{
    const char *e = getenv("MOCK_DEVICES");
    return e ? atoi(e) : 2;
}

struct spmv_hip_matrix {
    int elem, rows, cols, plan, relayout;
    int64_t nnz, first;
    int *rowptr; /* 0-based, local */
    int *col;    /* 1-based copies of the block */
    void *val;
};
static _Thread_local int g_relayout = 1;
int mock_creates, mock_relayout_creates;
void spmvd_set_relayout(int on) { g_relayout = on; }

spmv_hip_matrix *spmv_hip_matrix_create(const void *a, const int *rowstr, const int *colidx, int rows, int elem_size,
                                        int device, int row_begin, int row_end, int cols_hint)
{
    (void)device;
    for (int r = row_begin; r < row_end; ++r)
        if (rowstr[r + 1] < rowstr[r]) {
            spmv_set_error("rowstr decreases at row %d", r);
            return NULL;
        }
    spmv_hip_matrix *m = (spmv_hip_matrix *)calloc(1, sizeof *m);
    m->elem = elem_size;
    m->rows = row_end - row_begin;
    const int64_t first = (int64_t)rowstr[row_begin] - 1;
    m->nnz = (int64_t)rowstr[row_end] - rowstr[row_begin];
    m->first = first;
    m->rowptr = (int *)malloc(((size_t)m->rows + 1) * sizeof(int));
    for (int r = 0; r <= m->rows; ++r)
        m->rowptr[r] = rowstr[row_begin + r] - rowstr[row_begin];
    m->col = (int *)malloc((size_t)(m->nnz ? m->nnz : 1) * sizeof(int));
    m->val = malloc((size_t)(m->nnz ? m->nnz : 1) * (size_t)elem_size);
    memcpy(m->col, colidx + first, (size_t)m->nnz * sizeof(int));
    memcpy(m->val, (const char *)a + first * elem_size, (size_t)m->nnz * (size_t)elem_size);
    int mx = 0;
    const int64_t all = (int64_t)rowstr[rows] - rowstr[0];
    for (int64_t j = 0; j < all; ++j)
        mx = colidx[rowstr[0] - 1 + j] > mx ? colidx[rowstr[0] - 1 + j] : mx;
    m->cols = cols_hint > 0 ? cols_hint : mx;
    m->relayout = g_relayout;
    m->plan = g_relayout && m->rows >= 1000 ? SPMV_HIP_PLAN_STRIPED : SPMV_HIP_PLAN_STREAM;
    __atomic_fetch_add(&mock_creates, 1, __ATOMIC_RELAXED);
    __atomic_fetch_add(&mock_relayout_creates, m->plan == SPMV_HIP_PLAN_STRIPED, __ATOMIC_RELAXED);
    return m;
}
int mock_value_refreshes;
int spmv_hip_matrix_refresh_values(spmv_hip_matrix *m, const void *a, const int *colidx)
{
    (void)colidx;
    memcpy(m->val, (const char *)a + m->first * m->elem, (size_t)m->nnz * (size_t)m->elem);
    __atomic_fetch_add(&mock_value_refreshes, 1, __ATOMIC_RELAXED);
    return 0;
}
void spmv_hip_matrix_destroy(spmv_hip_matrix *m)
{
    if (!m)
        return;
    free(m->rowptr), free(m->col), free(m->val), free(m);
}
int spmv_hip_matrix_info(const spmv_hip_matrix *m, spmv_hip_info *out)
{
    memset(out, 0, sizeof *out);
    out->rows = m->rows, out->cols = m->cols, out->nnz = m->nnz, out->elem_size = m->elem, out->plan = m->plan;
    return 0;
}
int spmv_hip_matrix_set_plan(spmv_hip_matrix *m, int plan, int plan_param)
{
    (void)plan_param;
    m->plan = plan;
    return 0;
}
int spmv_hip_spmv_device(spmv_hip_matrix *m, const void *x, void *y, void *stream)
{
    (void)stream;
    for (int r = 0; r < m->rows; ++r) {
        if (m->elem == 8) {
            double acc = 0.0;
            for (int j = m->rowptr[r]; j < m->rowptr[r + 1]; ++j)
                acc += ((const double *)m->val)[j] * ((const double *)x)[m->col[j] - 1];
            ((double *)y)[r] = acc;
        } else {
            float acc = 0.0f;
            for (int j = m->rowptr[r]; j < m->rowptr[r + 1]; ++j)
                acc += ((const float *)m->val)[j] * ((const float *)x)[m->col[j] - 1];
            ((float *)y)[r] = acc;
        }
    }
    return 0;
}
void *spmv_hip_malloc(int device, size_t bytes)
{
    (void)device;
    return malloc(bytes ? bytes : 16);
}
void spmv_hip_free(int device, void *p)
{
    (void)device;
    free(p);
}
void *spmvd_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 16); }
void spmvd_host_free(void *p) { free(p); }
int mock_registered;
int spmvd_host_register(void *p, size_t bytes)
{
    (void)p, (void)bytes;
    mock_registered++;
    return 0;
}
void spmvd_host_unregister(void *p) { (void)p; }
int spmvd_copy_h2d_async(int device, void *dst, const void *src, size_t bytes)
{
    (void)device;
    memcpy(dst, src, bytes);
    return 0;
}
int spmvd_copy_d2h_async(int device, void *dst, const void *src, size_t bytes)
{
    (void)device;
    memcpy(dst, src, bytes);
    return 0;
}
int spmvd_stream_sync(int device)
{
    (void)device;
    return 0;
}
int spmvd_spmv_y_overlapped(spmv_hip_matrix *m, const void *x, void *y, void *y_host, int chunks)
{
    (void)chunks;
    if (getenv("MOCK_NO_OVERLAP") || m->plan != SPMV_HIP_PLAN_STREAM)
        return 0;
    spmv_hip_spmv_device(m, x, y, NULL);
    memcpy(y_host, y, (size_t)m->rows * (size_t)m->elem);
    return 1;
}
void *spmvd_host_devptr(void *p) { return getenv("MOCK_NO_DEVPTR") ? NULL : p; }
int mock_word_copies;
int spmvd_copy_words_async(int device, void *dst, const void *src, size_t bytes)
{
    (void)device;
    memcpy(dst, src, bytes);
    mock_word_copies++;
    return 0;
}
