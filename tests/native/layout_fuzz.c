/*
 * tests/native/layout_fuzz.c -- the host-side layout builders (spmv_plan.c, spmv_striped.c, spmv_lanes.c,
 * spmv_tiled.c) on a few thousand random CSR shapes, meant to be built with
 * -fsanitize=address,undefined (tests/test_native_sanitizers.py): the builders write into arrays at computed
 * offsets, which is exactly what a sanitizer is for.  No GPU code is linked; the two helpers the C files
 * expect from spmv_device.hip are stubbed here.
 */
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "spmv_internal.h"

void spmv_set_error(const char *fmt, ...)
{
    (void)fmt;
}
double spmv_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd(void)
{
    rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17;
    return rng_state;
}

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 2000;
    long built[4] = {0, 0, 0, 0}, declined[4] = {0, 0, 0, 0};
    for (int t = 0; t < cases; ++t) {
        /* mostly small shapes; every 25th is large enough for the builders' passes to split over threads */
        const int rows = (t % 25 == 24) ? 9000 + (int)(rnd() % 30000) : 1 + (int)(rnd() % 700);
        const int colsel[] = {1, 2, 15, 16, 17, 300, 5000, 9000, 20000, 70000};
        const int cols = colsel[rnd() % 10];
        const int kind = (int)(rnd() % 4);
        const int elem = (rnd() & 1) ? 8 : 4;
        int *rowstr = (int *)malloc(((size_t)rows + 1) * sizeof(int));
        int64_t nnz = 0;
        int first = 1 + (int)(rnd() % 5); /* rowstr need not start at 1 */
        rowstr[0] = first;
        for (int r = 0; r < rows; ++r) {
            int len;
            if (kind == 0)
                len = (int)(rnd() % 41);
            else if (kind == 1)
                len = (rnd() % 10 < 8) ? 0 : 1 + (int)(rnd() % 10);
            else if (kind == 2)
                len = (rnd() % 50 == 0) ? 500 + (int)(rnd() % 2500) : (int)(rnd() % 6);
            else
                len = 1 + (int)(rnd() % 3);
            nnz += len;
            rowstr[r + 1] = first + (int)nnz;
        }
        const size_t total = (size_t)(first - 1) + (size_t)nnz;
        int *colidx = (int *)malloc((total ? total : 1) * sizeof(int));
        void *a = malloc((total ? total : 1) * (size_t)elem);
        for (size_t j = 0; j < total; ++j) {
            colidx[j] = 1 + (int)(rnd() % (uint64_t)cols);
            if (elem == 8)
                ((double *)a)[j] = (double)(rnd() % 1000) / 7.0 - 70.0;
            else
                ((float *)a)[j] = (float)(rnd() % 1000) / 7.0f - 70.0f;
        }
        /* builders take the pointers as the ABI delivers them: entry j of the arrays is at index j, the
         * first row begins at rowstr[0] - 1 */
        spmv_row_stats st;
        spmv_plan_row_stats(rowstr, 0, rows, &st);
        spmv_stream_plan sp;
        if (spmv_plan_stream_build(rowstr, 0, rows, SPMV_STREAM_CAP - 4, SPMV_STREAM_ROWCAP, SPMV_LONG_SEG, &sp) == 0)
            spmv_plan_stream_free(&sp);
        spmv_bin_plan bp;
        if (spmv_plan_bins_build(rowstr, 0, rows, SPMV_STREAM_CAP - 4, &bp) == 0)
            spmv_plan_bins_free(&bp);
        int bounds[9];
        spmv_hip_partition_rows(rowstr, rows, 1 + (int)(rnd() % 8), bounds);

        const int xb = (rnd() & 1) ? 65536 : 131072, wpb = (rnd() & 1) ? 8 : 16, wgs = 1 + (int)(rnd() % 40);
        spmv_striped_host sh;
        spmv_striped_opts so = {xb, wpb, wgs, 64 << (rnd() % 5), (int)(rnd() % 4) == 0 ? 1 + (int)(rnd() % 9) : 0};
        int rc = spmv_striped_build(a, rowstr, colidx, 0, rows, cols, elem, &so, &sh);
        if (rc == 0) {
            built[0]++;
            spmv_striped_free(&sh);
        } else if (rc == -2) {
            declined[0]++;
        } else {
            fprintf(stderr, "striped rc=%d\n", rc);
            return 1;
        }
        spmv_tiled_host th;
        rc = spmv_tiled_build(a, rowstr, colidx, 0, rows, cols, elem, 1 + (int)(rnd() % 30), &th);
        if (rc == 0) {
            built[2]++;
            spmv_tiled_free(&th);
        } else if (rc == -2) {
            declined[2]++;
        } else {
            fprintf(stderr, "tiled rc=%d\n", rc);
            return 1;
        }
        /* BLOCKED's host side (spmv_blocked_plan.c): geometry, then the plan from one count per (stripe, bin) cell */
        {
            const int rb_sel[] = {0, 64, 128, 1024, 2560};
            spmv_blocked_plan g;
            if (spmv_blocked_geometry(rows, cols, elem, rb_sel[rnd() % 5], &g) == 0) {
                int *cnt = (int *)calloc((size_t)g.S * g.B, sizeof(int));
                if (!cnt)
                    return 1;
                for (int r = 0; r < rows; ++r)
                    for (int j = rowstr[r] - 1; j < rowstr[r + 1] - 1; ++j)
                        cnt[(size_t)((colidx[j] - 1) / g.wcols) * g.B + r / g.RB]++;
                rc = spmv_blocked_plan_build(cnt, nnz, (int)(rnd() & 1), &g);
                if (rc == 0) {
                    /* every cell inside both streams, the pieces' windows ascending up to the end of phase 2's */
                    int bad = g.n1 % SPMV_BLOCKED_WIN || g.n2 % SPMV_BLOCKED_WIN || g.c1q[(size_t)g.S * g.B] != (uint32_t)(g.n1 / 4) ||
                              g.pw[g.n_pieces] != (int)(g.n2 / SPMV_BLOCKED_WIN) || g.n_pieces < g.B;
                    for (size_t id = 0; id < (size_t)g.S * g.B && !bad; ++id)
                        bad = (int64_t)g.c1q[id] * 4 + cnt[id] > g.n1 || (int64_t)g.c2q[id] * 4 + cnt[id] > g.n2;
                    for (int i = 0; i < g.n_pieces && !bad; ++i)
                        bad = g.pw[i] > g.pw[i + 1] || g.pb[i] < 0 || g.pb[i] >= g.B || g.ps[i] >= g.n_slots;
                    if (bad) {
                        fprintf(stderr, "case %d: inconsistent BLOCKED plan\n", t);
                        return 1;
                    }
                    built[3]++;
                    spmv_blocked_plan_free(&g);
                } else if (rc == -2 || (rc == -1 && nnz == 0)) {
                    declined[3]++;
                } else {
                    fprintf(stderr, "case %d: spmv_blocked_plan_build rc=%d\n", t, rc);
                    return 1;
                }
                free(cnt);
            }
        }
        /* HUB's host side (spmv_hub_plan.c): the two tiers of hub columns from one count per column, then the chunks */
        {
            int *cnt = (int *)calloc((size_t)cols, sizeof(int));
            if (!cnt)
                return 1;
            for (int j = first - 1; j < first - 1 + (int)nnz; ++j)
                cnt[colidx[j] - 1]++;
            const int ksel[] = {0, 1, 7, 300, 16352, 32736}, k2sel[] = {0, 0, 5, 4000, 262144};
            const int kmax = ksel[rnd() % 6], kmax2 = k2sel[rnd() % 5], ch = (rnd() & 1) ? 1024 : 64;
            spmv_hub_plan g;
            memset(&g, 0, sizeof g);
            rc = spmv_hub_select(cnt, cols, kmax, kmax2, 1 + (int)(rnd() % 3), &g);
            if (rc == 0)
                rc = spmv_hub_rows(rowstr, rowstr[0], rows, ch, &g);
            if (rc == 0) {
                int64_t cov = 0;
                int bad = g.K > kmax || g.K2 > kmax2 || g.n_chunks != (int)((nnz + ch - 1) / ch) || g.rank0[g.n_chunks] != g.n_nzrows ||
                          g.n_nzrows + g.n_empty != rows;
                for (int k = 0; k < g.K + g.K2 && !bad; ++k) {
                    bad = g.hubcol[k] < 0 || g.hubcol[k] >= cols || g.slot_of[g.hubcol[k]] != k;
                    cov += cnt[g.hubcol[k]];
                }
                for (int c = 0; c + 1 <= g.n_chunks && !bad; ++c)
                    bad = g.rank0[c] > g.rank0[c + 1];
                for (int k = 0; k < g.n_fix && !bad; ++k)
                    bad = g.fix[3 * k + 1] >= g.fix[3 * k + 2] || g.fix[3 * k + 2] >= g.n_chunks || g.fix[3 * k] < 0 || g.fix[3 * k] >= rows;
                if (bad || cov != g.covered + g.covered2) {
                    fprintf(stderr, "case %d: inconsistent HUB plan\n", t);
                    return 1;
                }
                built[1]++;
            } else if (rc != -3) {
                fprintf(stderr, "case %d: HUB plan rc=%d\n", t, rc);
                return 1;
            }
            spmv_hub_plan_free(&g);
            free(cnt);
        }
        free(rowstr), free(colidx), free(a);
    }
    printf("layout_fuzz: %d cases; striped %ld built / %ld declined; tiled %ld / %ld; blocked plan %ld / %ld; hub plan %ld\n", cases, built[0],
           declined[0], built[2], declined[2], built[3], declined[3], built[1]);
    return 0;
}
