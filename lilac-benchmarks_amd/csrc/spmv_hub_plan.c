/*
 * csrc/spmv_hub_plan.c -- the host side of the HUB plan (spmv_hub.hip): every DECISION of the layout -- which columns of x
 * are hubs and get a slot of the LDS table, how the entry stream is cut into chunks, which row ends a chunk starts with,
 * which further columns get a slot of the dense table in memory behind it,
 * which rows cross a chunk boundary and are finished by the second launch, which rows are empty -- from one count per
 * column (taken on the device) and the block's row pointers.  Plain C so that the CPU tests can walk the result and
 * restate the kernel's data flow on top of it (tests/test_hub_layout.py); the data movement is the .hip file's.
 *
 * What the plan is for (profiles/r04_stream_ablation_rmat.log): on power-law graphs (bfs/library.cc:268-279,
 * pagerank/main.cpp:128-149 call one product per level / iteration over the whole adjacency matrix) the CSR kernel spends
 * half its time on gather misses and 3 % on the per-row work; a few thousand hub columns take most of the gathers
 * (R-MAT 2^20: 16 384 of 1 M columns hold 58 % of the entries).
 */
#include <stdlib.h>
#include <string.h>

#include "spmv_internal.h"

void spmv_hub_plan_free(spmv_hub_plan *g)
{
    free(g->hubcol), free(g->slot_of), free(g->rank0), free(g->nzrow), free(g->erow), free(g->fix);
    memset(g, 0, sizeof *g);
}

/* slots of the LDS table: what 160 KB leave next to the waves' rings of row sums (32 KB) and the ticket; the table's last
 * element + 1 is the zero the pads multiply with */
int spmv_hub_lds_slots(int elem_size) { return elem_size == 8 ? 16352 : 32736; }

/*
 * Marks the kmax columns referenced most often among those with tier[c] == 0 -- ties to the lower column, none referenced
 * fewer than min_count times -- with `mark`.  Returns how many, -3 without memory.
 */
static int64_t top_columns(const int *colcount, int cols, int64_t kmax, int min_count, unsigned char *tier, unsigned char mark)
{
    int maxc = 0;
    for (int c = 0; c < cols; ++c)
        if (!tier[c] && colcount[c] > maxc)
            maxc = colcount[c];
    /* how many columns have each count; from the top down until kmax are taken */
    int64_t *byc = (int64_t *)calloc((size_t)maxc + 2, sizeof(int64_t));
    if (!byc)
        return -3;
    for (int c = 0; c < cols; ++c)
        if (!tier[c])
            byc[colcount[c]]++;
    if (min_count < 1)
        min_count = 1;
    int t = maxc + 1; /* every column with count >= t is taken */
    int64_t taken = 0;
    while (t - 1 >= min_count && taken + byc[t - 1] <= kmax) {
        --t;
        taken += byc[t];
    }
    /* of the columns with count t - 1 the lowest ones fill what is left */
    int64_t partial = (t - 1 >= min_count) ? kmax - taken : 0;
    free(byc);
    int64_t k = 0;
    for (int c = 0; c < cols; ++c) {
        if (tier[c])
            continue;
        int take = colcount[c] >= t && colcount[c] >= min_count;
        if (!take && partial > 0 && colcount[c] == t - 1) {
            take = 1;
            --partial;
        }
        if (take) {
            tier[c] = mark;
            ++k;
        }
    }
    return k;
}

/*
 * The hubs, in two tiers.  Tier 0: the kmax columns referenced most often, ties to the lower column, none referenced fewer
 * than min_count times (a slot costs one gather when the table is packed: a column referenced once gains nothing) -- they
 * get the slots [0, K) of the LDS table.  Tier 1: the kmax2 most referenced of the REST, same rule -- slots [K, K + K2) of a
 * dense table in memory that the cache holds where it cannot hold the lines of x they are scattered over (one 128-byte line
 * per column otherwise).  hubcol ascending by column within each tier (the pack kernel's gathers then walk x front to back),
 * slot_of[c] = slot or -1.  covered / covered2 = entries that hit a tier-0 / tier-1 column.
 * 0, or -3 without memory.
 */
int spmv_hub_select(const int *colcount, int cols, int kmax, int kmax2, int min_count, spmv_hub_plan *g)
{
    g->K = g->K2 = 0;
    g->covered = g->covered2 = 0;
    g->hubcol = NULL;
    g->slot_of = (int *)malloc((size_t)(cols > 0 ? cols : 1) * sizeof(int));
    unsigned char *tier = (unsigned char *)calloc((size_t)(cols > 0 ? cols : 1), 1);
    if (!g->slot_of || !tier) {
        free(tier);
        return -3;
    }
    const int64_t K = top_columns(colcount, cols, kmax > 0 ? kmax : 0, min_count, tier, 1);
    const int64_t K2 = K < 0 ? K : (kmax2 > 0 ? top_columns(colcount, cols, kmax2, min_count, tier, 2) : 0);
    if (K < 0 || K2 < 0) {
        free(tier);
        return -3;
    }
    g->hubcol = (int *)malloc((size_t)(K + K2 > 0 ? K + K2 : 1) * sizeof(int));
    if (!g->hubcol) {
        free(tier);
        return -3;
    }
    int k0 = 0, k1 = (int)K;
    for (int c = 0; c < cols; ++c) {
        if (tier[c] == 1) {
            g->slot_of[c] = k0;
            g->hubcol[k0++] = c;
            g->covered += colcount[c];
        } else if (tier[c] == 2) {
            g->slot_of[c] = k1;
            g->hubcol[k1++] = c;
            g->covered2 += colcount[c];
        } else {
            g->slot_of[c] = -1;
        }
    }
    g->K = (int)K, g->K2 = (int)K2;
    g->cols_used = 0;
    for (int c = 0; c < cols; ++c)
        g->cols_used += colcount[c] > 0;
    free(tier);
    return 0;
}

/*
 * The rows.  rowptr - base: the block's 0-based row pointers (rows + 1; the caller's are 1-based: base = rowptr[0]).  The entry stream is cut into chunks
 * of CH entries, whatever the rows: chunk c = entries [c CH, (c + 1) CH).
 *   rank0[c]   the number of row ENDS before chunk c (rank0[n_chunks] = all of them): a wave numbers the row ends it meets
 *              from there; nzrow[rank] is the row that ends there (identity != 0: no row is empty, rank IS the row);
 *   fix        rows whose entries lie in more than one chunk: {row, first chunk, chunk of its end}: the main kernel leaves
 *              y[row] = the piece inside the last chunk and carry[c] = what chunk c holds behind its last row end; the
 *              second launch adds carry[first .. last) and the piece, in that order;
 *   erow       the empty rows (y = 0, native-impl.c:6,9), written by the second launch as well.
 * 0, -3 without memory.
 */
int spmv_hub_rows(const int *rowptr, int base, int rows, int ch, spmv_hub_plan *g)
{
    const int64_t nnz = rows > 0 ? (int64_t)rowptr[rows] - base : 0;
    g->CH = ch;
    g->n_chunks = (int)((nnz + ch - 1) / ch);
    g->n_nzrows = g->n_empty = g->n_fix = 0;
    for (int r = 0; r < rows; ++r) {
        const int f = rowptr[r] - base, e = rowptr[r + 1] - base;
        if (e == f) {
            g->n_empty++;
            continue;
        }
        g->n_nzrows++;
        if ((e - 1) / ch > f / ch)
            g->n_fix++;
    }
    g->identity = g->n_empty == 0;
    g->rank0 = (int *)malloc(((size_t)g->n_chunks + 1) * sizeof(int));
    g->nzrow = (int *)malloc(((size_t)g->n_nzrows + 1) * sizeof(int));
    g->erow = (int *)malloc(((size_t)g->n_empty + 1) * sizeof(int));
    g->fix = (int *)malloc(((size_t)g->n_fix * 3 + 1) * sizeof(int));
    if (!g->rank0 || !g->nzrow || !g->erow || !g->fix)
        return -3;
    int k = 0, ne = 0, nf = 0, c = 0;
    for (int r = 0; r < rows; ++r) {
        const int f = rowptr[r] - base, e = rowptr[r + 1] - base;
        if (e == f) {
            g->erow[ne++] = r;
            continue;
        }
        const int cl = (e - 1) / ch; /* the chunk the row ends in */
        while (c <= cl)
            g->rank0[c++] = k;       /* every chunk up to that one starts with row end k */
        g->nzrow[k++] = r;
        if (cl > f / ch) {
            g->fix[3 * nf] = r, g->fix[3 * nf + 1] = f / ch, g->fix[3 * nf + 2] = cl;
            ++nf;
        }
    }
    while (c <= g->n_chunks)
        g->rank0[c++] = k;
    return 0;
}
