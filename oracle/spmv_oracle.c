/*
 * oracle/spmv_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's sequential CSR y = A*x, used only as the
 * parity checker (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
 * leg).  The shipped library (libhip-spmv.so) never links, loads or calls
 * anything in this directory.
 *
 * Follows /root/reference/libspmv/native-impl.c:1-12 (double) and :14-25
 * (float), exported behind the ABI of /root/reference/libspmv/native.c:3-11:
 *
 *   - rowstr and colidx are 1-based (native-impl.c:7-8: j from rowstr[i]-1 to
 *     rowstr[i+1]-2, column colidx[j]-1);
 *   - one accumulator per row, products added left to right, the product is
 *     rounded before the add (the reference is built -std=c99 -O3,
 *     libspmv/Makefile:1, i.e. ISO mode => no FMA contraction, no
 *     reassociation).  This file MUST be compiled with -ffp-contract=off and
 *     without -ffast-math so that the bit pattern of every y[i] is the
 *     reference's;
 *   - every ov[i], i in [0,*rows), is overwritten; an empty row writes 0.0
 *     (native-impl.c:6,9).
 *
 * Parity pinning: this restatement is checked bit-for-bit (a) against the
 * reference's own known-answer test, libspmv/test.cpp:44-54, (b) against
 * oracle/_ref/native.so, the reference's native.c + native-impl.c compiled
 * from where they lie under /root/reference by oracle/Makefile, on seeded
 * random matrices (tests/test_oracle.py), and (c) through the golden vectors
 * under tests/golden/ that were produced with oracle/_ref/native.so by
 * tests/golden/make_golden.py.
 */
#include <stddef.h>
#include <stdint.h>

#if defined(__FAST_MATH__)
#error "the oracle must not be built with -ffast-math"
#endif

#pragma STDC FP_CONTRACT OFF

/* native-impl.c:1-12 */
void *oracle_spmv(double *ov, double *a, double *iv, int *rowstr, int *colidx, int *rows)
{
    const int n = *rows;
    for (int r = 0; r < n; ++r) {
        const int lo = rowstr[r] - 1;
        const int hi = rowstr[r + 1] - 1;
        double acc = 0.0;
        for (int k = lo; k < hi; ++k) {
            const double prod = a[k] * iv[colidx[k] - 1];
            acc = acc + prod;
        }
        ov[r] = acc;
    }
    return NULL;
}

/* native-impl.c:14-25 */
void *oracle_f_spmv(float *ov, float *a, float *iv, int *rowstr, int *colidx, int *rows)
{
    const int n = *rows;
    for (int r = 0; r < n; ++r) {
        const int lo = rowstr[r] - 1;
        const int hi = rowstr[r + 1] - 1;
        float acc = 0.0f;
        for (int k = lo; k < hi; ++k) {
            const float prod = a[k] * iv[colidx[k] - 1];
            acc = acc + prod;
        }
        ov[r] = acc;
    }
    return NULL;
}

/* native.c:3-6 / :8-11 -- the two ABI symbols, so that the oracle can stand in
 * as "the reference CPU backend" under the same drivers the HIP backend is
 * measured with (bench.py cpu_baseline, kind "port"). */
void *spmv_harness_(double *ov, double *a, double *iv, int *rowstr, int *colidx, int *rows)
{
    return oracle_spmv(ov, a, iv, rowstr, colidx, rows);
}

void *f_spmv_harness_(float *ov, float *a, float *iv, int *rowstr, int *colidx, int *rows)
{
    return oracle_f_spmv(ov, a, iv, rowstr, colidx, rows);
}

/*
 * Per-row condition bound used by the parity tests of the (re-ordering) HIP
 * kernels: absum[i] = sum_j |a[j] * x[col[j]]|.  Any summation order of the
 * row's products differs from the sequential one by at most
 * (len-1) * eps * absum[i] (Higham, Accuracy and Stability, section 4.2), so
 * tests assert |y_hip - y_oracle| <= len * eps * absum.  Not part of the
 * reference; a checker helper.
 */
void oracle_spmv_absum(double *absum, const double *a, const double *iv, const int *rowstr,
                       const int *colidx, int rows)
{
    for (int r = 0; r < rows; ++r) {
        double acc = 0.0;
        for (int k = rowstr[r] - 1; k < rowstr[r + 1] - 1; ++k) {
            double p = a[k] * iv[colidx[k] - 1];
            acc += p < 0 ? -p : p;
        }
        absum[r] = acc;
    }
}

void oracle_f_spmv_absum(double *absum, const float *a, const float *iv, const int *rowstr,
                         const int *colidx, int rows)
{
    for (int r = 0; r < rows; ++r) {
        double acc = 0.0;
        for (int k = rowstr[r] - 1; k < rowstr[r + 1] - 1; ++k) {
            double p = (double)a[k] * (double)iv[colidx[k] - 1];
            acc += p < 0 ? -p : p;
        }
        absum[r] = acc;
    }
}

/* FNV-1a 64-bit over raw bytes: the checksum SURVEY.md section 8(a) row a7
 * quotes for the NPB class S / A matrices. */
uint64_t oracle_fnv1a64(const void *data, size_t nbytes)
{
    const unsigned char *p = (const unsigned char *)data;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < nbytes; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}
