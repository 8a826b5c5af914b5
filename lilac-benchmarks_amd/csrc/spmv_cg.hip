/*
 * csrc/spmv_cg.hip -- opt-in, device-resident conjugate-gradient step.
 *
 * The drop-in ABI forces x and y through host memory on every call, and NPB's conj_grad
 * (/root/reference/NPB3.3.1/CG/cg.f:447-644) runs five host vector loops between consecutive
 * SpMVs; with the SpMV at ~80 us that round trip and those loops are most of the wall time.
 * These entry points keep p, q, r, z, x on the GPU for the whole of conj_grad and fuse the vector
 * work around the SpMV.  They are ADDITIONAL entry points: spmv_harness_ is untouched, and a caller
 * that does not know them loses nothing.
 *
 * One CG iteration (cg.f:513-620) = SpMV (q = A p) + three vector kernels, no host sync:
 *   k_dot      d_part[b]   = sum over block b of p*q   (folded into k_striped under the STRIPED plan)
 *   k_axpy     alpha = rho/d with rho = sum(rho_part_prev), d = sum(d_part) (every block adds the
 *              ~150 partials itself, in index order);  z += alpha p;  r -= alpha q;
 *              rho_part_new[b] = sum over block b of r*r
 *   k_pupd     beta = sum(rho_part_new)/sum(rho_part_prev);  p = r + beta p
 * All reductions are two-level with a fixed order (no atomics): results are reproducible run to
 * run.  They are NOT the Fortran loops' left-to-right sums, so zeta agrees with the reference to
 * ~1e-13 relative rather than bit for bit; NPB's own criterion is 1e-10 (cg.f:363-368).
 *
 * Round 4: the same step over N ROW PARTITIONS, one per GPU (spmv_hip_cg_create_multi; SURVEY.md 8(e) x 8(f)1).  Part k
 * holds its row block of A (x replicated: the SpMV's own partition), its row slices of x, z, q, r and a FULL copy of p.
 * No collective library; per CG iteration three exchanges, each closed by an event barrier between the parts' streams:
 *   1. q_k = A_k p;  d_k = p_k . q_k  -> one double per part in pinned host memory
 *   2. every part adds d_0 .. d_{N-1} and rho_0 .. rho_{N-1} in PART ORDER (the same bits everywhere), alpha = rho / d;
 *      z_k, r_k updated;  rho'_k = r_k . r_k -> pinned host memory
 *   3. beta from the rho' and rho slots, part order;  p_k = r_k + beta p_k written into the part's own copy of p, then
 *      pushed to the other N - 1 copies with hipMemcpyPeerAsync on the part's stream (class D over 8 GPUs: 1.5 MB to 7
 *      peers each, point to point over xGMI)
 * Results are a function of the partition (N and the row bounds), reproducible run to run; zeta agrees with the one-GPU
 * run to ~1e-13.  Per iteration and part: 9 launches, N - 1 copies, three exchanges -- ~75 us of API calls when ONE host
 * thread issues for everybody (profiles/r04_cg_parts_host_cost.log), i.e. more than the kernels of a class D row block at
 * N = 8; so when the parts lie on several devices every part gets a host thread of its own (cgm_thread_main; the threads
 * meet at a spinning barrier at every exchange).  SPMV_HIP_CG_THREADS = 0 / 1 forces either; the bits are the same.
 * UNMEASURED on more than one GPU: every test runs the parts wrapped onto one device, where the peer copy degenerates to a
 * device-to-device copy.
 */
#include <hip/hip_runtime.h>

#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "spmv_internal.h"

namespace {

constexpr int VB = 1024; /* elements per block of the vector kernels (256 threads x 4) */

__device__ __forceinline__ double block_sum(double v, double *s_w)
{
    /* wave tree via DPP-free shuffles (tiny kernels), then the 4 wave sums in order */
    for (int o = 32; o; o >>= 1)
        v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        s_w[w] = v;
    __syncthreads();
    return ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
}

__device__ __forceinline__ double sum_partials(const double *__restrict__ part, int nb)
{
    /* Every wave of every block needs the same scalar, bit for bit: lane l adds part[l], part[l+64],
     * ... in that order, then a butterfly over the wave -- the same tree in every lane (a + b and
     * b + a are the same double), so all lanes of all waves end with one value.  A plain loop over
     * the ~150 partials was a 150-deep dependent chain and cost 15 us per kernel. */
    const int lane = threadIdx.x & 63;
    double s = 0.0;
    for (int base = lane; base < nb; base += 64 * 16) { /* 16 loads in flight, added in index order */
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = base + 64 * k;
            v[k] = i < nb ? part[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            s += v[k]; /* + 0.0 past the end changes nothing (s is never -0.0 here... and -0.0 + 0.0 = 0.0 only
                          matters for a sum of no terms, which is 0.0 either way) */
    }
    for (int o = 32; o; o >>= 1)
        s += __shfl_xor(s, o, 64);
    return s;
}

/* r = p = x; z = q = 0; rho_part[b] = sum x*x     (cg.f:492-506) */
__global__ void __launch_bounds__(256) k_cg_init(const double *__restrict__ x, double *__restrict__ z,
                                                 double *__restrict__ p, double *__restrict__ q,
                                                 double *__restrict__ r, double *__restrict__ rho_part, int n)
{
    __shared__ double s_w[4];
    double acc = 0.0;
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * VB + k * 256 + threadIdx.x;
        if (i < n) {
            const double xi = x[i];
            z[i] = 0.0;
            q[i] = 0.0;
            r[i] = xi;
            p[i] = xi;
            acc += xi * xi;
        }
    }
    const double t = block_sum(acc, s_w);
    if (threadIdx.x == 0)
        rho_part[blockIdx.x] = t;
}

/* part[b] = sum a*b over block b; part2[b] = sum c*d (second pair optional) */
__global__ void __launch_bounds__(256) k_cg_dot(const double *__restrict__ a, const double *__restrict__ b,
                                                double *__restrict__ part, const double *__restrict__ c,
                                                const double *__restrict__ d, double *__restrict__ part2, int n)
{
    __shared__ double s_w[4];
    __shared__ double s_w2[4];
    double acc = 0.0, acc2 = 0.0;
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * VB + k * 256 + threadIdx.x;
        if (i < n) {
            acc += a[i] * b[i];
            if (part2)
                acc2 += c[i] * d[i];
        }
    }
    const double t = block_sum(acc, s_w);
    if (threadIdx.x == 0)
        part[blockIdx.x] = t;
    if (part2) {
        const double t2 = block_sum(acc2, s_w2);
        if (threadIdx.x == 0)
            part2[blockIdx.x] = t2;
    }
}

/* alpha = rho/d; z += alpha p; r -= alpha q; rho_new_part[b] = sum r*r   (cg.f:581-605) */
__global__ void __launch_bounds__(256) k_cg_axpy(double *__restrict__ z, double *__restrict__ r,
                                                 const double *__restrict__ p, const double *__restrict__ q,
                                                 const double *__restrict__ rho_part,
                                                 const double *__restrict__ d_part,
                                                 double *__restrict__ rho_new_part, int n, int nb, int nd)
{
    __shared__ double s_w[4];
    const double rho = sum_partials(rho_part, nb);
    const double d = sum_partials(d_part, nd);
    const double alpha = rho / d;
    double acc = 0.0;
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * VB + k * 256 + threadIdx.x;
        if (i < n) {
            z[i] = z[i] + alpha * p[i];
            const double ri = r[i] - alpha * q[i];
            r[i] = ri;
            acc += ri * ri;
        }
    }
    const double t = block_sum(acc, s_w);
    if (threadIdx.x == 0)
        rho_new_part[blockIdx.x] = t;
}

/* beta = rho_new/rho_old; p = r + beta p   (cg.f:610-617) */
__global__ void __launch_bounds__(256) k_cg_pupd(double *__restrict__ p, const double *__restrict__ r,
                                                 const double *__restrict__ rho_new_part,
                                                 const double *__restrict__ rho_old_part, int n, int nb)
{
    const double beta = sum_partials(rho_new_part, nb) / sum_partials(rho_old_part, nb);
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * VB + k * 256 + threadIdx.x;
        if (i < n)
            p[i] = r[i] + beta * p[i];
    }
}

/* part[b] = sum (x - r)^2   (cg.f:634-637) */
__global__ void __launch_bounds__(256) k_cg_resid(const double *__restrict__ x, const double *__restrict__ r,
                                                  double *__restrict__ part, int n)
{
    __shared__ double s_w[4];
    double acc = 0.0;
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * VB + k * 256 + threadIdx.x;
        if (i < n) {
            const double dd = x[i] - r[i];
            acc += dd * dd;
        }
    }
    const double t = block_sum(acc, s_w);
    if (threadIdx.x == 0)
        part[blockIdx.x] = t;
}

/* x = scale * z   (cg.f:343-345) ; fill */
__global__ void __launch_bounds__(256) k_cg_scale(double *__restrict__ x, const double *__restrict__ z,
                                                  double scale, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        x[i] = z ? scale * z[i] : scale;
}

/* ---- the row-partitioned step: scalars cross the parts through pinned host memory ---- */

/* *host_slot = the sum of this part's partials (sum_partials' order): one wave */
__global__ void __launch_bounds__(64) k_cgm_scalar(const double *__restrict__ part, int nb, double *host_slot)
{
    const double s = sum_partials(part, nb);
    if (threadIdx.x == 0)
        *host_slot = s;
}

/* out[0] = a[0] + ... + a[n-1], out[1] = b[0] + ... + b[n-1], in part order: one thread reads the pinned slots once, so that
 * the vector kernels behind it find the two scalars in device memory (every wave reading 2 n doubles across PCIe would cost
 * more than the kernels themselves) */
__global__ void __launch_bounds__(64) k_cgm_combine(const double *a, const double *b, int n, double *__restrict__ out)
{
    if (threadIdx.x != 0)
        return;
    double sa = 0.0, sb = 0.0;
    for (int k = 0; k < n; ++k) {
        sa += a[k];
        sb += b[k];
    }
    out[0] = sa;
    out[1] = sb;
}

/* alpha = sc[0] / sc[1]; z += alpha p; r -= alpha q; rho_new_part[b] = sum r*r over block b   (cg.f:581-605) */
__global__ void __launch_bounds__(256) k_cgm_axpy(double *__restrict__ z, double *__restrict__ r, const double *__restrict__ p,
                                                  const double *__restrict__ q, const double *__restrict__ sc,
                                                  double *__restrict__ rho_new_part, int n)
{
    __shared__ double s_w[4];
    const double alpha = sc[0] / sc[1];
    double acc = 0.0;
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * VB + k * 256 + threadIdx.x;
        if (i < n) {
            z[i] = z[i] + alpha * p[i];
            const double ri = r[i] - alpha * q[i];
            r[i] = ri;
            acc += ri * ri;
        }
    }
    const double t = block_sum(acc, s_w);
    if (threadIdx.x == 0)
        rho_new_part[blockIdx.x] = t;
}

/* beta = sc[0] / sc[1]; p = r + beta p   (cg.f:610-617) */
__global__ void __launch_bounds__(256) k_cgm_pupd(double *__restrict__ p, const double *__restrict__ r, const double *__restrict__ sc, int n)
{
    const double beta = sc[0] / sc[1];
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * VB + k * 256 + threadIdx.x;
        if (i < n)
            p[i] = r[i] + beta * p[i];
    }
}

} // namespace

/* one row partition of the multi-GPU step */
struct cg_part {
    spmv_hip_matrix *m = nullptr;
    int device = 0;
    int r0 = 0, rows = 0, nb = 0;
    double *pfull = nullptr;                                 /* n + 16: every part's copy of p (and of z for r = A z)   */
    double *x = nullptr, *z = nullptr, *q = nullptr, *r = nullptr; /* rows + 16: the part's row slices                  */
    double *part[4] = {nullptr, nullptr, nullptr, nullptr};  /* local partials: rho (two generations), d, scratch       */
    double *sc = nullptr;                                    /* 4 doubles: the combined scalars of the current step     */
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
};

struct spmv_hip_cg {
    spmv_hip_matrix *m = nullptr;
    int device = 0;
    int n = 0;  /* rows = cols */
    int nb = 0; /* partial sums per reduction */
    double *x = nullptr, *z = nullptr, *p = nullptr, *q = nullptr, *r = nullptr;
    double *part[4] = {nullptr, nullptr, nullptr, nullptr}; /* rho (two generations), d, scratch */
    double *h_part = nullptr;                               /* pinned, 2*nb */
    hipStream_t stream = nullptr;
    /* row-partitioned mode (spmv_hip_cg_create_multi): np > 0 */
    int np = 0;
    cg_part *parts = nullptr;
    double *h_slots = nullptr; /* pinned: [3][np] -- rho generation 0, rho generation 1, d: one double per part */
    hipEvent_t join = nullptr;
    /* a host thread per part (SPMV_HIP_CG_THREADS = 1 / 0; default: when the parts lie on more than one device) */
    int threads = 0;
    hipEvent_t *t_ev[2] = {nullptr, nullptr};
    std::atomic<int> t_fail{0}, t_go{0}, t_arrived{0}, t_gen{0};
};

namespace {

/* Every part's stream continues only after every part's stream has reached this point: N records, N - 1 waits on the hub
 * (part 0's stream), one record there, N - 1 waits on it.  Events are captured by the waits at the time of the call, so
 * the same event objects serve every barrier. */
int multi_barrier(spmv_hip_cg *c)
{
    cg_part *P = c->parts;
    for (int k = 0; k < c->np; ++k) {
        if (hipSetDevice(P[k].device) != hipSuccess || hipEventRecord(P[k].ev, P[k].stream) != hipSuccess)
            return -1;
    }
    if (hipSetDevice(P[0].device) != hipSuccess)
        return -1;
    for (int k = 1; k < c->np; ++k)
        if (hipStreamWaitEvent(P[0].stream, P[k].ev, 0) != hipSuccess)
            return -1;
    if (c->np > 1 && hipEventRecord(c->join, P[0].stream) != hipSuccess)
        return -1;
    for (int k = 1; k < c->np; ++k) {
        if (hipSetDevice(P[k].device) != hipSuccess || hipStreamWaitEvent(P[k].stream, c->join, 0) != hipSuccess)
            return -1;
    }
    return 0;
}

/* part k's row slice of `which` (0: the part's own copy of p, 1: z) into every OTHER part's full vector */
int multi_push_slice(spmv_hip_cg *c, int k, const double *src)
{
    cg_part *P = c->parts;
    if (P[k].rows == 0)
        return 0;
    if (hipSetDevice(P[k].device) != hipSuccess)
        return -1;
    for (int j = 0; j < c->np; ++j) {
        if (j == k)
            continue;
        if (hipMemcpyPeerAsync(P[j].pfull + P[k].r0, P[j].device, src, P[k].device, (size_t)P[k].rows * sizeof(double),
                               P[k].stream) != hipSuccess)
            return -1;
    }
    return 0;
}

void multi_free(spmv_hip_cg *c)
{
    for (int k = 0; k < c->np; ++k) {
        cg_part &p = c->parts[k];
        (void)hipSetDevice(p.device);
        if (p.stream)
            (void)hipStreamSynchronize(p.stream);
        double *bufs[] = {p.pfull, p.x, p.z, p.q, p.r, p.part[0], p.part[1], p.part[2], p.part[3], p.sc};
        for (double *b : bufs)
            if (b)
                (void)hipFree(b);
        if (p.ev)
            (void)hipEventDestroy(p.ev);
        if (p.stream)
            (void)hipStreamDestroy(p.stream);
    }
    delete[] c->parts;
    c->parts = nullptr;
    if (c->h_slots)
        (void)hipHostFree(c->h_slots);
    if (c->join)
        (void)hipEventDestroy(c->join);
    for (int set = 0; set < 2; ++set) {
        for (int k = 0; c->t_ev[set] && k < c->np; ++k)
            if (c->t_ev[set][k])
                (void)hipEventDestroy(c->t_ev[set][k]);
        delete[] c->t_ev[set];
        c->t_ev[set] = nullptr;
    }
}

#define CGM(expr)                                                                                       \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            spmv_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
            return -1;                                                                                  \
        }                                                                                               \
    } while (0)

/* ---- conj_grad over the row partitions, phase by phase (see the head of this file).  A phase is what ONE part enqueues
 * between two exchanges; the driver below runs the phases of all parts in turn from one thread, or every part's phases from
 * a thread of its own. */
struct cgm_ctx {
    spmv_hip_cg *c;
    double *h_rho[2], *h_d;
};

/* r = p = x, z = q = 0, rho (cg.f:492-506); the part's slice of p to every other copy */
int cgm_phase_init(const cgm_ctx &X, int k, int gen)
{
    cg_part &p = X.c->parts[k];
    CGM(hipSetDevice(p.device));
    if (p.rows > 0)
        hipLaunchKernelGGL(k_cg_init, dim3(p.nb), dim3(256), 0, p.stream, p.x, p.z, p.pfull + p.r0, p.q, p.r, p.part[0], p.rows);
    hipLaunchKernelGGL(k_cgm_scalar, dim3(1), dim3(64), 0, p.stream, p.part[0], p.nb, X.h_rho[gen] + k);
    CGM(hipGetLastError());
    return multi_push_slice(X.c, k, p.pfull + p.r0);
}

/* 1. q_k = A_k p and d_k = p_k . q_k (cg.f:532, 573-577) */
int cgm_phase_spmv(const cgm_ctx &X, int k)
{
    cg_part &p = X.c->parts[k];
    CGM(hipSetDevice(p.device));
    int nd = 0;
    if (p.rows > 0) {
        nd = spmvd_spmv_dot(p.m, p.pfull, p.q, p.stream, p.pfull + p.r0, p.part[2]);
        if (nd < 0)
            return -1;
        if (nd == 0) {
            if (spmv_hip_spmv_device(p.m, p.pfull, p.q, p.stream))
                return -1;
            CGM(hipSetDevice(p.device));
            hipLaunchKernelGGL(k_cg_dot, dim3(p.nb), dim3(256), 0, p.stream, p.pfull + p.r0, p.q, p.part[2], (const double *)nullptr,
                               (const double *)nullptr, (double *)nullptr, p.rows);
            nd = p.nb;
        }
    }
    hipLaunchKernelGGL(k_cgm_scalar, dim3(1), dim3(64), 0, p.stream, p.part[2], nd, X.h_d + k);
    CGM(hipGetLastError());
    return 0;
}

/* 2. alpha = rho / d; z, r; rho' (cg.f:581-605) */
int cgm_phase_axpy(const cgm_ctx &X, int k, int gen)
{
    cg_part &p = X.c->parts[k];
    CGM(hipSetDevice(p.device));
    hipLaunchKernelGGL(k_cgm_combine, dim3(1), dim3(64), 0, p.stream, X.h_rho[gen], X.h_d, X.c->np, p.sc);
    if (p.rows > 0)
        hipLaunchKernelGGL(k_cgm_axpy, dim3(p.nb), dim3(256), 0, p.stream, p.z, p.r, p.pfull + p.r0, p.q, p.sc, p.part[1], p.rows);
    hipLaunchKernelGGL(k_cgm_scalar, dim3(1), dim3(64), 0, p.stream, p.part[1], p.nb, X.h_rho[gen ^ 1] + k);
    CGM(hipGetLastError());
    return 0;
}

/* 3. beta = rho' / rho; p = r + beta p, into every copy of p (cg.f:610-617) */
int cgm_phase_pupd(const cgm_ctx &X, int k, int gen)
{
    cg_part &p = X.c->parts[k];
    CGM(hipSetDevice(p.device));
    hipLaunchKernelGGL(k_cgm_combine, dim3(1), dim3(64), 0, p.stream, X.h_rho[gen ^ 1], X.h_rho[gen], X.c->np, p.sc + 2);
    if (p.rows > 0)
        hipLaunchKernelGGL(k_cgm_pupd, dim3(p.nb), dim3(256), 0, p.stream, p.pfull + p.r0, p.r, p.sc + 2, p.rows);
    CGM(hipGetLastError());
    return multi_push_slice(X.c, k, p.pfull + p.r0);
}

/* r = A z (cg.f:628), first half: z gathered into every part's full vector (p is dead until the next call's init) */
int cgm_phase_gather_z(const cgm_ctx &X, int k)
{
    cg_part &p = X.c->parts[k];
    CGM(hipSetDevice(p.device));
    if (p.rows > 0)
        CGM(hipMemcpyAsync(p.pfull + p.r0, p.z, (size_t)p.rows * sizeof(double), hipMemcpyDeviceToDevice, p.stream));
    return multi_push_slice(X.c, k, p.z);
}

/* ... second half: the product, the partials of ||x - r||^2 to the host; the part's stream is drained */
int cgm_phase_resid(const cgm_ctx &X, int k)
{
    cg_part &p = X.c->parts[k];
    CGM(hipSetDevice(p.device));
    if (p.rows > 0) {
        if (spmv_hip_spmv_device(p.m, p.pfull, p.r, p.stream))
            return -1;
        CGM(hipSetDevice(p.device));
        hipLaunchKernelGGL(k_cg_resid, dim3(p.nb), dim3(256), 0, p.stream, p.x, p.r, p.part[3], p.rows);
        CGM(hipGetLastError());
        CGM(hipMemcpyAsync(X.c->h_part + p.r0 / VB + k, p.part[3], (size_t)p.nb * sizeof(double), hipMemcpyDeviceToHost, p.stream));
    }
    CGM(hipStreamSynchronize(p.stream));
    return 0;
}

/* ---- a thread per part: every part's launches, copies and waits are issued by its own host thread (one thread issuing for
 * eight GPUs is ~200 API calls per CG iteration: more than the kernels take).  The exchange points are the same; at each
 * the thread records its event, meets the others at a host barrier -- so that every record has happened --, then lets its
 * stream wait for the other parts' events.  Two sets of events alternate: a set is recorded again only two exchanges later,
 * when every thread is long past the waits that captured its previous record. */
struct cgm_thread {
    cgm_ctx X;
    int k, cgitmax, rc;
    pthread_t th;
};

int cgm_thread_exchange(const cgm_ctx &X, int k, int &set, int rc_so_far)
{
    spmv_hip_cg *c = X.c;
    cg_part &p = c->parts[k];
    if (rc_so_far)
        c->t_fail.store(1, std::memory_order_relaxed);
    if (hipSetDevice(p.device) != hipSuccess || hipEventRecord(c->t_ev[set][k], p.stream) != hipSuccess)
        c->t_fail.store(1, std::memory_order_relaxed);
    /* every thread comes here, failed or not: nobody is left waiting.  A spinning barrier (generation counter): the threads meet
     * three times per CG iteration, a futex sleep and wake-up each time would cost more than the launches between them */
    {
        const int gen0 = c->t_gen.load(std::memory_order_acquire);
        if (c->t_arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->np) {
            c->t_arrived.store(0, std::memory_order_relaxed);
            c->t_gen.store(gen0 + 1, std::memory_order_release);
        } else {
            for (unsigned spins = 0; c->t_gen.load(std::memory_order_acquire) == gen0; ++spins)
                if (spins > 20000)
                    sched_yield(); /* more threads than cores: let the others run */
        }
    }
    if (c->t_fail.load(std::memory_order_relaxed))
        return -1;
    for (int j = 0; j < c->np; ++j)
        if (j != k && hipStreamWaitEvent(p.stream, c->t_ev[set][j], 0) != hipSuccess) {
            spmv_set_error("hipStreamWaitEvent failed between row partitions %d and %d", k, j);
            c->t_fail.store(1, std::memory_order_relaxed); /* seen by everybody at the next exchange */
        }
    set ^= 1;
    return 0;
}

void *cgm_thread_main(void *arg)
{
    cgm_thread *T = (cgm_thread *)arg;
    const cgm_ctx &X = T->X;
    const int k = T->k;
    int set = 0, gen = 0;
    /* nobody enqueues anything before every thread exists (a thread that could not be created would leave the others at the
     * first exchange for ever): t_go 1 = all there, -1 = give up */
    for (int go = 0; (go = X.c->t_go.load(std::memory_order_acquire)) != 1;) {
        if (go < 0) {
            T->rc = -2;
            return nullptr;
        }
        sched_yield();
    }
    /* the number of exchanges is the same in every thread whatever happens: a failed phase only raises the flag */
#define CGM_STEP(call)                                  \
    if (cgm_thread_exchange(X, k, set, (call)) != 0) {  \
        T->rc = -1;                                     \
        return nullptr;                                 \
    }
    CGM_STEP(cgm_phase_init(X, k, gen));
    for (int it = 0; it < T->cgitmax; ++it) {
        CGM_STEP(cgm_phase_spmv(X, k));
        CGM_STEP(cgm_phase_axpy(X, k, gen));
        CGM_STEP(cgm_phase_pupd(X, k, gen));
        gen ^= 1;
    }
    CGM_STEP(cgm_phase_gather_z(X, k));
    CGM_STEP(cgm_phase_resid(X, k)); /* drains the stream; the exchange behind it: nobody starts the next call's init early */
#undef CGM_STEP
    T->rc = 0;
    return nullptr;
}

int multi_conj_grad(spmv_hip_cg *c, int cgitmax, double *rnorm)
{
    cg_part *P = c->parts;
    const int np = c->np;
    cgm_ctx X;
    X.c = c;
    X.h_rho[0] = c->h_slots, X.h_rho[1] = c->h_slots + np, X.h_d = c->h_slots + 2 * np;
    if (c->threads && np > 1) {
        cgm_thread *T = new cgm_thread[np];
        c->t_fail.store(0, std::memory_order_relaxed);
        c->t_arrived.store(0, std::memory_order_relaxed);
        c->t_go.store(0, std::memory_order_release);
        int started = 1; /* part 0 runs on the calling thread */
        for (int k = 0; k < np; ++k)
            T[k].X = X, T[k].k = k, T[k].cgitmax = cgitmax, T[k].rc = -1;
        for (int k = 1; k < np; ++k) {
            if (pthread_create(&T[k].th, nullptr, cgm_thread_main, &T[k]) != 0)
                break;
            ++started;
        }
        c->t_go.store(started == np ? 1 : -1, std::memory_order_release);
        if (started == np)
            cgm_thread_main(&T[0]);
        for (int k = 1; k < started; ++k)
            pthread_join(T[k].th, nullptr);
        int rc = started == np ? 0 : -1;
        if (started != np)
            spmv_set_error("spmv_hip_cg: could not start the %d host threads of the row partitions (SPMV_HIP_CG_THREADS=0 runs them from one)", np - 1);
        for (int k = 0; k < np && started == np; ++k)
            rc |= T[k].rc;
        delete[] T;
        if (rc)
            return -1;
    } else {
        int gen = 0; /* h_rho[gen]: rho of the current residual */
#define CGM_ALL(call)               \
    for (int k = 0; k < np; ++k)    \
        if ((call) != 0)            \
            return -1;              \
    if (multi_barrier(c))           \
        return -1;
        CGM_ALL(cgm_phase_init(X, k, gen));
        for (int it = 0; it < cgitmax; ++it) {
            CGM_ALL(cgm_phase_spmv(X, k));
            CGM_ALL(cgm_phase_axpy(X, k, gen));
            CGM_ALL(cgm_phase_pupd(X, k, gen));
            gen ^= 1;
        }
        CGM_ALL(cgm_phase_gather_z(X, k));
        CGM_ALL(cgm_phase_resid(X, k));
#undef CGM_ALL
    }
    double s = 0.0;
    for (int k = 0; k < np; ++k) {
        const double *hp = c->h_part + P[k].r0 / VB + k; /* part k's partials: at most rows_k / VB + 1 of them */
        for (int i = 0; i < (P[k].rows > 0 ? P[k].nb : 0); ++i)
            s += hp[i];
    }
    if (rnorm)
        *rnorm = sqrt(s);
    return 0;
}

} // namespace

#define CGCHK(expr)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            spmv_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                           __LINE__);                                                        \
            return -1;                                                                       \
        }                                                                                    \
    } while (0)

extern "C" {

void spmv_hip_cg_destroy(spmv_hip_cg *c)
{
    if (!c)
        return;
    if (c->np > 0) {
        multi_free(c);
        if (c->h_part)
            (void)hipHostFree(c->h_part);
        delete c;
        return;
    }
    (void)hipSetDevice(c->device);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    double *bufs[] = {c->x, c->z, c->p, c->q, c->r, c->part[0], c->part[1], c->part[2], c->part[3]};
    for (double *b : bufs)
        if (b)
            (void)hipFree(b);
    if (c->h_part)
        (void)hipHostFree(c->h_part);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
}

spmv_hip_cg *spmv_hip_cg_create(spmv_hip_matrix *m)
{
    spmv_hip_info inf;
    if (!m || spmv_hip_matrix_info(m, &inf))
        return nullptr;
    if (inf.elem_size != 8 || inf.rows < inf.cols) {
        spmv_set_error("spmv_hip_cg_create needs a square FP64 matrix (rows=%d cols=%d elem=%d)", inf.rows,
                       inf.cols, inf.elem_size);
        return nullptr;
    }
    if (hipSetDevice(inf.device) != hipSuccess)
        return nullptr;
    spmv_hip_cg *c = new spmv_hip_cg();
    c->m = m;
    c->device = inf.device;
    c->n = inf.rows;
    c->nb = (c->n + VB - 1) / VB;
    const size_t vb = ((size_t)c->n + 16) * sizeof(double);
    double **vecs[] = {&c->x, &c->z, &c->p, &c->q, &c->r};
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    for (double **v : vecs)
        ok = ok && hipMalloc((void **)v, vb) == hipSuccess && hipMemset(*v, 0, vb) == hipSuccess;
    for (int k = 0; k < 4; ++k)
        ok = ok && hipMalloc((void **)&c->part[k], ((size_t)c->n + 64) * sizeof(double)) ==
                       hipSuccess; /* room for one partial per wave range of the STRIPED plan (the folded dot: at most
                                      rows + 16 of them), nb <= that */
    ok = ok && hipHostMalloc((void **)&c->h_part, (size_t)(2 * c->nb + 2) * sizeof(double), hipHostMallocDefault) ==
                   hipSuccess;
    if (!ok) {
        spmv_set_error("spmv_hip_cg_create: device allocation failed");
        spmv_hip_cg_destroy(c);
        return nullptr;
    }
    return c;
}

/*
 * The same step over `nparts` row partitions of one square FP64 matrix: parts[k] holds rows [b_k, b_k+1) (in order, b_0 = 0,
 * b_nparts = cols; spmv_hip_partition_rows makes such bounds) with the full column range (cols_hint = the matrix order), each
 * on the device it was created on -- several parts may share a device (a rehearsal of the N-GPU path on fewer GPUs).
 * Every other spmv_hip_cg_* entry point takes the handle as it takes a one-GPU one.
 */
spmv_hip_cg *spmv_hip_cg_create_multi(spmv_hip_matrix **parts, int nparts)
{
    if (!parts || nparts < 1 || nparts > 64) {
        spmv_set_error("spmv_hip_cg_create_multi: 1 to 64 row partitions, got %d", nparts);
        return nullptr;
    }
    spmv_hip_cg *c = new spmv_hip_cg();
    c->np = nparts;
    c->parts = new cg_part[nparts];
    int r0 = 0, n = -1;
    bool ok = true;
    for (int k = 0; k < nparts && ok; ++k) {
        spmv_hip_info inf;
        if (!parts[k] || spmv_hip_matrix_info(parts[k], &inf)) {
            spmv_set_error("spmv_hip_cg_create_multi: part %d is not a matrix", k);
            ok = false;
            break;
        }
        if (n < 0)
            n = inf.cols;
        if (inf.elem_size != 8 || inf.cols != n) {
            spmv_set_error("spmv_hip_cg_create_multi: part %d: FP64 row blocks of one matrix wanted (elem %d, cols %d against %d)", k,
                           inf.elem_size, inf.cols, n);
            ok = false;
            break;
        }
        cg_part &p = c->parts[k];
        p.m = parts[k], p.device = inf.device, p.r0 = r0, p.rows = inf.rows, p.nb = (inf.rows + VB - 1) / VB;
        r0 += inf.rows;
    }
    if (ok && r0 != n) {
        spmv_set_error("spmv_hip_cg_create_multi: the parts hold %d rows of a matrix with %d columns (a square matrix, every row once)", r0, n);
        ok = false;
    }
    c->n = n;
    c->nb = n > 0 ? (n + VB - 1) / VB : 0;
    for (int k = 0; k < nparts && ok; ++k) {
        cg_part &p = c->parts[k];
        ok = hipSetDevice(p.device) == hipSuccess;
        for (int j = 0; j < nparts && ok; ++j) /* point-to-point copies between the parts' devices (xGMI); already on: fine */
            if (c->parts[j].device != p.device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, p.device, c->parts[j].device) == hipSuccess && can) {
                    (void)hipDeviceEnablePeerAccess(c->parts[j].device, 0);
                    (void)hipGetLastError();
                }
            }
        ok = ok && hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&p.ev, hipEventDisableTiming) == hipSuccess;
        const size_t vfull = ((size_t)n + 16) * sizeof(double), vloc = ((size_t)p.rows + 16) * sizeof(double);
        ok = ok && hipMalloc((void **)&p.pfull, vfull) == hipSuccess && hipMemset(p.pfull, 0, vfull) == hipSuccess;
        double **vecs[] = {&p.x, &p.z, &p.q, &p.r};
        for (double **v : vecs)
            ok = ok && hipMalloc((void **)v, vloc) == hipSuccess && hipMemset(*v, 0, vloc) == hipSuccess;
        for (int j = 0; j < 4; ++j) /* as in the one-GPU step: room for one partial per wave range of the STRIPED plan */
            ok = ok && hipMalloc((void **)&p.part[j], ((size_t)p.rows + 64) * sizeof(double)) == hipSuccess;
        ok = ok && hipMalloc((void **)&p.sc, 4 * sizeof(double)) == hipSuccess;
    }
    if (ok) {
        ok = hipSetDevice(c->parts[0].device) == hipSuccess && hipEventCreateWithFlags(&c->join, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&c->h_slots, (size_t)(3 * nparts + 2) * sizeof(double),
                                 hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess; /* every GPU reads and writes it */
        /* partials of the residual and of the two norms: part k's at r0_k / VB + k (at most rows_k / VB + 1 each) */
        ok = ok && hipHostMalloc((void **)&c->h_part, (size_t)(2 * (c->nb + nparts) + 2) * sizeof(double), hipHostMallocDefault) == hipSuccess;
        if (!ok)
            spmv_set_error("spmv_hip_cg_create_multi: allocation failed: %s", hipGetErrorString(hipGetLastError()));
    }
    if (ok) { /* who issues the work: one thread for all parts, or a thread per part (their events: two sets, see cgm_thread_exchange) */
        const char *e = getenv("SPMV_HIP_CG_THREADS");
        bool several_devices = false; /* threads pay when the parts' devices work side by side; parts wrapped onto one GPU share it anyway */
        for (int k = 1; k < nparts; ++k)
            several_devices = several_devices || c->parts[k].device != c->parts[0].device;
        c->threads = e && *e ? atoi(e) != 0 : several_devices;
        if (c->threads && nparts > 1) {
            for (int set = 0; set < 2 && ok; ++set) {
                c->t_ev[set] = new hipEvent_t[nparts]();
                for (int k = 0; k < nparts && ok; ++k)
                    ok = hipSetDevice(c->parts[k].device) == hipSuccess &&
                         hipEventCreateWithFlags(&c->t_ev[set][k], hipEventDisableTiming) == hipSuccess;
            }
            if (!ok)
                spmv_set_error("spmv_hip_cg_create_multi: could not set up the per-part threads' events");
        }
    }
    if (!ok) {
        spmv_hip_cg_destroy(c);
        return nullptr;
    }
    return c;
}

int spmv_hip_cg_set_x(spmv_hip_cg *c, const double *x_host)
{
    if (c->np > 0) {
        for (int k = 0; k < c->np; ++k) {
            cg_part &p = c->parts[k];
            CGCHK(hipSetDevice(p.device));
            if (p.rows > 0)
                CGCHK(hipMemcpyAsync(p.x, x_host + p.r0, (size_t)p.rows * sizeof(double), hipMemcpyHostToDevice, p.stream));
        }
        for (int k = 0; k < c->np; ++k) {
            CGCHK(hipSetDevice(c->parts[k].device));
            CGCHK(hipStreamSynchronize(c->parts[k].stream));
        }
        return 0;
    }
    CGCHK(hipSetDevice(c->device));
    CGCHK(hipMemcpyAsync(c->x, x_host, (size_t)c->n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    CGCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int spmv_hip_cg_fill_x(spmv_hip_cg *c, double value)
{
    if (c->np > 0) {
        for (int k = 0; k < c->np; ++k) {
            cg_part &p = c->parts[k];
            CGCHK(hipSetDevice(p.device));
            if (p.rows > 0)
                hipLaunchKernelGGL(k_cg_scale, dim3((p.rows + 255) / 256), dim3(256), 0, p.stream, p.x, (const double *)nullptr, value, p.rows);
            CGCHK(hipGetLastError());
        }
        return 0;
    }
    CGCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_cg_scale, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->x, (const double *)nullptr,
                       value, c->n);
    CGCHK(hipGetLastError());
    return 0;
}

/* which: 0 x, 1 z, 2 p, 3 q, 4 r */
int spmv_hip_cg_get(spmv_hip_cg *c, int which, double *host)
{
    if (which < 0 || which > 4)
        return -1;
    if (c->np > 0) {
        for (int k = 0; k < c->np; ++k) {
            cg_part &p = c->parts[k];
            const double *v[] = {p.x, p.z, p.pfull + p.r0, p.q, p.r};
            CGCHK(hipSetDevice(p.device));
            if (p.rows > 0)
                CGCHK(hipMemcpyAsync(host + p.r0, v[which], (size_t)p.rows * sizeof(double), hipMemcpyDeviceToHost, p.stream));
        }
        for (int k = 0; k < c->np; ++k) {
            CGCHK(hipSetDevice(c->parts[k].device));
            CGCHK(hipStreamSynchronize(c->parts[k].stream));
        }
        return 0;
    }
    double *v[] = {c->x, c->z, c->p, c->q, c->r};
    CGCHK(hipSetDevice(c->device));
    CGCHK(hipMemcpyAsync(host, v[which], (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CGCHK(hipStreamSynchronize(c->stream));
    return 0;
}

/* conj_grad of cg.f:447-644: z ~= A^-1 x by cgitmax CG iterations, *rnorm = ||x - A z||. */
int spmv_hip_cg_conj_grad(spmv_hip_cg *c, int cgitmax, double *rnorm)
{
    if (c->np > 0)
        return multi_conj_grad(c, cgitmax, rnorm);
    CGCHK(hipSetDevice(c->device));
    const int n = c->n, nb = c->nb;
    hipStream_t st = c->stream;
    double *rho_prev = c->part[0], *rho_new = c->part[1], *dpart = c->part[2];
    hipLaunchKernelGGL(k_cg_init, dim3(nb), dim3(256), 0, st, c->x, c->z, c->p, c->q, c->r, rho_prev, n);
    for (int it = 0; it < cgitmax; ++it) {
        /* q = A p (cg.f:532) and d = p.q (cg.f:573-577): the STRIPED plan leaves the block partials of
         * the dot behind in the kernel that writes q; other plans get a separate pass over q */
        int nd = spmvd_spmv_dot(c->m, c->p, c->q, st, c->p, dpart);
        if (nd < 0)
            return -1;
        if (nd == 0) {
            if (spmv_hip_spmv_device(c->m, c->p, c->q, st))
                return -1;
            hipLaunchKernelGGL(k_cg_dot, dim3(nb), dim3(256), 0, st, c->p, c->q, dpart, (const double *)nullptr,
                               (const double *)nullptr, (double *)nullptr, n);
            nd = nb;
        }
        hipLaunchKernelGGL(k_cg_axpy, dim3(nb), dim3(256), 0, st, c->z, c->r, c->p, c->q, rho_prev, dpart, rho_new,
                           n, nb, nd);
        hipLaunchKernelGGL(k_cg_pupd, dim3(nb), dim3(256), 0, st, c->p, c->r, rho_new, rho_prev, n, nb);
        double *t = rho_prev;
        rho_prev = rho_new;
        rho_new = t;
    }
    if (spmv_hip_spmv_device(c->m, c->z, c->r, st)) /* r = A z, cg.f:628 */
        return -1;
    hipLaunchKernelGGL(k_cg_resid, dim3(nb), dim3(256), 0, st, c->x, c->r, c->part[3], n);
    CGCHK(hipGetLastError());
    CGCHK(hipMemcpyAsync(c->h_part, c->part[3], (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, st));
    CGCHK(hipStreamSynchronize(st));
    double s = 0.0;
    for (int i = 0; i < nb; ++i)
        s += c->h_part[i];
    if (rnorm)
        *rnorm = sqrt(s);
    return 0;
}

/* norm_temp1 = x.z, norm_temp2 = z.z   (cg.f:323-327) */
int spmv_hip_cg_norms(spmv_hip_cg *c, double *x_dot_z, double *z_dot_z)
{
    if (c->np > 0) { /* per part the block partials of x.z and z.z; the host adds them part by part, in index order */
        double *h2 = c->h_part + c->nb + c->np + 1;
        for (int k = 0; k < c->np; ++k) {
            cg_part &p = c->parts[k];
            if (p.rows == 0)
                continue;
            CGCHK(hipSetDevice(p.device));
            hipLaunchKernelGGL(k_cg_dot, dim3(p.nb), dim3(256), 0, p.stream, p.x, p.z, p.part[2], p.z, p.z, p.part[3], p.rows);
            CGCHK(hipGetLastError());
            CGCHK(hipMemcpyAsync(c->h_part + p.r0 / VB + k, p.part[2], (size_t)p.nb * sizeof(double), hipMemcpyDeviceToHost, p.stream));
            CGCHK(hipMemcpyAsync(h2 + p.r0 / VB + k, p.part[3], (size_t)p.nb * sizeof(double), hipMemcpyDeviceToHost, p.stream));
        }
        double a = 0.0, b = 0.0;
        for (int k = 0; k < c->np; ++k) {
            cg_part &p = c->parts[k];
            if (p.rows == 0)
                continue;
            CGCHK(hipSetDevice(p.device));
            CGCHK(hipStreamSynchronize(p.stream));
            for (int i = 0; i < p.nb; ++i) {
                a += c->h_part[p.r0 / VB + k + i];
                b += h2[p.r0 / VB + k + i];
            }
        }
        *x_dot_z = a;
        *z_dot_z = b;
        return 0;
    }
    CGCHK(hipSetDevice(c->device));
    const int nb = c->nb;
    hipLaunchKernelGGL(k_cg_dot, dim3(nb), dim3(256), 0, c->stream, c->x, c->z, c->part[2], c->z, c->z, c->part[3],
                       c->n);
    CGCHK(hipGetLastError());
    CGCHK(hipMemcpyAsync(c->h_part, c->part[2], (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CGCHK(hipMemcpyAsync(c->h_part + nb, c->part[3], (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CGCHK(hipStreamSynchronize(c->stream));
    double a = 0.0, b = 0.0;
    for (int i = 0; i < nb; ++i) {
        a += c->h_part[i];
        b += c->h_part[nb + i];
    }
    *x_dot_z = a;
    *z_dot_z = b;
    return 0;
}

/* x = scale * z   (cg.f:343-345) */
int spmv_hip_cg_x_from_z(spmv_hip_cg *c, double scale)
{
    if (c->np > 0) {
        for (int k = 0; k < c->np; ++k) {
            cg_part &p = c->parts[k];
            CGCHK(hipSetDevice(p.device));
            if (p.rows > 0)
                hipLaunchKernelGGL(k_cg_scale, dim3((p.rows + 255) / 256), dim3(256), 0, p.stream, p.x, (const double *)p.z, scale, p.rows);
            CGCHK(hipGetLastError());
        }
        return 0;
    }
    CGCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_cg_scale, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->x, (const double *)c->z,
                       scale, c->n);
    CGCHK(hipGetLastError());
    return 0;
}

} // extern "C"
