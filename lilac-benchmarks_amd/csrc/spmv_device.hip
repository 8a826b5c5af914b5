/*
 * csrc/spmv_device.hip -- the thin HIP layer under the C host library.
 *
 * Owns everything that touches the HIP runtime: per-device streams, the
 * device-resident CSR block (spmv_hip_matrix), plan materialisation and kernel
 * dispatch, event timing, and the small copy helpers the C side
 * (spmv_harness.c) and the ctypes/C drivers use.  It takes over what the
 * reference GPU backend does in /root/reference/libspmv/gpu.c:36-86 (setup,
 * allocation), :217-223 (column-count inference), :236-240 (upload) and
 * :270-282 (the csrmv call).
 *
 * Layout in HBM, per matrix block (rows r0..r1 of the caller's CSR):
 *   rowptr  int32[rows+1]            0-based, local to the block
 *   colidx  int32[nnz + PAD]         1-based column ids as the ABI delivers them; pad = 1
 *   val     T[nnz + PAD]             pad = 0
 *   plan    int32 row-block / segment / bin lists (O(rows / 8) ints)
 * x and y are whatever device buffers the caller passes.  Everything is
 * allocated once per cached matrix; nothing is allocated per SpMV.
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>

#include "spmv_internal.h"

extern "C" void spmvk_set_tuning(int key, int value);
extern "C" void spmvk_blocked_set_bin_rows(int per_xcd);
extern "C" void spmvk_hub_set_min_cover(int per_cent);
extern "C" void spmvk_hub_set_table_bytes(int bytes);
extern "C" void spmvk_hub_set_lds_cap(int columns);
#ifdef SPMV_HIP_DEBUG_BUILD
extern "C" void spmvk_blocked_set_abl(int v);
#endif
extern "C" int spmvk_get_tuning_xcd(void);
static bool g_tune_xcd_enabled() { return spmvk_get_tuning_xcd() != 0; }

namespace {

thread_local char g_err[512] = "";

#define SPMV_MAX_DEVICES 16
#define SPMV_Y_CHUNKS_MAX 16
struct DevCtx {
    bool ready = false;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;     /* y chunks on their way to the host while later chunks are computed */
    hipEvent_t ev[2] = {nullptr, nullptr}; /* spmv_hip_time_device: created once, not inside the timed region */
    hipEvent_t chunk_ev[SPMV_Y_CHUNKS_MAX] = {};
};
DevCtx g_dev[SPMV_MAX_DEVICES];
std::mutex g_dev_mu;

#define HIPCHK(expr, onfail)                                                                   \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            spmv_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,     \
                           __LINE__);                                                          \
            onfail;                                                                            \
        }                                                                                      \
    } while (0)

int use_device(int device)
{
    if (device < 0 || device >= SPMV_MAX_DEVICES) {
        spmv_set_error("device ordinal %d out of range", device);
        return -1;
    }
    HIPCHK(hipSetDevice(device), return -1);
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (!g_dev[device].ready) {
        HIPCHK(hipStreamCreateWithFlags(&g_dev[device].stream, hipStreamNonBlocking), return -1);
        g_dev[device].ready = true;
    }
    return 0;
}

hipStream_t stream_of(int device, void *stream)
{
    return stream ? (hipStream_t)stream : g_dev[device].stream;
}

} // namespace

struct spmv_hip_matrix {
    int device = 0;
    int elem = 8;
    int rows = 0;
    int cols = 0;
    int64_t nnz = 0;
    int64_t first = 0; /* 0-based offset of the block's first entry in the caller's a / colidx */
    int *d_rowptr = nullptr;
    int *d_col = nullptr;
    void *d_val = nullptr;
    size_t device_bytes = 0;

    spmv_row_stats st{};
    int plan = SPMV_HIP_PLAN_STREAM;
    int param = SPMV_STREAM_CAP;

    /* STREAM + long rows */
    spmv_stream_plan sp{};
    int *d_blk = nullptr;
    int *d_seg = nullptr;
    int *d_lrow = nullptr;
    void *d_partials = nullptr;
    /* BINNED (built on demand; needs the host rowstr only at creation, so kept as a copy of lens) */
    spmv_bin_plan bp{};
    bool have_bins = false;
    int *d_binrows = nullptr;
    int *h_rowstr_local = nullptr; /* rows+1 ints, 1-based copy of the block's rowstr (for replans) */
    /* TILED (built on demand from the device CSR copy) */
    bool have_tiled = false;
    int tiled_rc = 0; /* last build status: 0 ok, -2 not representable, -3 no memory (host or device) for the second
                         copy -- neither is fatal for an automatic choice: the CSR copy serves --, -1 anything else */
    spmv_dev_tiled dt{};
    size_t tiled_bytes = 0;
    /* STRIPED (built on demand) */
    bool have_striped = false;
    bool striped_mostly_filler = false;
    size_t striped_bytes = 0;
    /* BLOCKED (built on demand, on the device) */
    bool have_blocked = false;
    int blocked_rc = 0;
    spmv_dev_blocked db{};
    int striped_rc = 0;
    spmv_dev_striped ds{};
    /* HUB (built on demand, on the device; holds no values of its own: a value refresh leaves it alone) */
    bool have_hub = false;
    int hub_rc = 0;
    spmv_dev_hub dh{};
};

extern "C" {

void spmv_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

const char *spmv_hip_last_error(void) { return g_err; }

const char *spmv_hip_version(void) { return "hip-spmv 0.1 (gfx950)"; }

double spmv_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int spmv_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

/* bytes of device memory free on `device` right now (hipMemGetInfo), -1 on error */
int64_t spmv_hip_device_free_bytes(int device)
{
    size_t free_b = 0, total_b = 0;
    if (use_device(device))
        return -1;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return (int64_t)free_b;
}

/* ---- memory helpers ------------------------------------------------------------- */
void *spmv_hip_malloc(int device, size_t bytes)
{
    if (use_device(device))
        return nullptr;
    void *p = nullptr;
    HIPCHK(hipMalloc(&p, bytes ? bytes : 16), return nullptr);
    return p;
}

void spmv_hip_free(int device, void *p)
{
    if (!p)
        return;
    if (hipSetDevice(device) == hipSuccess)
        (void)hipFree(p);
}

int spmv_hip_memcpy_h2d(int device, void *dst_dev, const void *src_host, size_t bytes)
{
    if (use_device(device))
        return -1;
    /* ordered on the library's stream, then drained: the stream is non-blocking,
     * so a NULL-stream hipMemcpy would not be ordered against our kernels (the
     * hazard SURVEY.md 5.2(a) points out in libspmv/gpu.c:41,264,285) */
    HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, g_dev[device].stream),
           return -1);
    HIPCHK(hipStreamSynchronize(g_dev[device].stream), return -1);
    return 0;
}

int spmv_hip_memcpy_d2h(int device, void *dst_host, const void *src_dev, size_t bytes)
{
    if (use_device(device))
        return -1;
    HIPCHK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, g_dev[device].stream),
           return -1);
    HIPCHK(hipStreamSynchronize(g_dev[device].stream), return -1);
    return 0;
}

int spmv_hip_memset(int device, void *dst_dev, int value, size_t bytes)
{
    if (use_device(device))
        return -1;
    HIPCHK(hipMemsetAsync(dst_dev, value, bytes, g_dev[device].stream), return -1);
    HIPCHK(hipStreamSynchronize(g_dev[device].stream), return -1);
    return 0;
}

int spmv_hip_synchronize(int device)
{
    if (use_device(device))
        return -1;
    HIPCHK(hipStreamSynchronize(g_dev[device].stream), return -1);
    HIPCHK(hipDeviceSynchronize(), return -1);
    return 0;
}

/* ---- internal helpers for the C host side (spmv_harness.c) ------------------------ */
void *spmvd_host_alloc(size_t bytes)
{
    void *p = nullptr;
    /* portable: pinned for every GPU of the process (SPMV_HIP_DEVICES > 1 copies x from the same buffer to all
     * of them); mapped: the zero-copy path of the harness reads / writes it from kernels */
    HIPCHK(hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocPortable | hipHostMallocMapped), return nullptr);
    return p;
}

void spmvd_host_free(void *p)
{
    if (p)
        (void)hipHostFree(p);
}

int spmvd_host_register(void *p, size_t bytes)
{
    HIPCHK(hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped), return -1);
    return 0;
}

void spmvd_host_unregister(void *p) { (void)hipHostUnregister(p); }

int spmvd_copy_h2d_async(int device, void *dst_dev, const void *src, size_t bytes)
{
    if (use_device(device))
        return -1;
    HIPCHK(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, g_dev[device].stream), return -1);
    return 0;
}

int spmvd_copy_d2h_async(int device, void *dst, const void *src_dev, size_t bytes)
{
    if (use_device(device))
        return -1;
    HIPCHK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, g_dev[device].stream), return -1);
    return 0;
}

int spmvd_stream_sync(int device)
{
    if (use_device(device))
        return -1;
    HIPCHK(hipStreamSynchronize(g_dev[device].stream), return -1);
    return 0;
}

void *spmvd_host_devptr(void *pinned_host)
{
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, pinned_host, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return d;
}

int spmvd_copy_words_async(int device, void *dst_dev, const void *src_devptr, size_t bytes)
{
    if (use_device(device))
        return -1;
    int rc = spmvk_copy_words(dst_dev, src_devptr, bytes, g_dev[device].stream);
    if (rc) {
        spmv_set_error("copy kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        return -1;
    }
    return 0;
}

/* ---- matrices ------------------------------------------------------------------- */

static void free_relayouts(spmv_hip_matrix *m);

static void free_hub(spmv_hip_matrix *m)
{
    if (m->have_hub) {
        m->device_bytes -= m->dh.bytes;
        spmv_hub_free(&m->dh);
        m->have_hub = false;
    }
}

static void free_plans(spmv_hip_matrix *m)
{
    free_hub(m);
    if (m->d_blk)
        (void)hipFree(m->d_blk);
    if (m->d_seg)
        (void)hipFree(m->d_seg);
    if (m->d_lrow)
        (void)hipFree(m->d_lrow);
    if (m->d_partials)
        (void)hipFree(m->d_partials);
    if (m->d_binrows)
        (void)hipFree(m->d_binrows);
    free_relayouts(m);
    m->d_blk = m->d_seg = m->d_lrow = m->d_binrows = nullptr;
    m->d_partials = nullptr;
    spmv_plan_stream_free(&m->sp);
    if (m->have_bins)
        spmv_plan_bins_free(&m->bp);
    m->have_bins = false;
}

/* the second, re-laid-out copies (STRIPED / TILED) */
static void free_relayouts(spmv_hip_matrix *m)
{
    if (m->have_striped) {
        m->device_bytes -= m->striped_bytes;
        (void)hipFree((void *)m->ds.tptr);
        (void)hipFree((void *)m->ds.rrow);
        (void)hipFree((void *)m->ds.wgd);
        (void)hipFree((void *)m->ds.col);
        (void)hipFree((void *)m->ds.val);
        if (m->ds.partial)
            (void)hipFree(m->ds.partial);
        m->have_striped = false;
    }
    if (m->have_blocked) {
        m->device_bytes -= m->db.bytes;
        spmv_blocked_free(&m->db);
        m->have_blocked = false;
    }
    if (m->have_tiled) {
        m->device_bytes -= m->tiled_bytes;
        (void)hipFree((void *)m->dt.wptr);
        (void)hipFree((void *)m->dt.seg);
        (void)hipFree((void *)m->dt.col);
        (void)hipFree((void *)m->dt.val);
        m->have_tiled = false;
    }
}

void spmv_hip_matrix_destroy(spmv_hip_matrix *m)
{
    if (!m)
        return;
    if (hipSetDevice(m->device) == hipSuccess) {
        if (g_dev[m->device].ready)
            (void)hipStreamSynchronize(g_dev[m->device].stream);
        free_plans(m);
        if (m->d_rowptr)
            (void)hipFree(m->d_rowptr);
        if (m->d_col)
            (void)hipFree(m->d_col);
        if (m->d_val)
            (void)hipFree(m->d_val);
    }
    free(m->h_rowstr_local);
    delete m;
}

static int upload_ints(int **dst, const int *src, size_t n, size_t *acct)
{
    *dst = nullptr;
    HIPCHK(hipMalloc((void **)dst, (n ? n : 4) * sizeof(int)), return -1);
    if (n)
        HIPCHK(hipMemcpy(*dst, src, n * sizeof(int), hipMemcpyHostToDevice), return -1);
    *acct += (n ? n : 4) * sizeof(int);
    return 0;
}

static int ensure_bins(spmv_hip_matrix *m)
{
    if (m->have_bins)
        return 0;
    if (spmv_plan_bins_build(m->h_rowstr_local, 0, m->rows, SPMV_STREAM_CAP - 4, &m->bp)) {
        spmv_set_error("out of host memory building row bins");
        return -1;
    }
    m->have_bins = true;
    if (upload_ints(&m->d_binrows, m->bp.bin_rows, (size_t)m->bp.bin_off[SPMV_NBINS], &m->device_bytes))
        return -1;
    return 0;
}

#ifdef SPMV_HIP_DEBUG_BUILD
/* debug build, tests: the next re-layout upload behaves as if the device were out of memory (tuning key 14); consumed by
 * exactly one of the upload threads */
static std::atomic<int> g_fail_relayout_alloc{0};
static bool take_injected_oom() { return g_fail_relayout_alloc.exchange(0) != 0; }
#else
static bool take_injected_oom() { return false; }
#endif

/*
 * Builds the TILED layout.  host_a / host_col may be the caller's arrays (creation time, `first`
 * = 0-based offset of the block's first entry); when they are NULL the device CSR copy is
 * downloaded instead (plan forced later through spmv_hip_matrix_set_plan).
 */
static int ensure_tiled(spmv_hip_matrix *m, const void *host_a, const int *host_col, int64_t first,
                        bool auto_choice)
{
    if (m->have_tiled)
        return 0;
    void *tmp_a = nullptr;
    int *tmp_c = nullptr;
    if (!host_a) {
        tmp_a = malloc((size_t)(m->nnz ? m->nnz : 1) * m->elem);
        tmp_c = (int *)malloc((size_t)(m->nnz ? m->nnz : 1) * sizeof(int));
        if (!tmp_a || !tmp_c) {
            free(tmp_a), free(tmp_c);
            spmv_set_error("out of host memory for the TILED conversion");
            m->tiled_rc = -3;
            return -3;
        }
        hipStream_t st = g_dev[m->device].stream;
        if (m->nnz > 0) {
            HIPCHK(hipMemcpyAsync(tmp_a, m->d_val, (size_t)m->nnz * m->elem, hipMemcpyDeviceToHost, st),
                   { free(tmp_a); free(tmp_c); return -1; });
            HIPCHK(hipMemcpyAsync(tmp_c, m->d_col, (size_t)m->nnz * sizeof(int), hipMemcpyDeviceToHost, st),
                   { free(tmp_a); free(tmp_c); return -1; });
        }
        HIPCHK(hipStreamSynchronize(st), { free(tmp_a); free(tmp_c); return -1; });
        host_a = tmp_a;
        host_col = tmp_c;
        first = 0;
    }
    /* h_rowstr_local is the block's own 1-based rowstr; shift the array bases so that its
     * absolute indices (rowstr[r]-1) land on the block's first entry */
    const int64_t shift = first - ((int64_t)m->h_rowstr_local[0] - 1);
    spmv_tiled_host th;
    int rc = spmv_tiled_build((const char *)host_a + shift * m->elem, m->h_rowstr_local, host_col + shift, 0,
                              m->rows, m->cols, m->elem, SPMV_TILE_TARGET_BLOCKS, &th);
    free(tmp_a), free(tmp_c);
    if (rc == 0 && m->cols < 16) {
        spmv_tiled_free(&th);
        rc = -2;
    }
    if (rc == 0 && auto_choice && (double)th.nonempty_chunks < 0.5 * (double)th.n_groups * th.S) {
        spmv_tiled_free(&th);
        rc = -2; /* columns are local to the rows: x gathers already hit in cache, STREAM is the better plan */
    }
    m->tiled_rc = rc == -1 ? -3 : rc;
    if (rc == -1) {
        spmv_set_error("out of host memory for the TILED conversion");
        return -3;
    }
    if (rc == -2) {
        spmv_set_error("matrix not representable in the TILED layout (a chunk exceeds 65535 entries) or not suited to it");
        return -2;
    }
    const size_t nch = (size_t)th.n_groups * th.S;
    const size_t n_alloc = (size_t)th.n_packed + SPMV_TILE_PAD;
    void *d_w = nullptr, *d_s = nullptr, *d_c = nullptr, *d_v = nullptr;
    hipError_t e = hipMalloc(&d_w, (nch + 1) * sizeof(int32_t));
    if (e == hipSuccess)
        e = hipMalloc(&d_s, (nch * (th.RW + 1) + 1) * sizeof(uint16_t));
    if (e == hipSuccess)
        e = hipMalloc(&d_c, n_alloc * sizeof(uint16_t));
    if (e == hipSuccess)
        e = hipMalloc(&d_v, n_alloc * (size_t)m->elem);
    if (e == hipSuccess)
        e = hipMemcpy(d_w, th.wptr, (nch + 1) * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(d_s, th.seg, (nch * (th.RW + 1) + 1) * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(d_c, th.col, n_alloc * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(d_v, th.val, n_alloc * (size_t)m->elem, hipMemcpyHostToDevice);
    if (e == hipSuccess && take_injected_oom())
        e = hipErrorOutOfMemory;
    if (e != hipSuccess) {
        spmv_set_error("TILED upload failed: %s", hipGetErrorString(e));
        (void)hipFree(d_w), (void)hipFree(d_s), (void)hipFree(d_c), (void)hipFree(d_v);
        spmv_tiled_free(&th);
        (void)hipGetLastError();
        /* no room for the second copy: not fatal for an automatic choice, the CSR copy serves (-3) */
        m->tiled_rc = e == hipErrorOutOfMemory ? -3 : -1;
        return m->tiled_rc;
    }
    m->dt.wptr = (const int32_t *)d_w;
    m->dt.seg = (const uint16_t *)d_s;
    m->dt.col = (const uint16_t *)d_c;
    m->dt.val = d_v;
    m->dt.elem = m->elem;
    m->dt.W = th.W;
    m->dt.S = th.S;
    m->dt.WPB = th.WPB;
    m->dt.RW = th.RW;
    m->dt.B = th.B;
    m->dt.rows = m->rows;
    m->dt.cols = m->cols;
    m->tiled_bytes = (nch + 1) * 4 + (nch * (th.RW + 1) + 1) * 2 + n_alloc * (2 + (size_t)m->elem);
    m->device_bytes += m->tiled_bytes;
    m->have_tiled = true;
    spmv_tiled_free(&th);
    return 0;
}

/* spmvd_set_relayout(0): matrices created from now on get no re-laid-out copy (STRIPED / TILED) from
 * the automatic choice -- for the ABI path when a matrix keeps changing between calls and the
 * conversion would be paid every time (spmv_harness.c) */
static thread_local int g_relayout = 1; /* per calling thread: the ABI path's upload threads set their own, a direct-API
                                           user creating matrices on another thread is not affected */
extern "C" void spmvd_set_relayout(int on) { g_relayout = on ? 1 : 0; }

/* geometry knobs of the STRIPED plan (experiments through spmv_hip_set_tuning keys 9, 11, 12, 15); read by the upload
 * threads of a row partition side by side, so atomics; set them only while no upload is in flight */
static std::atomic<int> g_stripe_wgs{0}, g_stripe_groups{0}, g_stripe_wave_rows{SPMV_STRIPE_WAVE_ROWS}, g_stripe_shape{0};
static std::atomic<int> g_stripe_xcd_bias{SPMV_STRIPE_XCD_BIAS};

/* Builds the STRIPED layout; same conventions as ensure_tiled.  The conversion runs on the GPU, from the CSR copy that
 * is already in HBM (spmv_striped_gpu.hip), unless the matrix is not a case for it (rows not sorted by column, more than
 * 512 stripes) or tuning key 17 says host: then on host threads (spmv_striped.c) from the caller's arrays or, when those
 * are not at hand, from a download of the CSR copy.  Both builders produce the same bytes. */
static std::atomic<int> g_striped_builder{0}; /* 0: device, host when it must; 1: host; 2: device or fail (tests) */

static int ensure_striped(spmv_hip_matrix *m, const void *host_a, const int *host_col, int64_t first,
                          bool auto_choice)
{
    if (m->have_striped) {
        /* built earlier on explicit request: the automatic choice still declines a layout that is
         * mostly filler */
        if (auto_choice && m->striped_mostly_filler) {
            m->striped_rc = -2;
            return -2;
        }
        return 0;
    }
    spmv_striped_host th;
    memset(&th, 0, sizeof th);
    spmv_striped_opts so;
    spmv_striped_default_opts(&so);
    const int shape = g_stripe_shape.load() ? g_stripe_shape.load() : spmv_striped_choose_shape(&m->st, m->elem, m->cols);
    if (shape == 2) { /* 8 waves over 64-KiB stripes, two workgroups per CU */
        so.xbytes = SPMV_STRIPE2_XBYTES;
        so.wpb = SPMV_STRIPE2_WPB;
        so.target_wgs = 2 * SPMV_STRIPE_TARGET_WGS;
    }
    if (g_stripe_wgs.load() > 0)
        so.target_wgs = g_stripe_wgs.load();
    so.groups = g_stripe_groups.load();
    so.wave_rows = g_stripe_wave_rows.load();
    so.xcd_bias = g_tune_xcd_enabled() ? g_stripe_xcd_bias.load() : 0;

    void *d_c = nullptr, *d_v = nullptr; /* the packed arrays, when the device built them */
    int rc = -4;
    const int builder = g_striped_builder.load();
    if (m->cols < 16)
        rc = -2;
    else if (builder != 1)
        rc = spmv_striped_build_device(m->d_rowptr, m->d_col, m->d_val, m->rows, m->cols, m->elem, m->nnz, &so,
                                       g_dev[m->device].stream, &th, &d_c, &d_v);
    if (rc == -1) { /* a HIP error inside the device builder: its message stands */
        m->striped_rc = -1;
        return -1;
    }
    if (rc == -4 && builder == 2) {
        spmv_set_error("the device builder declined this matrix");
        return -1;
    }
    if (rc == -4) { /* the host builder */
        void *tmp_a = nullptr;
        int *tmp_c = nullptr;
        if (!host_a) {
            tmp_a = malloc((size_t)(m->nnz ? m->nnz : 1) * m->elem);
            tmp_c = (int *)malloc((size_t)(m->nnz ? m->nnz : 1) * sizeof(int));
            if (!tmp_a || !tmp_c) {
                free(tmp_a), free(tmp_c);
                spmv_set_error("out of host memory for the STRIPED conversion");
                m->striped_rc = -3;
                return -3;
            }
            hipStream_t st = g_dev[m->device].stream;
            if (m->nnz > 0) {
                HIPCHK(hipMemcpyAsync(tmp_a, m->d_val, (size_t)m->nnz * m->elem, hipMemcpyDeviceToHost, st),
                       { free(tmp_a); free(tmp_c); return -1; });
                HIPCHK(hipMemcpyAsync(tmp_c, m->d_col, (size_t)m->nnz * sizeof(int), hipMemcpyDeviceToHost, st),
                       { free(tmp_a); free(tmp_c); return -1; });
            }
            HIPCHK(hipStreamSynchronize(st), { free(tmp_a); free(tmp_c); return -1; });
            host_a = tmp_a;
            host_col = tmp_c;
            first = 0;
        }
        const int64_t shift = first - ((int64_t)m->h_rowstr_local[0] - 1);
        rc = spmv_striped_build((const char *)host_a + shift * m->elem, m->h_rowstr_local, host_col + shift, 0, m->rows,
                                m->cols, m->elem, &so, &th);
        free(tmp_a), free(tmp_c);
    }

    /* many (row, stripe) pairs are empty: the columns are local to the rows, x gathers already
     * hit in cache and the STREAM kernel is the better plan (measured: 73 us vs 133 us; with only two
     * stripes a banded matrix leaves just under half of the pairs empty, hence 40 %, while scattered
     * columns at the 1.6-per-segment limit of spmv_plan_choose leave 20 %) -- but only when the empties are more than
     * scattered columns would leave: a row of n entries thrown at S stripes misses S (1 - 1/S)^n of them, and a matrix of
     * many short rows (heavy-tailed row lengths: tools/dispatch_audit.py, 40 000 rows, 3.4 M nonzeros, STRIPED 20.6 us
     * against STREAM 27.8) has its empties for that reason, not because its columns are local; or one dense row makes
     * a wave's stream many times longer than the rest (STREAM cuts long rows into segments) */
    bool local_columns = false;
    if (rc == 0 && th.n_dummies * 5 >= 2 * (int64_t)th.S * m->rows) {
        double expected = 0.0; /* empties if every row's columns were scattered uniformly */
        const double lq = th.S > 1 ? log(1.0 - 1.0 / (double)th.S) : -1e9;
        for (int r = 0; r < m->rows; ++r)
            expected += (double)th.S * exp(lq * (double)(m->h_rowstr_local[r + 1] - m->h_rowstr_local[r]));
        local_columns = (double)th.n_dummies > 1.5 * expected + 0.05 * (double)th.S * m->rows;
    }
    m->striped_mostly_filler = rc == 0 && (local_columns || th.skewed);
    if (auto_choice && m->striped_mostly_filler) {
        spmv_striped_free(&th);
        (void)hipFree(d_c), (void)hipFree(d_v);
        d_c = d_v = nullptr;
        rc = -2;
    }
    m->striped_rc = rc == -1 ? -3 : rc;
    if (rc == -1 || rc == -3) {
        spmv_set_error("out of memory for the STRIPED conversion");
        m->striped_rc = -3;
        return -3;
    }
    if (rc == -2) {
        spmv_set_error("matrix not representable in the STRIPED layout or not suited to it");
        return -2;
    }
    const size_t ntp = (size_t)th.n_ranges * ((size_t)th.GM + 1);
    const size_t n_alloc = (size_t)th.n_packed + SPMV_TILE_PAD;
    const size_t pbytes = th.NG > 1 ? (size_t)th.NG * (size_t)m->rows * (size_t)m->elem + 64 : 0;
    void *d_t = nullptr, *d_r = nullptr, *d_p = nullptr, *d_g = nullptr;
    hipError_t e = hipMalloc(&d_t, ntp * sizeof(int32_t));
    if (e == hipSuccess)
        e = hipMalloc(&d_r, ((size_t)th.n_ranges * 2 + 2) * sizeof(int32_t));
    if (e == hipSuccess)
        e = hipMalloc(&d_g, ((size_t)th.n_wg * 4 + 4) * sizeof(int32_t));
    if (e == hipSuccess)
        e = hipMemcpy(d_g, th.wgd, (size_t)th.n_wg * 4 * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && !d_c)
        e = hipMalloc(&d_c, n_alloc * sizeof(uint16_t));
    if (e == hipSuccess && !d_v)
        e = hipMalloc(&d_v, n_alloc * (size_t)m->elem);
    if (e == hipSuccess && pbytes)
        e = hipMalloc(&d_p, pbytes);
    if (e == hipSuccess)
        e = hipMemcpy(d_t, th.tptr, ntp * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(d_r, th.rrow, (size_t)th.n_ranges * 2 * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && th.col)
        e = hipMemcpy(d_c, th.col, n_alloc * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && th.val)
        e = hipMemcpy(d_v, th.val, n_alloc * (size_t)m->elem, hipMemcpyHostToDevice);
    if (e == hipSuccess && take_injected_oom())
        e = hipErrorOutOfMemory;
    if (e != hipSuccess) {
        spmv_set_error("STRIPED upload failed: %s", hipGetErrorString(e));
        (void)hipFree(d_t), (void)hipFree(d_r), (void)hipFree(d_c), (void)hipFree(d_v), (void)hipFree(d_p);
        (void)hipFree(d_g);
        spmv_striped_free(&th);
        (void)hipGetLastError();
        /* no room for the second copy: not fatal for an automatic choice, the CSR copy serves (-3) */
        m->striped_rc = e == hipErrorOutOfMemory ? -3 : -1;
        return m->striped_rc;
    }
    m->ds.tptr = (const int32_t *)d_t;
    m->ds.rrow = (const int32_t *)d_r;
    m->ds.wgd = (const int32_t *)d_g;
    m->ds.n_wg = th.n_wg;
    m->ds.col = (const uint16_t *)d_c;
    m->ds.val = d_v;
    m->ds.partial = d_p;
    m->ds.elem = m->elem;
    m->ds.W = th.W;
    m->ds.S = th.S;
    m->ds.WPB = th.WPB;
    m->ds.NB = th.NB;
    m->ds.NG = th.NG;
    m->ds.GM = th.GM;
    m->ds.EPL = th.EPL;
    m->ds.n_ranges = th.n_ranges;
    m->ds.wave_rows = so.wave_rows;
    m->ds.xbytes = so.xbytes; /* with WPB: which of the two kernel shapes */
    m->ds.rows = m->rows;
    m->ds.cols = m->cols;
    m->ds.n_packed = th.n_packed;
    m->striped_bytes = ntp * 4 + ((size_t)th.n_ranges * 2 + 2) * 4 + ((size_t)th.n_wg * 4 + 4) * 4 + n_alloc * (2 + (size_t)m->elem) + pbytes;
    m->device_bytes += m->striped_bytes;
    m->have_striped = true;
    spmv_striped_free(&th);
    return 0;
}

/* Builds the BLOCKED layout from the device's CSR copy (spmv_blocked.hip): 0, -2 not representable, -3 no memory, -1 error */
static int ensure_blocked(spmv_hip_matrix *m, bool auto_choice)
{
    if (m->have_blocked)
        return 0;
    int rc = spmv_blocked_build_device(m->d_rowptr, m->d_col, m->d_val, m->rows, m->cols, m->elem, m->nnz, auto_choice ? 1 : 0,
                                       g_dev[m->device].stream, &m->db);
    if (rc == 0 && take_injected_oom()) {
        spmv_blocked_free(&m->db);
        rc = -3;
    }
    m->blocked_rc = rc;
    if (rc == -2)
        spmv_set_error("matrix not representable in the BLOCKED layout (more than 2048 column stripes or 2^24 cells) or not suited to it");
    else if (rc == -3)
        spmv_set_error("out of memory for the BLOCKED layout");
    if (rc)
        return rc;
    m->db.runs = m->st.max_len > 2 * (int64_t)m->db.S; /* a row with two entries per stripe on average: runs of one row's products */
    m->device_bytes += m->db.bytes;
    m->have_blocked = true;
    return 0;
}

static int ensure_hub(spmv_hip_matrix *m, bool auto_choice)
{
    if (m->have_hub)
        return 0;
    int rc = spmv_hub_build_device(m->d_rowptr, m->h_rowstr_local, m->d_col, m->d_val, m->rows, m->cols, m->elem, m->nnz,
                                   auto_choice ? 1 : 0, g_dev[m->device].stream, &m->dh);
    if (rc == 0 && take_injected_oom()) {
        spmv_hub_free(&m->dh);
        rc = -3;
    }
    m->hub_rc = rc;
    if (rc == -2)
        spmv_set_error("matrix not representable in the HUB layout (2^29 columns or more) or not suited to it (too few entries in hub columns)");
    else if (rc == -3)
        spmv_set_error("out of memory for the HUB layout");
    if (rc)
        return rc;
    m->device_bytes += m->dh.bytes;
    m->have_hub = true;
    return 0;
}

/* AUTO's last question, asked when the CSR kernel would otherwise serve a matrix that is large enough to care (2 M nonzeros, 1 MB of x): do a few
 * thousand columns take a good share of the gathers?  (R-MAT 2^20: STREAM 107 us, HUB see profiles/r04_hub_*.log.)  The
 * builder counts the references per column and declines when they do not. */
static bool hub_nominated(const spmv_hip_matrix *m)
{
    return m->nnz >= 2000000 && (int64_t)m->cols * m->elem >= (1ll << 20) && m->cols < (1 << 29);
}

spmv_hip_matrix *spmv_hip_matrix_create(const void *a, const int *rowstr, const int *colidx, int rows,
                                        int elem_size, int device, int row_begin, int row_end,
                                        int cols_hint)
{
    if (elem_size != 8 && elem_size != 4) {
        spmv_set_error("elem_size must be 8 or 4, got %d", elem_size);
        return nullptr;
    }
    if (row_begin < 0 || row_end > rows || row_begin > row_end) {
        spmv_set_error("bad row range [%d,%d) of %d rows", row_begin, row_end, rows);
        return nullptr;
    }
    if (rowstr[row_begin] < 1) {
        spmv_set_error("rowstr[%d] = %d: row pointers are 1-based", row_begin, rowstr[row_begin]);
        return nullptr;
    }
    for (int r = row_begin; r < row_end; ++r)
        if (rowstr[r + 1] < rowstr[r]) {
            spmv_set_error("rowstr decreases at row %d (%d -> %d)", r, rowstr[r], rowstr[r + 1]);
            return nullptr;
        }
    if (use_device(device))
        return nullptr;
    hipStream_t st = g_dev[device].stream;

    spmv_hip_matrix *m = new spmv_hip_matrix();
    m->device = device;
    m->elem = elem_size;
    m->rows = row_end - row_begin;
    const int64_t first = (int64_t)rowstr[row_begin] - 1; /* 0-based offset into a / colidx */
    m->first = first;
    m->nnz = (int64_t)rowstr[row_end] - (int64_t)rowstr[row_begin];
    const size_t n_alloc = (size_t)m->nnz + SPMV_PAD_ELEMS;

#define FAIL()                         \
    do {                               \
        spmv_hip_matrix_destroy(m);    \
        return nullptr;                \
    } while (0)

    HIPCHK(hipMalloc((void **)&m->d_rowptr, ((size_t)m->rows + 1) * sizeof(int)), FAIL());
    HIPCHK(hipMalloc((void **)&m->d_col, n_alloc * sizeof(int)), FAIL());
    HIPCHK(hipMalloc(&m->d_val, n_alloc * (size_t)elem_size), FAIL());
    m->device_bytes = ((size_t)m->rows + 1) * sizeof(int) + n_alloc * (sizeof(int) + (size_t)elem_size);

    HIPCHK(hipMemcpyAsync(m->d_rowptr, rowstr + row_begin, ((size_t)m->rows + 1) * sizeof(int),
                          hipMemcpyHostToDevice, st),
           FAIL());
    if (m->nnz > 0) {
        HIPCHK(hipMemcpyAsync(m->d_col, colidx + first, (size_t)m->nnz * sizeof(int),
                              hipMemcpyHostToDevice, st),
               FAIL());
        HIPCHK(hipMemcpyAsync(m->d_val, (const char *)a + first * elem_size, (size_t)m->nnz * elem_size,
                              hipMemcpyHostToDevice, st),
               FAIL());
    }
    /* padding: colidx = 1 (a valid column), a = 0 */
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(m->d_col + m->nnz), 1, SPMV_PAD_ELEMS, st), FAIL());
    HIPCHK(hipMemsetAsync((char *)m->d_val + (size_t)m->nnz * elem_size, 0,
                          (size_t)SPMV_PAD_ELEMS * elem_size, st),
           FAIL());
    if (spmvk_rebase(m->d_rowptr, m->rows + 1, rowstr[row_begin], st)) {
        spmv_set_error("rebase kernel launch failed");
        FAIL();
    }

    /*
     * Column range of the block, by a device reduction over what was just uploaded.  The reference
     * trusts the caller (an index < 1 reads before x, native-impl.c:9); on a GPU shared with other
     * jobs an out-of-range gather can fault the device for everybody, so it is refused here.
     */
    int h_range[2] = {0, 0x7fffffff};
    if (m->nnz > 0) {
        int *d_range = nullptr;
        HIPCHK(hipMalloc((void **)&d_range, 2 * sizeof(int)), FAIL());
        hipError_t e0 = hipMemcpyAsync(d_range, h_range, sizeof h_range, hipMemcpyHostToDevice, st);
        int rc = spmvk_colrange(m->d_col, m->nnz, d_range, st);
        hipError_t e1 = hipMemcpyAsync(h_range, d_range, sizeof h_range, hipMemcpyDeviceToHost, st);
        hipError_t e2 = hipStreamSynchronize(st);
        (void)hipFree(d_range);
        if (rc || e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess) {
            spmv_set_error("column-range reduction failed");
            FAIL();
        }
        if (h_range[1] < 1) {
            spmv_set_error("column index %d < 1 (indices are 1-based)", h_range[1]);
            FAIL();
        }
    }
    /* number of columns = max(colidx) over the whole matrix (libspmv/gpu.c:217-223) */
    if (cols_hint > 0) {
        if (h_range[0] > cols_hint) {
            spmv_set_error("column index %d beyond the %d columns announced", h_range[0], cols_hint);
            FAIL();
        }
        m->cols = cols_hint;
    } else if (row_begin == 0 && row_end == rows) {
        m->cols = h_range[0];
    } else {
        const int64_t all = (int64_t)rowstr[rows] - rowstr[0];
        const int *c = colidx + (rowstr[0] - 1);
        int mx = 0;
        for (int64_t j = 0; j < all; ++j)
            mx = c[j] > mx ? c[j] : mx;
        m->cols = mx;
    }

    /* host-side planning from the caller's rowstr */
    m->h_rowstr_local = (int *)malloc(((size_t)m->rows + 1) * sizeof(int));
    if (!m->h_rowstr_local) {
        spmv_set_error("out of host memory");
        FAIL();
    }
    memcpy(m->h_rowstr_local, rowstr + row_begin, ((size_t)m->rows + 1) * sizeof(int));
    spmv_plan_row_stats(m->h_rowstr_local, 0, m->rows, &m->st);
    if (spmv_plan_stream_build(m->h_rowstr_local, 0, m->rows, SPMV_STREAM_CAP - 4, SPMV_STREAM_ROWCAP,
                               SPMV_LONG_SEG, &m->sp)) {
        spmv_set_error("out of host memory building the row-block plan");
        FAIL();
    }
    if (upload_ints(&m->d_blk, m->sp.blk, 2 * ((size_t)m->sp.nblk + 1), &m->device_bytes))
        FAIL();
    if (m->sp.nseg > 0) {
        if (upload_ints(&m->d_seg, m->sp.seg, 4 * (size_t)m->sp.nseg, &m->device_bytes))
            FAIL();
        if (upload_ints(&m->d_lrow, m->sp.lrow, 3 * (size_t)m->sp.nlrow, &m->device_bytes))
            FAIL();
        HIPCHK(hipMalloc(&m->d_partials, ((size_t)m->sp.nslots + 1) * (size_t)elem_size), FAIL());
        m->device_bytes += ((size_t)m->sp.nslots + 1) * (size_t)elem_size;
    }
    HIPCHK(hipStreamSynchronize(st), FAIL());

    spmv_plan_choose(&m->st, elem_size, m->cols, &m->plan, &m->param);
    if (!g_relayout && (m->plan == SPMV_HIP_PLAN_TILED || m->plan == SPMV_HIP_PLAN_STRIPED || m->plan == SPMV_HIP_PLAN_BLOCKED)) {
        m->plan = SPMV_HIP_PLAN_STREAM; /* the caller expects to upload this matrix again soon */
        m->param = SPMV_STREAM_CAP;
    }
    if (m->plan == SPMV_HIP_PLAN_TILED) {
        int rc = ensure_tiled(m, a, colidx, first, true);
        if (rc == -2 || rc == -3) { /* not representable, or no memory for a second copy: the row-segment kernel takes it */
            m->plan = SPMV_HIP_PLAN_STREAM;
            m->param = SPMV_STREAM_CAP;
        } else if (rc) {
            FAIL();
        }
    }
    if (m->plan == SPMV_HIP_PLAN_STRIPED) {
        int rc = ensure_striped(m, a, colidx, first, true);
        if (rc == -2 || rc == -3) {
            m->plan = SPMV_HIP_PLAN_STREAM;
            m->param = SPMV_STREAM_CAP;
        } else if (rc) {
            FAIL();
        }
    }
    /* a power-law graph is nominated for BLOCKED by its size alone and declined by BLOCKED's builder after a count kernel of
     * 2-4.5 ms (R-MAT 2^20: a crowded cell); its rows say what it is -- the longest 64 times the mean -- so HUB is asked first
     * there, and BLOCKED only when HUB's builder finds no hubs */
    if (m->plan == SPMV_HIP_PLAN_BLOCKED && g_relayout && hub_nominated(m) && (double)m->st.max_len > 64.0 * m->st.mean_len) {
        int rc = ensure_hub(m, true);
        if (rc == 0) {
            m->plan = SPMV_HIP_PLAN_HUB;
            m->param = m->dh.K;
        } else if (rc != -2 && rc != -3) {
            FAIL();
        }
    }
    if (m->plan == SPMV_HIP_PLAN_BLOCKED) {
        int rc = ensure_blocked(m, true);
        if (rc == -2 || rc == -3) {
            m->plan = SPMV_HIP_PLAN_STREAM;
            m->param = SPMV_STREAM_CAP;
        } else if (rc) {
            FAIL();
        } else {
            m->param = m->db.wcols;
        }
    }
    if (m->plan == SPMV_HIP_PLAN_STREAM && g_relayout && hub_nominated(m) && !m->have_hub && m->hub_rc == 0) {
        int rc = ensure_hub(m, true);
        if (rc == 0) {
            m->plan = SPMV_HIP_PLAN_HUB;
            m->param = m->dh.K;
        } else if (rc != -2 && rc != -3) {
            FAIL();
        }
    }
    if (m->plan == SPMV_HIP_PLAN_BINNED && ensure_bins(m))
        FAIL();
#undef FAIL
    return m;
}

int spmv_hip_matrix_refresh_values(spmv_hip_matrix *m, const void *a, const int *colidx)
{
    if (!m || !a) {
        spmv_set_error("refresh_values: no matrix or no values");
        return -1;
    }
    if (use_device(m->device))
        return -1;
    hipStream_t st = g_dev[m->device].stream;
    if (m->nnz > 0)
        HIPCHK(hipMemcpyAsync(m->d_val, (const char *)a + m->first * m->elem, (size_t)m->nnz * m->elem,
                              hipMemcpyHostToDevice, st),
               return -1);
    HIPCHK(hipStreamSynchronize(st), return -1); /* the caller may write `a` again as soon as this returns */
    if (m->have_striped || m->have_tiled || m->have_blocked) {
        /* the re-laid-out copies hold the values in their own order: the one in use is packed again (the plan is kept
         * when that works), one that is not in use is dropped (its values are stale) */
        const bool was_striped = m->plan == SPMV_HIP_PLAN_STRIPED, was_tiled = m->plan == SPMV_HIP_PLAN_TILED;
        const bool was_blocked = m->plan == SPMV_HIP_PLAN_BLOCKED;
        free_relayouts(m);
        int rc = 0;
        if (was_blocked)
            rc = ensure_blocked(m, false);
        else if (was_striped)
            rc = ensure_striped(m, colidx ? a : nullptr, colidx, m->first, false);
        else if (was_tiled)
            rc = ensure_tiled(m, colidx ? a : nullptr, colidx, m->first, false);
        if (rc) {
            /* -2 / -3 cannot happen for a layout that existed a moment ago, short of memory; any other failure is reported.
             * Either way the re-laid-out copy is gone and the CSR copy holds the new values: the CSR kernels serve from
             * here on, also for a caller that ignores the error (the plan used to stay BLOCKED with no layout
             * behind it, and the next product returned 0 without writing y) */
            m->plan = SPMV_HIP_PLAN_STREAM;
            m->param = SPMV_STREAM_CAP;
            if (rc != -2 && rc != -3)
                return -1;
        }
    }
    return 0;
}

int spmv_hip_matrix_set_plan(spmv_hip_matrix *m, int plan, int plan_param)
{
    if (!m)
        return -1;
    if (hipSetDevice(m->device) != hipSuccess)
        return -1;
    switch (plan) {
    case SPMV_HIP_PLAN_AUTO:
        spmv_plan_choose(&m->st, m->elem, m->cols, &m->plan, &m->param);
        if (m->plan == SPMV_HIP_PLAN_TILED && ensure_tiled(m, nullptr, nullptr, 0, true)) {
            if (m->tiled_rc != -2 && m->tiled_rc != -3)
                return -1;
            m->plan = SPMV_HIP_PLAN_STREAM;
            m->param = SPMV_STREAM_CAP;
        }
        if (m->plan == SPMV_HIP_PLAN_STRIPED && ensure_striped(m, nullptr, nullptr, 0, true)) {
            if (m->striped_rc != -2 && m->striped_rc != -3)
                return -1;
            m->plan = SPMV_HIP_PLAN_STREAM;
            m->param = SPMV_STREAM_CAP;
        }
        if (m->plan == SPMV_HIP_PLAN_BLOCKED && hub_nominated(m) && (double)m->st.max_len > 64.0 * m->st.mean_len) { /* as at creation */
            if (ensure_hub(m, true) == 0) {
                m->plan = SPMV_HIP_PLAN_HUB;
                m->param = m->dh.K;
            } else if (m->hub_rc != -2 && m->hub_rc != -3) {
                return -1;
            }
        }
        if (m->plan == SPMV_HIP_PLAN_BLOCKED) {
            if (ensure_blocked(m, true)) {
                if (m->blocked_rc != -2 && m->blocked_rc != -3)
                    return -1;
                m->plan = SPMV_HIP_PLAN_STREAM;
                m->param = SPMV_STREAM_CAP;
            } else {
                m->param = m->db.wcols;
            }
        }
        if (m->plan == SPMV_HIP_PLAN_STREAM && hub_nominated(m)) {
            if (ensure_hub(m, true) == 0) {
                m->plan = SPMV_HIP_PLAN_HUB;
                m->param = m->dh.K;
            } else if (m->hub_rc != -2 && m->hub_rc != -3) {
                return -1;
            }
        }
        break;
    case SPMV_HIP_PLAN_HUB:
        if (ensure_hub(m, false))
            return -1;
        m->plan = plan;
        m->param = m->dh.K;
        break;
    case SPMV_HIP_PLAN_BLOCKED:
        if (ensure_blocked(m, false))
            return -1;
        m->plan = plan;
        m->param = m->db.wcols;
        break;
    case SPMV_HIP_PLAN_TILED:
        if (ensure_tiled(m, nullptr, nullptr, 0, false))
            return -1;
        m->plan = plan;
        m->param = m->dt.W;
        break;
    case SPMV_HIP_PLAN_STRIPED:
        if (ensure_striped(m, nullptr, nullptr, 0, false))
            return -1;
        m->plan = plan;
        m->param = m->ds.W;
        break;
    case SPMV_HIP_PLAN_ORDERED:
        m->plan = plan;
        m->param = 1;
        break;
    case SPMV_HIP_PLAN_STREAM:
        m->plan = plan;
        m->param = SPMV_STREAM_CAP;
        break;
    case SPMV_HIP_PLAN_VECTOR: {
        int l = plan_param > 0 ? plan_param : 64;
        if (l > 64 || (l & (l - 1))) {
            spmv_set_error("VECTOR lanes must be a power of two <= 64, got %d", l);
            return -1;
        }
        m->plan = plan;
        m->param = l;
        break;
    }
    case SPMV_HIP_PLAN_BINNED:
        m->plan = plan;
        m->param = 0;
        break;
    default: spmv_set_error("unknown plan %d", plan); return -1;
    }
    if (m->plan == SPMV_HIP_PLAN_BINNED && ensure_bins(m))
        return -1;
    return 0;
}

static int n_launches(const spmv_hip_matrix *m)
{
    int longk = m->sp.nseg > 0 ? 1 + (m->sp.nlrow > 0) : 0;
    switch (m->plan) {
    case SPMV_HIP_PLAN_STRIPED: return m->ds.NG > 1 ? 2 : 1;
    case SPMV_HIP_PLAN_BLOCKED: return m->db.n_join > 0 ? 3 : 2; /* + k_blk_join when a bin was cut */
    case SPMV_HIP_PLAN_STREAM: return 1 + longk;
    case SPMV_HIP_PLAN_HUB: return 1 + (m->dh.K + m->dh.K2 > 0) + (m->dh.n_fix > 0 || m->dh.n_empty > 0); /* k_hub_pack, k_hub, k_hub_fix */
    case SPMV_HIP_PLAN_BINNED: {
        int n = longk;
        for (int b = 0; b < SPMV_NBINS; ++b)
            n += m->bp.bin_off[b + 1] > m->bp.bin_off[b];
        return n;
    }
    default: return 1;
    }
}

int spmv_hip_matrix_info(const spmv_hip_matrix *m, spmv_hip_info *out)
{
    if (!m || !out)
        return -1;
    memset(out, 0, sizeof *out);
    out->rows = m->rows;
    out->cols = m->cols;
    out->nnz = m->nnz;
    out->elem_size = m->elem;
    out->device = m->device;
    out->plan = m->plan;
    out->plan_param = m->param;
    out->min_row_len = m->rows ? m->st.min_len : 0;
    out->max_row_len = m->st.max_len;
    out->mean_row_len = m->st.mean_len;
    out->n_blocks = m->plan == SPMV_HIP_PLAN_TILED     ? m->dt.B
                    : m->plan == SPMV_HIP_PLAN_STRIPED ? m->ds.n_wg
                    : m->plan == SPMV_HIP_PLAN_BLOCKED ? m->db.n_wg
                    : m->plan == SPMV_HIP_PLAN_HUB     ? m->dh.n_chunks
                                                       : m->sp.nblk;
    out->n_long_rows = m->sp.n_long_rows;
    out->n_launches = n_launches(m);
    out->algorithmic_bytes = m->nnz * (int64_t)(m->elem + 4) + ((int64_t)m->rows + 1) * 4 +
                             (int64_t)m->cols * m->elem + (int64_t)m->rows * m->elem;
    out->device_bytes = (int64_t)m->device_bytes;
    return 0;
}

/*
 * y = A x and, in the same launches, the block partials of y . v (spmv_cg.hip: q = A p and p.q of a
 * CG step).  Only the STRIPED plan can fold the dot into its last kernel; returns the number of
 * partials written to dot_part (one per 256 rows), 0 when the plan cannot (the caller then runs its
 * own dot over y), -1 on error.
 */
int spmvd_spmv_dot(spmv_hip_matrix *m, const void *x_dev, void *y_dev, void *stream, const void *v_dev,
                   void *dot_part)
{
    if (!m || m->rows == 0 || m->plan != SPMV_HIP_PLAN_STRIPED || (reinterpret_cast<uintptr_t>(x_dev) & 15u) != 0)
        return 0;
    HIPCHK(hipSetDevice(m->device), return -1);
    hipStream_t st = stream_of(m->device, stream);
    int rc = spmvk_striped(&m->ds, x_dev, y_dev, st, v_dev, dot_part);
    if (rc) {
        spmv_set_error("kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        return -1;
    }
    return m->ds.NG > 1 ? (m->rows + SPMV_WG - 1) / SPMV_WG : m->ds.n_ranges;
}

int spmv_hip_spmv_device(spmv_hip_matrix *m, const void *x_dev, void *y_dev, void *stream)
{
    if (!m)
        return -1;
    if (m->rows == 0)
        return 0;
    HIPCHK(hipSetDevice(m->device), return -1);
    hipStream_t st = stream_of(m->device, stream);
    spmv_dev_csr d;
    d.rowptr = m->d_rowptr;
    d.colidx = m->d_col;
    d.val = m->d_val;
    d.rows = m->rows;
    d.nnz = m->nnz;
    d.elem_size = m->elem;
    int rc = 0;
    switch (m->plan) {
    case SPMV_HIP_PLAN_ORDERED: rc = spmvk_ordered(&d, x_dev, y_dev, st); break;
    case SPMV_HIP_PLAN_HUB:
        if (m->have_hub) {
            rc = spmvk_hub(&m->dh, x_dev, y_dev, st);
            break;
        }
        rc = spmvk_stream(&d, m->d_blk, m->sp.nblk, x_dev, y_dev, st);
        if (!rc && m->sp.nseg > 0)
            rc = spmvk_long(&d, m->d_seg, m->sp.nseg, m->d_lrow, m->sp.nlrow, m->d_partials, x_dev, y_dev, st);
        break;
    case SPMV_HIP_PLAN_BLOCKED:
    case SPMV_HIP_PLAN_STRIPED:
    case SPMV_HIP_PLAN_TILED:
        if ((reinterpret_cast<uintptr_t>(x_dev) & 15u) == 0 && (m->plan != SPMV_HIP_PLAN_BLOCKED || m->have_blocked)) {
            rc = m->plan == SPMV_HIP_PLAN_TILED     ? spmvk_tiled(&m->dt, x_dev, y_dev, st)
                 : m->plan == SPMV_HIP_PLAN_BLOCKED ? spmvk_blocked(&m->db, x_dev, y_dev, st)
                                                    : spmvk_striped(&m->ds, x_dev, y_dev, st, nullptr, nullptr);
            break;
        }
        /* the three re-laid-out plans stage x with 16-byte loads; a misaligned x goes to the row-segment kernel */
        rc = spmvk_stream(&d, m->d_blk, m->sp.nblk, x_dev, y_dev, st);
        if (!rc && m->sp.nseg > 0)
            rc = spmvk_long(&d, m->d_seg, m->sp.nseg, m->d_lrow, m->sp.nlrow, m->d_partials, x_dev,
                            y_dev, st);
        break;
    case SPMV_HIP_PLAN_VECTOR: rc = spmvk_vector(&d, m->param, nullptr, m->rows, x_dev, y_dev, st); break;
    case SPMV_HIP_PLAN_BINNED:
        for (int b = 0; b < SPMV_NBINS && !rc; ++b) {
            int n = m->bp.bin_off[b + 1] - m->bp.bin_off[b];
            if (n > 0)
                rc = spmvk_vector(&d, spmv_plan_bin_lanes(b), m->d_binrows + m->bp.bin_off[b], n, x_dev,
                                  y_dev, st);
        }
        if (!rc && m->sp.nseg > 0)
            rc = spmvk_long(&d, m->d_seg, m->sp.nseg, m->d_lrow, m->sp.nlrow, m->d_partials, x_dev,
                            y_dev, st);
        break;
    case SPMV_HIP_PLAN_STREAM:
    default:
        rc = spmvk_stream(&d, m->d_blk, m->sp.nblk, x_dev, y_dev, st);
        if (!rc && m->sp.nseg > 0)
            rc = spmvk_long(&d, m->d_seg, m->sp.nseg, m->d_lrow, m->sp.nlrow, m->d_partials, x_dev,
                            y_dev, st);
        break;
    }
    if (rc) {
        spmv_set_error("kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        return -1;
    }
    return 0;
}

/*
 * y = A x with the rows cut into `chunks` launches, each chunk's rows of y setting off for host memory on a second
 * stream as soon as its launch is done, so that the D2H copy of a long y (SparseBench 170^3: 39 MB, 0.7 ms at PCIe
 * speed) runs beside the kernels that still compute (libspmv/gpu.c:285 copies y after the whole product).  Only for the
 * plans whose launches write whole, contiguous runs of rows (STREAM and its long-row kernels, VECTOR).  Returns 1 when
 * the product and the copy were done this way (both complete on return), 0 when the plan does not cut into row chunks
 * (nothing was launched), -1 on error.
 */
int spmvd_spmv_y_overlapped(spmv_hip_matrix *m, const void *x_dev, void *y_dev, void *y_host, int chunks)
{
    if (!m || m->rows == 0 || (m->plan != SPMV_HIP_PLAN_STREAM && m->plan != SPMV_HIP_PLAN_VECTOR))
        return 0;
    if (chunks > SPMV_Y_CHUNKS_MAX)
        chunks = SPMV_Y_CHUNKS_MAX;
    const int units = m->plan == SPMV_HIP_PLAN_STREAM ? m->sp.nblk : m->rows;
    if (chunks < 2 || units < chunks)
        return 0;
    if (use_device(m->device))
        return -1;
    DevCtx &dc = g_dev[m->device];
    if (!dc.copy_stream) {
        HIPCHK(hipStreamCreateWithFlags(&dc.copy_stream, hipStreamNonBlocking), return -1);
        for (int c = 0; c < SPMV_Y_CHUNKS_MAX; ++c)
            HIPCHK(hipEventCreateWithFlags(&dc.chunk_ev[c], hipEventDisableTiming), return -1);
    }
    hipStream_t st = dc.stream;
    spmv_dev_csr d;
    d.rowptr = m->d_rowptr, d.colidx = m->d_col, d.val = m->d_val, d.rows = m->rows, d.nnz = m->nnz, d.elem_size = m->elem;
    int rc = 0;
    if (m->plan == SPMV_HIP_PLAN_STREAM && m->sp.nseg > 0) /* rows too long for a row block: their y first */
        rc = spmvk_long(&d, m->d_seg, m->sp.nseg, m->d_lrow, m->sp.nlrow, m->d_partials, x_dev, y_dev, st);
    int row_lo[SPMV_Y_CHUNKS_MAX + 1];
    for (int c = 0; c < chunks && !rc; ++c) {
        const int u0 = (int)((int64_t)units * c / chunks), u1 = (int)((int64_t)units * (c + 1) / chunks);
        if (m->plan == SPMV_HIP_PLAN_STREAM) {
            row_lo[c] = m->sp.blk[2 * u0];
            row_lo[c + 1] = m->sp.blk[2 * u1];
            rc = spmvk_stream(&d, m->d_blk + 2 * u0, u1 - u0, x_dev, y_dev, st);
        } else {
            row_lo[c] = u0, row_lo[c + 1] = u1;
            spmv_dev_csr dd = d;
            dd.rowptr = m->d_rowptr + u0;
            dd.rows = u1 - u0;
            rc = spmvk_vector(&dd, m->param, nullptr, u1 - u0, x_dev, (char *)y_dev + (size_t)u0 * m->elem, st);
        }
        if (!rc)
            HIPCHK(hipEventRecord(dc.chunk_ev[c], st), return -1);
    }
    if (rc) {
        spmv_set_error("kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        return -1;
    }
    /* every launch is queued; now the copies, each behind its chunk's event (a copy to pageable memory holds the host
     * thread until it is done, which is why they come last) */
    for (int c = 0; c < chunks; ++c) {
        const size_t off = (size_t)row_lo[c] * m->elem, len = (size_t)(row_lo[c + 1] - row_lo[c]) * m->elem;
        HIPCHK(hipStreamWaitEvent(dc.copy_stream, dc.chunk_ev[c], 0), return -1);
        if (len)
            HIPCHK(hipMemcpyAsync((char *)y_host + off, (const char *)y_dev + off, len, hipMemcpyDeviceToHost, dc.copy_stream),
                   return -1);
    }
    HIPCHK(hipStreamSynchronize(dc.copy_stream), return -1);
    HIPCHK(hipStreamSynchronize(st), return -1);
    return 1;
}

/*
 * One process, several GPUs: `reps` products of each of n device-resident matrices (usually the row blocks of one
 * matrix, one per GPU), launched rep by rep on each matrix's own stream; ms_each[i] = matrix i's elapsed time between
 * HIP events on ITS stream.  All streams have drained on return.  bench.py times its single-process N-GPU runs with it.
 */
int spmv_hip_time_multi(spmv_hip_matrix **ms, void *const *x_dev, void *const *y_dev, int n, int reps, float *ms_each)
{
    if (!ms || n <= 0 || n > 64 || reps <= 0 || !ms_each)
        return -1;
    hipEvent_t ev[64][2];
    for (int i = 0; i < n; ++i) {
        if (!ms[i] || use_device(ms[i]->device))
            return -1;
        HIPCHK(hipEventCreate(&ev[i][0]), return -1);
        HIPCHK(hipEventCreate(&ev[i][1]), return -1);
    }
    int rc = 0;
    for (int i = 0; i < n && !rc; ++i) {
        HIPCHK(hipSetDevice(ms[i]->device), rc = -1);
        HIPCHK(hipEventRecord(ev[i][0], g_dev[ms[i]->device].stream), rc = -1);
    }
    for (int r = 0; r < reps && !rc; ++r)
        for (int i = 0; i < n && !rc; ++i)
            rc = spmv_hip_spmv_device(ms[i], x_dev[i], y_dev[i], nullptr);
    for (int i = 0; i < n && !rc; ++i) {
        HIPCHK(hipSetDevice(ms[i]->device), rc = -1);
        HIPCHK(hipEventRecord(ev[i][1], g_dev[ms[i]->device].stream), rc = -1);
    }
    for (int i = 0; i < n; ++i) {
        if (!rc) {
            HIPCHK(hipSetDevice(ms[i]->device), rc = -1);
            HIPCHK(hipEventSynchronize(ev[i][1]), rc = -1);
            HIPCHK(hipEventElapsedTime(&ms_each[i], ev[i][0], ev[i][1]), rc = -1);
        }
        (void)hipEventDestroy(ev[i][0]);
        (void)hipEventDestroy(ev[i][1]);
    }
    return rc;
}

int spmv_hip_time_device(spmv_hip_matrix *m, const void *x_dev, void *y_dev, int reps, void *stream,
                         float *ms_total)
{
    if (!m || reps <= 0 || !ms_total)
        return -1;
    HIPCHK(hipSetDevice(m->device), return -1);
    hipStream_t st = stream_of(m->device, stream);
    DevCtx &dc = g_dev[m->device];
    if (!dc.ev[0]) {
        HIPCHK(hipEventCreate(&dc.ev[0]), return -1);
        HIPCHK(hipEventCreate(&dc.ev[1]), return -1);
    }
    const hipEvent_t e0 = dc.ev[0], e1 = dc.ev[1];
    int rc = 0;
    HIPCHK(hipEventRecord(e0, st), rc = -1);
    for (int i = 0; i < reps && !rc; ++i)
        rc = spmv_hip_spmv_device(m, x_dev, y_dev, st);
    if (!rc) {
        HIPCHK(hipEventRecord(e1, st), rc = -1);
    }
    if (!rc) {
        HIPCHK(hipEventSynchronize(e1), rc = -1);
    }
    if (!rc) {
        HIPCHK(hipEventElapsedTime(ms_total, e0, e1), rc = -1);
    }
    return rc;
}

/*
 * What one ABI call moves besides the kernel (libspmv/gpu.c:264,285): x from host memory to the device, the
 * product, y back to host memory -- `reps` times back to back on one stream, bracketed by HIP events.  The
 * host pointers should be pinned (spmv_hip_host_alloc / spmv_hip_host_register) for the copies to be DMA
 * transfers that overlap nothing but themselves; pageable memory works and is slower.  y_host receives the
 * matrix's rows, i.e. a rank of the row-partitioned path passes the address of ITS slice of the shared y.
 */
int spmv_hip_time_call(spmv_hip_matrix *m, const void *x_host, void *y_host, void *x_dev, void *y_dev, int reps,
                       void *stream, float *ms_total)
{
    if (!m || reps <= 0 || !ms_total || !x_host || !y_host)
        return -1;
    HIPCHK(hipSetDevice(m->device), return -1);
    hipStream_t st = stream_of(m->device, stream);
    DevCtx &dc = g_dev[m->device];
    if (!dc.ev[0]) {
        HIPCHK(hipEventCreate(&dc.ev[0]), return -1);
        HIPCHK(hipEventCreate(&dc.ev[1]), return -1);
    }
    const size_t xb = (size_t)m->cols * m->elem, yb = (size_t)m->rows * m->elem;
    HIPCHK(hipEventRecord(dc.ev[0], st), return -1);
    for (int i = 0; i < reps; ++i) {
        HIPCHK(hipMemcpyAsync(x_dev, x_host, xb, hipMemcpyHostToDevice, st), return -1);
        if (spmv_hip_spmv_device(m, x_dev, y_dev, st))
            return -1;
        if (yb)
            HIPCHK(hipMemcpyAsync(y_host, y_dev, yb, hipMemcpyDeviceToHost, st), return -1);
    }
    HIPCHK(hipEventRecord(dc.ev[1], st), return -1);
    HIPCHK(hipEventSynchronize(dc.ev[1]), return -1);
    HIPCHK(hipEventElapsedTime(ms_total, dc.ev[0], dc.ev[1]), return -1);
    return 0;
}

void *spmv_hip_host_alloc(size_t bytes) { return spmvd_host_alloc(bytes); }
void spmv_hip_host_free(void *p) { spmvd_host_free(p); }
int spmv_hip_host_register(void *p, size_t bytes) { return spmvd_host_register(p, bytes); }
void spmv_hip_host_unregister(void *p) { spmvd_host_unregister(p); }

#ifdef SPMV_HIP_DEBUG_BUILD
extern "C" int spmvk_striped_stamps(unsigned long long *out, int n);
/* debug build (include/spmv_hip_debug.h): the per-workgroup clock stamps a k_striped<DIAG> launch leaves under tuning
 * key 6 value 64 (100-MHz ticks: entry, first stripe staged, wave 0 done, last wave done) */
int spmv_hip_debug_stamps(int device, unsigned long long *out, int n)
{
    if (use_device(device))
        return -1;
    HIPCHK(hipDeviceSynchronize(), return -1);
    return spmvk_striped_stamps(out, n);
}

/* debug build: FNV-1a digests of the five arrays of a STRIPED layout (tptr, rrow, wgd, col, val incl. the look-ahead
 * padding), downloaded: two layouts are the same bytes iff these agree (the device builder against the host builder) */
int spmv_hip_debug_striped_digest(const spmv_hip_matrix *m, unsigned long long *out5)
{
    if (!m || !m->have_striped || use_device(m->device))
        return -1;
    const spmv_dev_striped &d = m->ds;
    size_t n_packed = 0;
    {
        int32_t last = 0;
        const size_t ntp = (size_t)d.n_ranges * ((size_t)d.GM + 1);
        if (hipMemcpy(&last, d.tptr + ntp - 1, 4, hipMemcpyDeviceToHost) != hipSuccess)
            return -1;
        n_packed = (size_t)last;
    }
    const size_t n_alloc = n_packed + SPMV_TILE_PAD;
    const void *ptr[5] = {d.tptr, d.rrow, d.wgd, d.col, d.val};
    const size_t len[5] = {(size_t)d.n_ranges * ((size_t)d.GM + 1) * 4, (size_t)d.n_ranges * 8, (size_t)d.n_wg * 16, n_alloc * 2,
                           n_alloc * (size_t)d.elem};
    for (int k = 0; k < 5; ++k) {
        unsigned char *h = (unsigned char *)malloc(len[k] ? len[k] : 1);
        if (!h || hipMemcpy(h, ptr[k], len[k], hipMemcpyDeviceToHost) != hipSuccess) {
            free(h);
            return -1;
        }
        unsigned long long x = 1469598103934665603ull;
        const unsigned long long *w = (const unsigned long long *)h;
        for (size_t i = 0; i < len[k] / 8; ++i) /* word-wise FNV-1a: a digest, not a standard hash */
            x = (x ^ w[i]) * 1099511628211ull;
        for (size_t i = len[k] / 8 * 8; i < len[k]; ++i)
            x = (x ^ h[i]) * 1099511628211ull;
        out5[k] = x;
        free(h);
    }
    return 0;
}

/* debug build: one of the five arrays of a STRIPED layout (0 tptr, 1 rrow, 2 wgd, 3 col, 4 val) copied to `out` (at most
 * `bytes`); returns the array's size in bytes, -1 on error */
long long spmv_hip_debug_striped_array(const spmv_hip_matrix *m, int which, void *out, long long bytes)
{
    if (!m || !m->have_striped || which < 0 || which > 4 || use_device(m->device))
        return -1;
    const spmv_dev_striped &d = m->ds;
    int32_t last = 0;
    const size_t ntp = (size_t)d.n_ranges * ((size_t)d.GM + 1);
    if (hipMemcpy(&last, d.tptr + ntp - 1, 4, hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    const size_t n_alloc = (size_t)last + SPMV_TILE_PAD;
    const void *ptr[5] = {d.tptr, d.rrow, d.wgd, d.col, d.val};
    const size_t len[5] = {ntp * 4, (size_t)d.n_ranges * 8, (size_t)d.n_wg * 16, n_alloc * 2, n_alloc * (size_t)d.elem};
    const size_t n = len[which] < (size_t)bytes ? len[which] : (size_t)bytes;
    if (out && n && hipMemcpy(out, ptr[which], n, hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    return (long long)len[which];
}

/* debug build: what each workgroup of a STRIPED layout holds -- per workgroup {first stripe, end stripe, group, entries of
 * all its waves (pads included), entries of its longest wave, rows}; returns the number of workgroups, -1 on error */
int spmv_hip_debug_striped_wgs(const spmv_hip_matrix *m, long long *out, int max_wgs)
{
    if (!m || !m->have_striped || use_device(m->device))
        return -1;
    const spmv_dev_striped &d = m->ds;
    const size_t ntp = (size_t)d.n_ranges * ((size_t)d.GM + 1);
    int32_t *tp = (int32_t *)malloc(ntp * 4), *wgd = (int32_t *)malloc((size_t)d.n_wg * 16), *rr = (int32_t *)malloc((size_t)d.n_ranges * 8);
    if (!tp || !wgd || !rr || hipMemcpy(tp, d.tptr, ntp * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(wgd, d.wgd, (size_t)d.n_wg * 16, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(rr, d.rrow, (size_t)d.n_ranges * 8, hipMemcpyDeviceToHost) != hipSuccess) {
        free(tp), free(wgd), free(rr);
        return -1;
    }
    const int n = d.n_wg < max_wgs ? d.n_wg : max_wgs;
    for (int q = 0; q < n; ++q) {
        long long all = 0, longest = 0, rows = 0;
        for (int w = 0; w < d.WPB; ++w) {
            const size_t u = (size_t)q * d.WPB + w;
            const long long len = (long long)tp[u * (d.GM + 1) + d.GM] - tp[u * (d.GM + 1)];
            all += len;
            longest = len > longest ? len : longest;
            rows += rr[2 * u + 1] - rr[2 * u];
        }
        out[6 * q + 0] = wgd[4 * q], out[6 * q + 1] = wgd[4 * q + 1], out[6 * q + 2] = wgd[4 * q + 2];
        out[6 * q + 3] = all, out[6 * q + 4] = longest, out[6 * q + 5] = rows;
    }
    free(tp), free(wgd), free(rr);
    return n;
}
#endif

void spmv_hip_set_tuning(int key, int value)
{
    if (key == 2)
        spmv_tiled_set_wpb(value);
    else if (key == 3)
        spmv_tiled_set_xbytes(value);
    else if (key == 4)
        spmv_tiled_set_blocks(value);
    else if (key == 9 && value >= 0)
        g_stripe_wgs = value;
    else if (key == 11 && value >= 0)
        g_stripe_groups = value;
    else if (key == 12 && value >= 64 && value <= SPMV_STRIPE_WAVE_ROWS && value % 64 == 0)
        g_stripe_wave_rows = value;
    else if (key == 19 && value >= -200 && value <= 200)
        g_stripe_xcd_bias = value; /* STRIPED: per mille of weight moved from even-XCD workgroups to odd-XCD ones */
    else if (key == 18)
        spmvk_blocked_set_bin_rows(value); /* BLOCKED: rows per bin of layouts built from now on */
    else if (key == 21)
        spmvk_hub_set_min_cover(value); /* HUB: per cent of the entries that must hit a hub column for AUTO to keep the plan */
    else if (key == 23)
        spmvk_hub_set_lds_cap(value); /* HUB: at most this many columns in the LDS table of layouts built from now on (< 0: what fits) */
    else if (key == 22)
        spmvk_hub_set_table_bytes(value); /* HUB: bytes of the dense second-tier table in memory of layouts built from now on (0: none) */
    else if (key == 17 && value >= 0 && value <= 2)
        g_striped_builder = value; /* 0: the device builds the STRIPED layout (the host when it must), 1: host, 2: device or fail */
    else if (key == 15 && value >= 0 && value <= 2)
        g_stripe_shape = value; /* 0: chosen per matrix, 1: 16 waves x 128 KiB, 2: 8 waves x 64 KiB */
#ifdef SPMV_HIP_DEBUG_BUILD
    else if (key == 26)
        spmvk_blocked_set_abl(value); /* BLOCKED: timing-only probes of the byte budget (results are wrong) */
    else if (key == 14)
        g_fail_relayout_alloc = value != 0;
#endif
    else
        spmvk_set_tuning(key, value);
}

} // extern "C"
