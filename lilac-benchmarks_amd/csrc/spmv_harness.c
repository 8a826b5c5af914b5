/*
 * csrc/spmv_harness.c -- the C host side of libhip-spmv.so: the two ABI
 * symbols and the policy around them.  Plain C; every HIP call lives behind
 * the thin layer in spmv_device.hip.
 *
 * Mirrors the operator interface of the reference backends, same names, same
 * argument meaning, same error behaviour:
 *
 *   spmv_harness_    /root/reference/libspmv/native.c:3-6   (gpu.c:211-289)
 *   f_spmv_harness_  /root/reference/libspmv/native.c:8-11  (gpu.c:291-369)
 *
 * What the reference GPU backend does and where this file does it differently:
 *
 *   device-copy cache keyed on the host pointers (gpu.c:227-234)
 *       -> kept, but several matrices can be cached at once (SPMV_HIP_CACHE_SLOTS)
 *          and double/float copies are independent (gpu.c shares d_csrRowPtrA /
 *          d_csrColIndA between setup() and f_setup(), gpu.c:26-27,122-123).
 *   write detection by mprotect + SIGSEGV (gpu.c:162-209,245-261)
 *       -> kept as the default policy, with the defect of SURVEY.md 5.2(d)
 *          removed: only pages that lie ENTIRELY inside a matrix array are
 *          protected; the partial first/last page of each array (which Fortran
 *          COMMON blocks share with the CG vectors, NPB3.3.1/CG/cg.f:69-80) is
 *          instead compared byte-for-byte against a saved copy on every call.
 *          A write to a neighbour variable therefore never faults and never
 *          disarms detection.
 *       -> SPMV_HIP_CACHE=pointer reproduces gpu-unsafe.c:86-127 (pointer
 *          equality only), SPMV_HIP_CACHE=off reproduces gpu-slow.c:155-163
 *          (upload on every call).
 *   x H2D / csrmv / y D2H on every call (gpu.c:264-285)
 *       -> same traffic, but ordered on ONE stream (gpu.c:41 vs :264,285 races,
 *          SURVEY.md 5.2(a)), through pinned staging buffers.
 *   errors: assert -> abort (gpu.c:42-58)
 *       -> message on stderr + abort().  There is no CPU fallback: without a
 *          GPU the first call aborts (tiny products run on the host by choice,
 *          see SPMV_HIP_DISPATCH, but only on a machine that has the GPU).
 *
 * Environment (read once, at the first call):
 *   SPMV_HIP_CACHE   = mprotect (default) | compare | pointer | off
 *                      mprotect: write detection by page protection for arrays of >= 64 whole
 *                      pages, byte comparison against a saved copy for smaller ones and for the
 *                      partial first/last pages.  Like libspmv/gpu.c it leaves the caller's matrix
 *                      pages read-only while the copy is cached: a caller that FREES a cached
 *                      matrix should call spmv_hip_release() first (or use compare/pointer/off),
 *                      because recycled heap pages would still be read-only for system calls.
 *                      compare: no page protection, every call compares the arrays byte for byte.
 *   SPMV_HIP_ZEROCOPY = 1 (default) | 0     vectors up to 1 MB cross PCIe through kernels
 *                                           reading / writing pinned host memory instead of DMA
 *   SPMV_HIP_YOVERLAP = 1 (default) | 0     a y of 4 MB or more leaves in row chunks while later chunks compute
 *                                           (STREAM / VECTOR plans, one part)
 *   SPMV_HIP_PLAN    = auto (default) | ordered | stream | vector[:L] | binned | tiled | striped | blocked
 *   SPMV_HIP_DEVICES = number of GPUs to row-partition over (default 1; "all")
 *   SPMV_HIP_PARTS   = row partitions per call (default = devices); more parts
 *                      than devices wraps around -- lets the multi-GPU code path
 *                      run on one GPU
 *   SPMV_HIP_PIN     = auto (default) | staging | register | pageable
 *                      auto: vectors with static storage duration (Fortran COMMON, C globals --
 *                      recognised with dladdr) are pinned in place once, everything else goes
 *                      through pinned staging buffers; register: pin whatever the caller passes
 *                      (only safe if the caller never frees it); pageable: plain hipMemcpy
 *   SPMV_HIP_DISPATCH = auto (default) | calibrated | device | host
 *                      auto: a product the host finishes sooner than a PCIe round trip goes to a plain sequential loop in
 *                      this library (dispatch_host below; precedent: libspmv/mkl.c:30, mkl-model.c:152-165), and the HIP
 *                      context is not created before a matrix is worth it -- a pure function of the shapes multiplied so
 *                      far, so the same run makes the same choices on every host; calibrated: the same rule with the host
 *                      loop's rate (measured on the products it runs anyway) and the PCIe round trip (measured when the
 *                      context is created) of THIS machine instead of the built-in constants, at the price of choices that
 *                      can differ from run to run near the boundary; device / host force one side.  A machine
 *                      without a GPU aborts at the first call whatever this says: the host loop is a dispatch target on
 *                      a GPU machine, not a fallback.
 *   SPMV_HIP_STATS   = 1: summary on stderr at exit; 2: also synchronise
 *                      between the phases of every call to attribute time
 *   SPMV_HIP_VERBOSE = 1: log uploads / invalidations
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <pthread.h>
#include <signal.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include "spmv_harness.h"
#include "spmv_internal.h"

#define MAX_SLOTS 8
#define MAX_PARTS 16
#define MAX_REG 32

enum { CACHE_MPROTECT = 0, CACHE_POINTER = 1, CACHE_OFF = 2, CACHE_COMPARE = 3 };
#define MIN_PROTECT_PAGES 64 /* arrays with fewer whole pages are compared byte for byte instead */
enum { PIN_STAGING = 0, PIN_REGISTER = 1, PIN_PAGEABLE = 2, PIN_AUTO = 3 };
enum { DISPATCH_AUTO = 0, DISPATCH_DEVICE = 1, DISPATCH_HOST = 2 };

/* one guarded host array: [p, p+bytes) */
typedef struct guard {
    const char *p;
    size_t bytes;
    char *lo, *hi;           /* protected interior pages [lo,hi), lo==hi when none     */
    size_t head_n, tail_n;   /* bytes of the partial first / last page                  */
    unsigned char *head, *tail; /* saved copies of those bytes                            */
    volatile sig_atomic_t armed; /* [lo,hi) is read-only and a write into it faults       */
    volatile sig_atomic_t dirty; /* 0, GUARD_WRITTEN or GUARD_EXPOSED                      */
} guard;
/* why a guard no longer vouches for its array: a write faulted in it, or its pages went writable together with a
 * neighbour's (another cached matrix guards the same pages) and what happened to them since is unknown */
enum { GUARD_WRITTEN = 1, GUARD_EXPOSED = 2 };

typedef struct slot {
    int used;
    int elem;
    const void *a;
    const int *rowstr;
    const int *colidx;
    int rows;
    int first;               /* rowstr[0] at upload time                                */
    int64_t nnz;
    int cols;
    int nparts;
    int bounds[MAX_PARTS + 1];
    spmv_hip_matrix *part[MAX_PARTS];
    int part_dev[MAX_PARTS];
    void *dx[MAX_PARTS];     /* device x per part (cols elements)                       */
    void *dy[MAX_PARTS];     /* device y per part (rows of the part)                    */
    guard g[3];              /* a, rowstr, colidx                                        */
    int uncacheable;         /* memory that cannot be protected: uploaded on every call  */
    int protect;             /* how its arrays are watched: 1 page protection, 2 saved copies, 0 not at all (pointer policy) */
    uint64_t last_use;
    int64_t uses;            /* products this copy has served                            */
    /* the values of a cached matrix may change while its sparsity pattern stays (PATHSAMPLE's PFOLD rewrites the
     * array in the `a` position before every call, Pfold.f90:219-223): see refresh_values() */
    int64_t uses_since_refresh; /* products since the values were last uploaded                          */
    int values_volatile;     /* the values change every call: no guard on `a`, they are uploaded each call      */
    int probing;             /* `a` was re-armed to find out whether it still changes                           */
    int64_t volatile_calls, probe_every;
} slot;

static struct {
    int inited;      /* environment read, GPU presence checked                        */
    int dev_inited;  /* HIP runtime initialised, device context exists                 */
    int dispatch;    /* DISPATCH_AUTO / _DEVICE / _HOST                                */
    double host_spent; /* seconds this process has spent in host products so far       */
    int cache, pin, stats, verbose, zero_copy, y_overlap;
    int plan, plan_param;
    int ndev, nparts, nslots;
    long page;
    slot slots[MAX_SLOTS];
    uint64_t tick;
    slot *last;
    int calibrated;      /* SPMV_HIP_DISPATCH=calibrated: host rate and PCIe round trip measured, not assumed */
    double host_ns_per_nnz, pcie_rtt_s;
    /* pinned staging */
    void *xstage, *ystage;
    size_t xstage_n, ystage_n;
    /* registered caller ranges (SPMV_HIP_PIN=register) */
    struct {
        char *lo, *hi;
    } reg[MAX_REG];
    int nreg;
    /* SIGSEGV chaining */
    int handler_installed;
    struct sigaction old_sa;
    spmv_hip_stats st;
} G;

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

/*
 * Matrices that keep changing (the caller rewrites an array between calls, as PATHSAMPLE's argument order does,
 * Pfold.f90:219-223).  A device copy that is caught changed after serving fewer than SHORT_LIVED_USES products was
 * "short-lived"; two short-lived copies of the same three pointers in a row make the matrix CHURNING, and then
 *   - its uploads stop asking for a re-laid-out plan (STRIPED costs 0.1 s per upload of a class-C-sized matrix and pays
 *     off only over hundreds of products);
 *   - its arrays are no longer page-protected but compared byte for byte: the runtime's copy leaves the caller's pages
 *     registered with the GPU driver, and every mprotect() of registered pages makes the driver evict and restore the
 *     process's queues -- 30 ms for the next copy and 30 ms for the next free, measured with the PATHSAMPLE counterpart
 *     (profiles/r02_pathsample_driver.log: 60 ms per call, against 5 ms with byte comparison);
 *   - under SPMV_HIP_DISPATCH=auto the product runs on the host loop whenever that is faster than another upload; every
 *     CHURN_PROBE_CALLS-th call tries the device again, so a matrix that has stopped changing gets its device copy back.
 * A copy that serves SHORT_LIVED_USES products or more ends the churning.  Forgotten at spmv_hip_release().
 */
#define MAX_CHURN 16
#define SHORT_LIVED_USES 3
#define CHURN_PROBE_CALLS 64
typedef struct churn_entry {
    const void *a, *rowstr, *colidx;
    int short_lived;      /* consecutive short-lived copies                                     */
    int host_streak;      /* host products since the last upload (auto dispatch)                */
    double last_upload_s; /* what the last upload of these arrays cost                          */
} churn_entry;
static churn_entry g_churn[MAX_CHURN];
static int g_nchurn;

/* matrices that have not been worth an upload yet (dispatch_host) */
#define MAX_PENDING 8
typedef struct pending_entry {
    const void *a, *rowstr, *colidx;
    int rows;
    double spent;
} pending_entry;
static pending_entry g_pending[MAX_PENDING];
static int g_pending_next;

static churn_entry *churn_find(const void *a, const void *rowstr, const void *colidx, int create)
{
    for (int i = 0; i < g_nchurn; ++i)
        if (g_churn[i].a == a && g_churn[i].rowstr == rowstr && g_churn[i].colidx == colidx)
            return &g_churn[i];
    if (!create)
        return NULL;
    const int i = g_nchurn < MAX_CHURN ? g_nchurn++ : 0; /* full: recycle the first entry */
    memset(&g_churn[i], 0, sizeof g_churn[i]);
    g_churn[i].a = a, g_churn[i].rowstr = rowstr, g_churn[i].colidx = colidx;
    return &g_churn[i];
}

static int churning(const void *a, const void *rowstr, const void *colidx)
{
    const churn_entry *e = churn_find(a, rowstr, colidx, 0);
    return e && e->short_lived >= 2;
}

static void die(const char *fmt, ...)
{
    va_list ap;
    fprintf(stderr, "libhip-spmv: fatal: ");
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fprintf(stderr, "\n");
    abort(); /* the reference's convention: assert() -> abort(), libspmv/gpu.c:42-58 */
}

#define CHECK(call)                                                        \
    do {                                                                   \
        if ((call) != 0)                                                   \
            die("%s: %s", #call, spmv_hip_last_error());                   \
    } while (0)

/* ---------------------------------------------------------------- write detection */

static void guard_unprotect(guard *g)
{
    if (g->hi > g->lo)
        mprotect(g->lo, (size_t)(g->hi - g->lo), PROT_READ | PROT_WRITE);
}

/*
 * Takes the page protection off `first` (a write faulted in it, or its slot is going away).  Cached matrices may share
 * host pages (spmv_harness_ and f_spmv_harness_ on one rowstr/colidx, two matrices with one sparsity pattern, the same
 * arrays with another `rows`): making `first`'s pages writable again silently takes the write detection away from every
 * other armed guard over one of them, so those are marked GUARD_EXPOSED (their slots upload again at their next call
 * and re-arm), and so on until no armed guard overlaps a disarmed one.  Guards, not slots, are the unit: a write into
 * the values of a matrix leaves the guards of its rowstr / colidx -- and every other matrix that shares only those --
 * armed.  Async-signal-safe: mprotect, pointer compares and plain stores only.
 */
static void disarm_cascade(guard *first, int why)
{
    first->dirty = why;
    first->armed = 0;
    for (int changed = 1; changed;) {
        changed = 0;
        for (int t = 0; t < MAX_SLOTS * 3; ++t) {
            guard *gt = &G.slots[t / 3].g[t % 3];
            if (!gt->armed)
                continue;
            for (int u = 0; u < MAX_SLOTS * 3; ++u) {
                const guard *gu = &G.slots[u / 3].g[u % 3];
                if (gu != gt && !gu->armed && gu->dirty && gu->hi > gu->lo && gt->lo < gu->hi && gu->lo < gt->hi) {
                    gt->dirty = GUARD_EXPOSED;
                    gt->armed = 0;
                    changed = 1;
                    break;
                }
            }
        }
    }
    for (int t = 0; t < MAX_SLOTS * 3; ++t) {
        guard *gt = &G.slots[t / 3].g[t % 3];
        if (gt->armed || !gt->dirty || gt->hi <= gt->lo)
            continue;
        guard_unprotect(gt);
        gt->hi = gt->lo; /* nothing protected any more: later cascades must not see it */
    }
}

/* async-signal-safe: only mprotect and stores to sig_atomic_t / guard fields */
static void on_segv(int sig, siginfo_t *si, void *ctx)
{
    char *addr = (char *)si->si_addr;
    for (int t = 0; t < MAX_SLOTS * 3; ++t) {
        guard *g = &G.slots[t / 3].g[t % 3];
        if (g->armed && addr >= g->lo && addr < g->hi) {
            /* the caller is editing a cached matrix: let the write through and refresh the device copy at the next
             * call (gpu.c:162-183) -- of every slot that guards these pages */
            disarm_cascade(g, GUARD_WRITTEN);
            return;
        }
    }
    /* not ours: chain (gpu.c:180-182) */
    if (G.old_sa.sa_flags & SA_SIGINFO) {
        if (G.old_sa.sa_sigaction) {
            G.old_sa.sa_sigaction(sig, si, ctx);
            return;
        }
    } else if (G.old_sa.sa_handler != SIG_DFL && G.old_sa.sa_handler != SIG_IGN) {
        G.old_sa.sa_handler(sig);
        return;
    }
    /* default action: re-raise with the default disposition */
    signal(SIGSEGV, SIG_DFL);
}

static void install_handler(void)
{
    if (G.handler_installed)
        return;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_flags = SA_SIGINFO;
    sigemptyset(&sa.sa_mask);
    sa.sa_sigaction = on_segv;
    if (sigaction(SIGSEGV, &sa, &G.old_sa) == 0)
        G.handler_installed = 1;
}

/* the guard must not be armed any more (disarm_cascade) */
static void guard_free(guard *g)
{
    free(g->head);
    free(g->tail);
    memset(g, 0, sizeof *g);
}

/* Splits [p,p+bytes) into protected interior pages and saved partial pages. */
static int guard_arm(guard *g, const void *p, size_t bytes, int protect)
{
    memset(g, 0, sizeof *g);
    g->p = (const char *)p;
    g->bytes = bytes;
    if (bytes == 0)
        return 0;
    const uintptr_t pg = (uintptr_t)G.page;
    uintptr_t b = (uintptr_t)p, e = b + bytes;
    uintptr_t lo = (b + pg - 1) / pg * pg; /* first page boundary >= b */
    uintptr_t hi = e / pg * pg;            /* last page boundary  <= e */
    if (protect == 2 || (protect && hi < lo + (uintptr_t)MIN_PROTECT_PAGES * pg)) {
        /* CACHE_COMPARE, or a small array: keep a full copy and compare it on every call.  Small
         * arrays usually live in the malloc heap; leaving heap pages read-only after the caller
         * frees them would make a later read(2) into the recycled memory fail with EFAULT. */
        g->head_n = bytes;
        g->head = (unsigned char *)malloc(bytes);
        if (!g->head)
            return -1;
        memcpy(g->head, p, bytes);
        g->lo = g->hi = (char *)e;
        return 0;
    }
    if (!protect || hi <= lo) {
        /* no whole page inside: the array is covered by saved copies only */
        lo = hi = e;
        if (!protect) { /* pointer policy: nothing saved, nothing protected */
            g->lo = g->hi = (char *)e;
            return 0;
        }
        g->head_n = bytes;
    } else {
        g->head_n = lo - b;
        g->tail_n = e - hi;
    }
    if (g->head_n) {
        g->head = (unsigned char *)malloc(g->head_n);
        if (!g->head)
            return -1;
        memcpy(g->head, (const void *)b, g->head_n);
    }
    if (g->tail_n) {
        g->tail = (unsigned char *)malloc(g->tail_n);
        if (!g->tail)
            return -1;
        memcpy(g->tail, (const void *)hi, g->tail_n);
    }
    g->lo = (char *)lo;
    g->hi = (char *)hi;
    if (hi > lo) {
        g->armed = 1; /* before protecting: a fault may come at once */
        if (mprotect((void *)lo, hi - lo, PROT_READ) != 0) {
            g->armed = 0;
            g->lo = g->hi = (char *)e;
            return -2; /* cannot protect this memory */
        }
    }
    return 0;
}

static int guard_fragments_changed(const guard *g)
{
    if (g->head_n && memcmp(g->head, g->p, g->head_n) != 0)
        return 1;
    if (g->tail_n && memcmp(g->tail, g->p + g->bytes - g->tail_n, g->tail_n) != 0)
        return 1;
    return 0;
}

/* ---------------------------------------------------------------- slots */

static void slot_drop(slot *sl)
{
    if (!sl->used)
        return;
    for (int k = 0; k < 3; ++k) {
        if (sl->g[k].armed)
            disarm_cascade(&sl->g[k], GUARD_WRITTEN); /* guards of other slots over its pages lose their protection with it */
        guard_free(&sl->g[k]);
    }
    for (int p = 0; p < sl->nparts; ++p) {
        spmv_hip_matrix_destroy(sl->part[p]);
        spmv_hip_free(sl->part_dev[p], sl->dx[p]);
        spmv_hip_free(sl->part_dev[p], sl->dy[p]);
    }
    if (G.last == sl)
        G.last = NULL;
    memset(sl, 0, sizeof *sl);
}

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    if (!v || !*v)
        return dflt;
    return atoi(v);
}

static void print_stats(void)
{
    const spmv_hip_stats *s = &G.st;
    fprintf(stderr,
            "libhip-spmv: calls=%lld (host %lld) uploads=%lld value_refreshes=%lld invalidations=%lld init=%.3fs upload=%.3fs "
            "x_h2d=%.3fs kernel=%.3fs y_d2h=%.3fs host=%.3fs total=%.3fs pinned_in_place=%d\n",
            (long long)s->calls, (long long)s->host_calls, (long long)s->uploads, (long long)s->value_refreshes,
            (long long)s->invalidations,
            s->t_init, s->t_upload, s->t_x_h2d, s->t_kernel, s->t_y_d2h, s->t_host, s->t_total,
            G.nreg /* caller vectors pinned where they lie: static storage (COMMON blocks, C globals) or SPMV_HIP_PIN=register */);
}

void spmv_hip_release(void)
{
    pthread_mutex_lock(&g_mu);
    for (int s = 0; s < MAX_SLOTS; ++s)
        slot_drop(&G.slots[s]);
    for (int r = 0; r < G.nreg; ++r)
        spmvd_host_unregister(G.reg[r].lo);
    G.nreg = 0;
    g_nchurn = 0;
    memset(g_pending, 0, sizeof g_pending);
    spmvd_host_free(G.xstage);
    spmvd_host_free(G.ystage);
    G.xstage = G.ystage = NULL;
    G.xstage_n = G.ystage_n = 0;
    if (G.handler_installed) {
        struct sigaction cur;
        if (sigaction(SIGSEGV, NULL, &cur) == 0 && (cur.sa_flags & SA_SIGINFO) &&
            cur.sa_sigaction == on_segv)
            sigaction(SIGSEGV, &G.old_sa, NULL);
        G.handler_installed = 0;
    }
    pthread_mutex_unlock(&g_mu);
}

/*
 * Teardown.  Registered with atexit() from lazy_init (a process that only ever used the host loop still
 * prints its statistics) and AGAIN from device_init, i.e. after the HIP runtime has initialised, so that it
 * runs BEFORE the runtime's own exit handlers (handlers run in reverse order of registration) and, being
 * registered from inside this DSO, also when the library is dlclose()d (pagerank does that,
 * /root/reference/pagerank/main.cpp:25-30).  Idempotent.
 */
static void lib_fini(void)
{
    if (!G.inited)
        return;
    if (G.stats)
        print_stats();
    spmv_hip_release();
    G.inited = 0;
}

/* Is there a GPU at all?  Answered without touching the HIP runtime (whose start-up is what small callers
 * should not pay): the compute driver's device node.  The authoritative check is device_init()'s. */
static int gpu_node_present(void)
{
#ifdef SPMV_TESTING /* the mock-device builds of tests/native have no GPU to find: the test names the node to look for */
    if (getenv("SPMV_HIP_KFD_NODE"))
        return access(getenv("SPMV_HIP_KFD_NODE"), R_OK | W_OK) == 0;
#endif
    /* a device list emptied on purpose (HIP_VISIBLE_DEVICES= / ROCR_VISIBLE_DEVICES=, or -1) hides every GPU from the
     * runtime: small products must not run on the host loop in a process whose first large one would then abort */
    static const char *const lists[] = {"HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"};
    for (size_t i = 0; i < sizeof lists / sizeof lists[0]; ++i) {
        const char *v = getenv(lists[i]);
        if (v && (!*v || !strcmp(v, "-1")))
            return 0;
    }
    return access("/dev/kfd", R_OK | W_OK) == 0;
}

/* first call: the environment, and the refusal to run on a machine without a GPU */
static void lazy_init(void)
{
    if (G.inited)
        return;
    G.page = sysconf(_SC_PAGESIZE);
    const char *v;

    G.cache = CACHE_MPROTECT;
    if ((v = getenv("SPMV_HIP_CACHE"))) {
        if (!strcmp(v, "pointer") || !strcmp(v, "unsafe"))
            G.cache = CACHE_POINTER;
        else if (!strcmp(v, "off") || !strcmp(v, "slow"))
            G.cache = CACHE_OFF;
        else if (!strcmp(v, "compare"))
            G.cache = CACHE_COMPARE;
        else if (strcmp(v, "mprotect"))
            die("SPMV_HIP_CACHE=%s: expected mprotect|compare|pointer|off", v);
    }
    G.pin = PIN_AUTO;
    if ((v = getenv("SPMV_HIP_PIN"))) {
        if (!strcmp(v, "register"))
            G.pin = PIN_REGISTER;
        else if (!strcmp(v, "pageable"))
            G.pin = PIN_PAGEABLE;
        else if (!strcmp(v, "staging"))
            G.pin = PIN_STAGING;
        else if (strcmp(v, "auto"))
            die("SPMV_HIP_PIN=%s: expected auto|staging|register|pageable", v);
    }
    G.plan = SPMV_HIP_PLAN_AUTO;
    G.plan_param = 0;
    if ((v = getenv("SPMV_HIP_PLAN"))) {
        if (!strcmp(v, "ordered"))
            G.plan = SPMV_HIP_PLAN_ORDERED;
        else if (!strcmp(v, "stream"))
            G.plan = SPMV_HIP_PLAN_STREAM;
        else if (!strncmp(v, "vector", 6)) {
            G.plan = SPMV_HIP_PLAN_VECTOR;
            G.plan_param = v[6] == ':' ? atoi(v + 7) : 64;
        } else if (!strcmp(v, "binned"))
            G.plan = SPMV_HIP_PLAN_BINNED;
        else if (!strcmp(v, "tiled"))
            G.plan = SPMV_HIP_PLAN_TILED;
        else if (!strcmp(v, "striped"))
            G.plan = SPMV_HIP_PLAN_STRIPED;
        else if (!strcmp(v, "blocked"))
            G.plan = SPMV_HIP_PLAN_BLOCKED;
        else if (!strcmp(v, "hub"))
            G.plan = SPMV_HIP_PLAN_HUB;
        else if (strcmp(v, "auto"))
            die("SPMV_HIP_PLAN=%s: expected auto|ordered|stream|vector[:L]|binned|tiled|striped|blocked|hub", v);
    }
    G.dispatch = DISPATCH_AUTO;
    if ((v = getenv("SPMV_HIP_DISPATCH"))) {
        if (!strcmp(v, "device"))
            G.dispatch = DISPATCH_DEVICE;
        else if (!strcmp(v, "host"))
            G.dispatch = DISPATCH_HOST;
        else if (!strcmp(v, "calibrated"))
            G.calibrated = 1; /* auto, with this host's loop rate and this box's PCIe round trip measured */
        else if (strcmp(v, "auto"))
            die("SPMV_HIP_DISPATCH=%s: expected auto|calibrated|device|host", v);
    }
    if (G.dispatch == DISPATCH_AUTO && (G.plan != SPMV_HIP_PLAN_AUTO || getenv("SPMV_HIP_PARTS") || getenv("SPMV_HIP_DEVICES")))
        G.dispatch = DISPATCH_DEVICE; /* who names a kernel or a row partition wants the device */
    G.zero_copy = env_int("SPMV_HIP_ZEROCOPY", 1);
    G.y_overlap = env_int("SPMV_HIP_YOVERLAP", 1);
    G.stats = env_int("SPMV_HIP_STATS", 0);
    G.verbose = env_int("SPMV_HIP_VERBOSE", 0);
    G.nslots = env_int("SPMV_HIP_CACHE_SLOTS", 4);
    if (G.nslots < 1)
        G.nslots = 1;
    if (G.nslots > MAX_SLOTS)
        G.nslots = MAX_SLOTS;
    if (!gpu_node_present())
        die("no HIP device is visible (this backend has no CPU fallback)");
    atexit(lib_fini);
    G.inited = 1;
}

/* first product that goes to the device: the HIP runtime, the context, the code object */
static void device_init(void)
{
    if (G.dev_inited)
        return;
    double t0 = spmv_now();
    const char *v;
    int visible = spmv_hip_device_count();
    if (visible <= 0)
        die("no HIP device is visible (this backend has no CPU fallback)");
    G.ndev = 1;
    if ((v = getenv("SPMV_HIP_DEVICES")))
        G.ndev = !strcmp(v, "all") ? visible : atoi(v);
    if (G.ndev < 1 || G.ndev > visible)
        die("SPMV_HIP_DEVICES=%d but %d device(s) visible", G.ndev, visible);
    if (G.ndev > MAX_PARTS)
        G.ndev = MAX_PARTS;
    G.nparts = env_int("SPMV_HIP_PARTS", G.ndev);
    if (G.nparts < 1 || G.nparts > MAX_PARTS)
        die("SPMV_HIP_PARTS must be in [1,%d]", MAX_PARTS);

    /* touch device 0 now so that the context / code-object load is attributed to init */
    void *probe = spmv_hip_malloc(0, 256);
    if (!probe)
        die("device initialisation failed: %s", spmv_hip_last_error());
    spmv_hip_free(0, probe);
    /* again, now that the HIP runtime has registered its own exit handlers: handlers run in reverse order of
     * registration, so this one frees the device copies BEFORE the runtime goes away (lib_fini runs once) */
    atexit(lib_fini);
    G.dev_inited = 1;
    if (G.calibrated) {
        /* the PCIe round trip of a small vector on THIS box: 4 KB down, 4 KB back, through pinned memory; the best of 8 */
        void *h = spmvd_host_alloc(8192), *d = spmv_hip_malloc(0, 8192);
        if (h && d) {
            memset(h, 0, 8192);
            double best = 1.0;
            for (int i = 0; i < 8; ++i) {
                const double t = spmv_now();
                if (spmvd_copy_h2d_async(0, d, h, 4096) || spmvd_copy_d2h_async(0, (char *)h + 4096, d, 4096) || spmvd_stream_sync(0))
                    break;
                const double dt = spmv_now() - t;
                best = dt < best ? dt : best;
            }
            if (best < 1.0)
                G.pcie_rtt_s = best;
        }
        spmvd_host_free(h);
        spmv_hip_free(0, d);
    }
    G.st.t_init = spmv_now() - t0;
}

/* ---------------------------------------------------------------- host / device dispatch */

/*
 * The arithmetic of libspmv/native-impl.c:1-25 as it stands there: one accumulator, the product rounded
 * before the add (this file is compiled with -ffp-contract=off), left to right.  Results are bit-identical to
 * the reference's CPU backend.  It exists because a product of a few thousand nonzeros takes the host a few
 * microseconds and a PCIe round trip twenty (profiles/r01_xferbench_pcie_round_trip.log), and because creating
 * the HIP context costs 0.2 s, more than the whole run of parboil's or a small graph's caller
 * (profiles/r01_reference_programs_native_vs_hip.log): the reference's MKL backend makes the same per-call
 * choice between its native loop and the library (libspmv/mkl.c:30, mkl-model.c:152-165).
 */
static void host_spmv_f64(double *ov, const double *a, const double *iv, const int *rowstr, const int *colidx, int rows)
{
    for (int i = 0; i < rows; ++i) {
        double value = 0.0;
        for (int j = rowstr[i] - 1; j < rowstr[i + 1] - 1; ++j)
            value += a[j] * iv[colidx[j] - 1];
        ov[i] = value;
    }
}

static void host_spmv_f32(float *ov, const float *a, const float *iv, const int *rowstr, const int *colidx, int rows)
{
    for (int i = 0; i < rows; ++i) {
        float value = 0.0f;
        for (int j = rowstr[i] - 1; j < rowstr[i + 1] - 1; ++j)
            value += a[j] * iv[colidx[j] - 1];
        ov[i] = value;
    }
}

/* seconds: the host loop multiplies a nonzero in about 0.35 ns while the matrix stays in the host's caches (parboil
 * `medium`, 149 090 nonzeros: 56 us per product; NPB class W, 508 402: 0.16 ms) and in about 1 ns when it streams from
 * memory (native.c on the GPU box's EPYC: 2.1 GFLOP/s on class C; PATHSAMPLE's 2.6 M nonzeros: 1.2 ns).  A function of
 * the shape alone: see host_product on why the dispatch must not depend on measured time. */
#define HOST_CACHED_BYTES ((int64_t)16 << 20)
static double host_estimate(int rows, int64_t nnz)
{
    double ns = nnz * 12 <= HOST_CACHED_BYTES ? 0.35 : 1.0;
    if (G.calibrated && G.host_ns_per_nnz > 0.0)
        ns = G.host_ns_per_nnz; /* measured under =calibrated */
    return 1.0e-9 * ns * ((double)nnz + (double)rows) + 0.2e-6;
}

/* seconds for a device call once the matrix is cached: the PCIe round trip of x and y (19 us zero-copy under 1 MB, DMA
 * at ~25 GB/s each way above) plus a kernel that streams 12 bytes per nonzero at a few TB/s */
static double device_estimate(int rows, int64_t nnz, int elem)
{
    const double vec = 2.0 * (double)rows * elem; /* x and y; cols is not known before the upload: square */
    const double rtt = G.calibrated && G.pcie_rtt_s > 0.0 ? G.pcie_rtt_s : 19.0e-6;
    return rtt + (vec > 2.0e6 ? vec / 25.0e9 : 0.0) + 5.0e-6 + (double)nnz * (elem + 4) / 3.0e12;
}

/*
 * Before the HIP context exists, going to the device costs its creation on top (0.15-0.2 s): a process stays on the
 * host until it has spent about half of that there (the ski-rental rule: never more than ~1.5x the better choice),
 * unless the matrix is plainly device-class -- from about 1.5 M nonzeros a few dozen products repay the context (NPB
 * class A, 1.85 M: 416 calls, 0.21 s on the device against 0.55 s on the host).  Below that lie the callers whose whole
 * run is shorter than the context creation: parboil's datasets (0.07 / 2.9 ms for 50 products), BFS on a 2^16-vertex
 * graph (44 ms for 16 searches), NPB class S (16 ms).
 */
#define HOST_BUDGET_BEFORE_INIT 0.100 /* s of host products before a process pays for the HIP context              */
#define DEVICE_CLASS_SECONDS 1.5e-3   /* est. host time per product from which the device is taken at once          */

/*
 * The same rule per MATRIX (round 3): a matrix that is not cached yet costs an upload before its first device product --
 * allocations, the copies, validation, the plan, arming the write detection: about 10 ms plus its bytes at PCIe speed
 * (parboil `medium`, 149 090 nonzeros, inside a process whose context exists: 31.8 ms for 50 products on the device, 2.8 ms
 * on the host loop; profiles/r03_bench_line.json before this rule) -- so it stays on the host until it has spent that much
 * there, in ESTIMATED seconds like the budget above.  Device-class matrices skip it.
 */
static double upload_estimate(int64_t nnz, int elem) { return 10.0e-3 + (double)nnz * (elem + 4) / 20.0e9; }

static pending_entry *pending_find(const void *a, const void *rowstr, const void *colidx, int rows)
{
    for (int i = 0; i < MAX_PENDING; ++i)
        if (g_pending[i].a == a && g_pending[i].rowstr == rowstr && g_pending[i].colidx == colidx && g_pending[i].rows == rows)
            return &g_pending[i];
    pending_entry *e = &g_pending[g_pending_next++ % MAX_PENDING]; /* round robin: an old entry only loses its credit */
    e->a = a, e->rowstr = rowstr, e->colidx = colidx, e->rows = rows, e->spent = 0.0;
    return e;
}

/* 1: this call runs on the host.  `cached`: the matrix has a device copy (then only the steady-state comparison counts) */
static int dispatch_host(const void *a, const int *rowstr, const int *colidx, int rows, int64_t nnz, int elem, int cached)
{
    if (G.dispatch == DISPATCH_HOST)
        return 1;
    if (G.dispatch == DISPATCH_DEVICE)
        return 0;
    const double th = host_estimate(rows, nnz);
    if (th <= device_estimate(rows, nnz, elem))
        return 1; /* steady state: the host is done before x would have crossed PCIe */
    if (cached || th >= DEVICE_CLASS_SECONDS)
        return 0;
    pending_entry *e = pending_find(a, rowstr, colidx, rows);
    const int worth_upload = e->spent + th >= upload_estimate(nnz, elem);
    if (!worth_upload || (!G.dev_inited && G.host_spent + th < HOST_BUDGET_BEFORE_INIT)) {
        e->spent += th; /* host: this matrix has not been worth its upload yet, or the process not its context (0.2 s) */
        return 1;
    }
    return 0;
}

/* what crossing PCIe again costs the values of a cached matrix (pageable memory: ~25 GB/s plus the copy's set-up) */
static double refresh_estimate(int64_t nnz, int elem) { return 30.0e-6 + (double)nnz * elem / 25.0e9; }

/* the product on the host loop, accounted */
static void host_product(void *ov, const void *a, const void *iv, const int *rowstr, const int *colidx, int rows,
                         int64_t nnz, int elem, double t_in)
{
    if (elem == 8)
        host_spmv_f64((double *)ov, (const double *)a, (const double *)iv, rowstr, colidx, rows);
    else
        host_spmv_f32((float *)ov, (const float *)a, (const float *)iv, rowstr, colidx, rows);
    const double dt = spmv_now() - t_in;
    /*
     * What a process may spend on the host before it pays for the HIP context is charged in ESTIMATED seconds, a pure
     * function of the shapes it has multiplied so far (mkl-model.c:152-165 predicts from rows and nnz likewise): the call
     * at which a process moves from the host loop (the reference's summation order) to the kernels (another order) is
     * then the same in every run, whatever the host's speed or load.  SPMV_HIP_DISPATCH=calibrated charges measured time.
     */
    G.host_spent += G.calibrated ? dt : host_estimate(rows, nnz);
    if (G.calibrated && nnz >= 20000) { /* long enough for the clock: the host's own rate, smoothed */
        const double rate = 1e9 * dt / (double)(nnz + rows);
        G.host_ns_per_nnz = G.host_ns_per_nnz > 0.0 ? 0.75 * G.host_ns_per_nnz + 0.25 * rate : rate;
    }
    G.st.calls++, G.st.host_calls++;
    G.st.t_host += dt, G.st.t_total += dt;
}

static slot *slot_find(const void *a, const int *rowstr, const int *colidx, int rows, int64_t nnz,
                       int elem)
{
    for (int s = 0; s < G.nslots; ++s) {
        slot *sl = &G.slots[s];
        if (sl->used && sl->elem == elem && sl->a == a && sl->rowstr == rowstr && sl->colidx == colidx &&
            sl->rows == rows && sl->nnz == nnz)
            return sl;
    }
    return NULL;
}

static slot *slot_victim(void)
{
    slot *best = NULL;
    for (int s = 0; s < G.nslots; ++s) {
        slot *sl = &G.slots[s];
        if (!sl->used)
            return sl;
        if (!best || sl->last_use < best->last_use)
            best = sl;
    }
    slot_drop(best);
    return best;
}

typedef struct colmax_ctx {
    const int *c;
    int max;
} colmax_ctx;

static void colmax_range(int64_t lo, int64_t hi, void *v)
{
    colmax_ctx *cc = (colmax_ctx *)v;
    int mx = 0;
    for (int64_t j = lo; j < hi; ++j)
        mx = cc->c[j] > mx ? cc->c[j] : mx;
    int seen = __atomic_load_n(&cc->max, __ATOMIC_RELAXED);
    while (mx > seen && !__atomic_compare_exchange_n(&cc->max, &seen, mx, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED))
        ;
}

typedef struct upload_job {
    slot *sl;
    int p;
    const void *a;
    const int *rowstr, *colidx;
    int rows, elem, cols_hint, relayout, ok;
    char err[400];
    pthread_t th;
} upload_job;

/* uploads part p of the slot: device CSR block + plan, x and y buffers; errors are reported by the caller */
static void *upload_part(void *v)
{
    upload_job *j = (upload_job *)v;
    slot *sl = j->sl;
    const int p = j->p, dev = sl->part_dev[p];
    spmvd_set_relayout(j->relayout); /* a per-thread setting of the device layer */
    sl->part[p] = spmv_hip_matrix_create(j->a, j->rowstr, j->colidx, j->rows, j->elem, dev, sl->bounds[p],
                                         sl->bounds[p + 1], sl->nparts == 1 ? 0 : j->cols_hint);
    int ok = sl->part[p] != NULL;
    if (ok && G.plan != SPMV_HIP_PLAN_AUTO)
        ok = spmv_hip_matrix_set_plan(sl->part[p], G.plan, G.plan_param) == 0;
    if (ok) {
        spmv_hip_info inf;
        spmv_hip_matrix_info(sl->part[p], &inf);
        const size_t prow = (size_t)(sl->bounds[p + 1] - sl->bounds[p]);
        sl->dx[p] = spmv_hip_malloc(dev, ((size_t)inf.cols + 1) * j->elem);
        sl->dy[p] = spmv_hip_malloc(dev, (prow + 1) * j->elem);
        ok = sl->dx[p] && sl->dy[p];
    }
    if (!ok)
        snprintf(j->err, sizeof j->err, "%s", spmv_hip_last_error()); /* the message is per thread */
    j->ok = ok;
    return NULL;
}

/* the three guarded arrays of a slot: 0 = a, 1 = rowstr, 2 = colidx (the entries the matrix uses) */
static void slot_array(const slot *sl, int k, const void **p, size_t *bytes)
{
    const size_t off = (size_t)(sl->first - 1);
    if (k == 0)
        *p = (const char *)sl->a + off * (size_t)sl->elem, *bytes = (size_t)sl->nnz * (size_t)sl->elem;
    else if (k == 1)
        *p = sl->rowstr, *bytes = ((size_t)sl->rows + 1) * sizeof(int);
    else
        *p = sl->colidx + off, *bytes = (size_t)sl->nnz * sizeof(int);
}

/* protect: 1 page protection (+ saved partial pages), 2 a full saved copy compared on every call, 0 nothing */
static void slot_arm_guard(slot *sl, int k, int protect)
{
    const void *p;
    size_t bytes;
    slot_array(sl, k, &p, &bytes);
    if (protect == 1)
        install_handler();
    const int rc = guard_arm(&sl->g[k], p, bytes, protect);
    if (rc == -1)
        die("out of host memory arming write detection");
    if (rc == -2) {
        /* memory that cannot be mprotect'ed: be safe, fall back to upload-per-call for it */
        if (G.verbose)
            fprintf(stderr, "libhip-spmv: mprotect failed (%s); matrix will not be cached\n", strerror(errno));
        sl->uncacheable = 1;
        for (int j = 0; j < 3; ++j)
            if (sl->g[j].armed)
                disarm_cascade(&sl->g[j], GUARD_WRITTEN);
    }
}

static void slot_upload(slot *sl, const void *a, const int *rowstr, const int *colidx, int rows,
                        int64_t nnz, int elem)
{
    double t0 = spmv_now();
    memset(sl, 0, sizeof *sl);
    sl->used = 1;
    sl->elem = elem;
    sl->a = a;
    sl->rowstr = rowstr;
    sl->colidx = colidx;
    sl->rows = rows;
    sl->nnz = nnz;
    sl->first = rowstr[0];
    sl->nparts = G.nparts > rows ? (rows > 0 ? rows : 1) : G.nparts;
    spmv_hip_partition_rows(rowstr, rows, sl->nparts, sl->bounds);

    const int churn = G.cache == CACHE_OFF || churning(a, rowstr, colidx);
    /* the length of x: max(colidx) over the WHOLE matrix (libspmv/gpu.c:217-223); one part finds it on the device
     * while it uploads, several parts need it up front so that they can upload side by side */
    int cols = 0;
    if (sl->nparts > 1) {
        colmax_ctx cc = {colidx + (rowstr[0] - 1), 0};
        spmv_parallel_for(nnz, 1 << 20, colmax_range, &cc);
        cols = cc.max;
    }
    upload_job job[MAX_PARTS];
    for (int p = 0; p < sl->nparts; ++p) {
        sl->part_dev[p] = p % G.ndev;
        memset(&job[p], 0, sizeof job[p]);
        job[p].sl = sl, job[p].p = p, job[p].a = a, job[p].rowstr = rowstr, job[p].colidx = colidx;
        job[p].rows = rows, job[p].elem = elem, job[p].cols_hint = cols, job[p].relayout = !churn;
    }
    /*
     * One host thread per part: each uploads its row block to its device, re-lays it out and allocates its x / y
     * (class D over 8 parts: the conversions ran one after the other for about 4 s; side by side they take as long
     * as the slowest).  Part 0 runs on the calling thread.
     */
    int started = 0;
    for (int p = 1; p < sl->nparts; ++p) {
        if (pthread_create(&job[p].th, NULL, upload_part, &job[p]) != 0)
            break;
        started = p;
    }
    upload_part(&job[0]);
    for (int p = 1; p <= started; ++p)
        pthread_join(job[p].th, NULL);
    for (int p = started + 1; p < sl->nparts; ++p) /* threads that could not be started */
        upload_part(&job[p]);
    spmvd_set_relayout(1); /* the calling thread ran part 0 */
    for (int p = 0; p < sl->nparts; ++p)
        if (!job[p].ok)
            die("matrix upload failed (part %d of %d): %s", p, sl->nparts, job[p].err);
    if (sl->nparts == 1) {
        spmv_hip_info inf;
        spmv_hip_matrix_info(sl->part[0], &inf);
        cols = inf.cols;
    }
    sl->cols = cols;

    /* arm write detection AFTER the upload (the copies read the arrays) */
    if (G.cache != CACHE_OFF) {
        /* a churning matrix is compared, not protected: see the note at g_churn */
        sl->protect = G.cache == CACHE_MPROTECT ? (churn ? 2 : 1) : (G.cache == CACHE_COMPARE ? 2 : 0);
        for (int k = 0; k < 3 && !sl->uncacheable; ++k)
            slot_arm_guard(sl, k, sl->protect);
    }
    G.st.uploads++;
    G.st.t_upload += spmv_now() - t0;
    {
        churn_entry *e = churn_find(a, rowstr, colidx, 0);
        if (e)
            e->last_upload_s = spmv_now() - t0, e->host_streak = 0;
    }
    if (G.verbose) {
        spmv_hip_info inf;
        spmv_hip_matrix_info(sl->part[0], &inf);
        fprintf(stderr,
                "libhip-spmv: uploaded %s matrix rows=%d cols=%d nnz=%lld parts=%d plan=%d/%d "
                "(row len %d..%d, mean %.1f) in %.3f s\n",
                elem == 8 ? "f64" : "f32", rows, cols, (long long)nnz, sl->nparts, inf.plan,
                inf.plan_param, inf.min_row_len, inf.max_row_len, inf.mean_row_len, spmv_now() - t0);
    }
}

/*
 * The values of a cached matrix changed and nothing else did (its rowstr / colidx guards still vouch for them): the new
 * values cross PCIe into the device copy that is there -- no free, no allocation, no validation of the pattern, no new
 * plan (a re-laid-out plan re-packs its second copy on the device side).  What happens to the watch on `a`:
 *   - the copy had served SHORT_LIVED_USES products or more: the array is watched again as before;
 *   - it had not (PATHSAMPLE's PFOLD writes new values before every call, Pfold.f90:219-223): the values are VOLATILE --
 *     no watch at all (every mprotect() of pages the runtime has registered for its copies makes the driver evict and
 *     restore the process's queues, tens of milliseconds; a compare reads as many bytes as the copy moves), they are
 *     simply uploaded on every call; every probe_every-th call a copy of them is kept (CHURN_PROBE_CALLS, doubling up to
 *     4096 calls) and the next call compares: unchanged values end the volatility.
 */
static void refresh_values(slot *sl, const void *a, const int *colidx)
{
    const double t0 = spmv_now();
    const int short_lived = sl->uses_since_refresh < SHORT_LIVED_USES;
    if (!sl->values_volatile && (short_lived || sl->probing) && G.plan == SPMV_HIP_PLAN_AUTO)
        /* about to become volatile: a re-laid-out plan would pack its second copy again on every call (STRIPED costs
         * 0.1 s per class-C-sized matrix and pays off only over hundreds of products); the CSR kernels from here on */
        for (int p = 0; p < sl->nparts; ++p)
            if (spmv_hip_matrix_set_plan(sl->part[p], SPMV_HIP_PLAN_STREAM, 0) != 0)
                die("switching a volatile matrix to the CSR kernels failed: %s", spmv_hip_last_error());
    for (int p = 0; p < sl->nparts; ++p)
        if (spmv_hip_matrix_refresh_values(sl->part[p], a, colidx) != 0)
            die("refreshing the values of a cached matrix failed (part %d of %d): %s", p, sl->nparts, spmv_hip_last_error());
    sl->uses_since_refresh = 0;
    if (sl->values_volatile) {
        if (++sl->volatile_calls >= sl->probe_every) { /* look again: keep a copy, the next call compares */
            sl->values_volatile = 0;
            sl->probing = 1;
            slot_arm_guard(sl, 0, 2);
        }
    } else {
        /* a write that faulted has disarmed the guard already; an edit of the saved head / tail fragment alone has not:
         * the interior pages are still read-only and must go writable before the record of them is forgotten (a volatile
         * matrix never re-arms, and the caller's next interior write would find no guard to let it through) */
        if (sl->g[0].armed)
            disarm_cascade(&sl->g[0], GUARD_WRITTEN);
        guard_free(&sl->g[0]);
        if (short_lived || sl->probing) {
            sl->probe_every = sl->probing && sl->probe_every < 4096 ? 2 * sl->probe_every : (sl->probing ? 4096 : CHURN_PROBE_CALLS);
            sl->values_volatile = 1;
            sl->volatile_calls = 0;
            sl->probing = 0;
        } else {
            slot_arm_guard(sl, 0, sl->protect);
        }
    }
    G.st.value_refreshes++;
    G.st.t_upload += spmv_now() - t0;
    if (G.verbose)
        fprintf(stderr, "libhip-spmv: values refreshed in %.3f ms%s\n", 1e3 * (spmv_now() - t0),
                sl->values_volatile ? " (volatile: uploaded on every call)" : "");
}

/* ---------------------------------------------------------------- pinned transfers */

/* a long vector into / out of the pinned staging buffer on a few host threads (12 MB: 1.2 ms on one, 0.3 ms on six) */
typedef struct cpy_ctx {
    char *d;
    const char *s;
} cpy_ctx;
static void cpy_range(int64_t lo, int64_t hi, void *v)
{
    const cpy_ctx *c = (const cpy_ctx *)v;
    memcpy(c->d + lo, c->s + lo, (size_t)(hi - lo));
}
static void par_memcpy(void *d, const void *s, size_t n)
{
    if (n < ((size_t)4 << 20)) {
        memcpy(d, s, n);
        return;
    }
    cpy_ctx c = {(char *)d, (const char *)s};
    spmv_parallel_for((int64_t)n, (int64_t)2 << 20, cpy_range, &c);
}

static void *stage_buf(void **buf, size_t *cap, size_t need)
{
    if (*cap < need) {
        spmvd_host_free(*buf);
        size_t n = need + need / 4 + 4096;
        *buf = spmvd_host_alloc(n);
        if (!*buf)
            die("pinned staging allocation failed: %s", spmv_hip_last_error());
        *cap = n;
    }
    return *buf;
}

/* SPMV_HIP_PIN=register: pin the caller's vector in place (page-rounded), once. */
static int ensure_registered(const void *p, size_t bytes)
{
    char *lo = (char *)((uintptr_t)p / (uintptr_t)G.page * (uintptr_t)G.page);
    char *hi = (char *)(((uintptr_t)p + bytes + G.page - 1) / (uintptr_t)G.page * (uintptr_t)G.page);
    for (int r = 0; r < G.nreg; ++r)
        if (lo >= G.reg[r].lo && hi <= G.reg[r].hi)
            return 1;
    for (int r = 0; r < G.nreg; ++r)
        if (lo < G.reg[r].hi && hi > G.reg[r].lo)
            return 0; /* overlaps a registered range partially: use staging for this one */
    if (G.nreg == MAX_REG)
        return 0;
    if (spmvd_host_register(lo, (size_t)(hi - lo)) != 0)
        return 0;
    G.reg[G.nreg].lo = lo;
    G.reg[G.nreg].hi = hi;
    G.nreg++;
    return 1;
}

/*
 * SPMV_HIP_PIN=auto: a vector with static storage duration (a Fortran COMMON block such as
 * NPB's p/q/r/z, NPB3.3.1/CG/cg.f:69-80, or a C global) lives inside a loaded ELF object, which
 * dladdr() can tell; such memory is never freed and re-mapped, so pinning it in place is safe and
 * saves the two staging copies per call.  Heap and stack vectors are never pinned by us.
 */
static int has_static_storage(const void *p, size_t bytes)
{
    Dl_info a, b;
    if (bytes == 0 || !dladdr(p, &a) || !a.dli_fbase)
        return 0;
    if (!dladdr((const char *)p + bytes - 1, &b) || b.dli_fbase != a.dli_fbase)
        return 0;
    return 1;
}

/*
 * Which way a host vector crosses PCIe (measured on MI355X, profiles/r01_abi_transfer_modes.log):
 * pinned in place is the fastest and is what static vectors get; a heap or stack vector of a
 * megabyte or more is handed to the runtime as it is (its pageable path pins the pages for the
 * duration of the copy and moves 39 MB in 0.9 ms, where memcpy into our staging buffer plus the DMA
 * takes 2.3 ms); below that the fixed cost of the pageable path loses to the staging buffer.
 *
 * The pageable path is what the reference does for every vector (cudaMemcpy on the caller's memory,
 * libspmv/gpu.c:264,285) and it is the right default for solvers, whose work vectors live as long as
 * the solve: SparseBench 170^3 0.55 s against 1.1 s through staging, 100^3 0.22 against 0.33,
 * PageRank 0.20 against 0.30.  It is the wrong one for a caller that allocates a fresh vector for
 * every call: the runtime then pins new pages each time (8 MB: 4-27 ms instead of 0.2), which makes
 * our BFS driver -- like bfs.cc it creates new vectors as it goes -- take 0.30 s instead of 0.23 s
 * (SPMV_HIP_PIN=staging; 0.43 s before untouched memory was recognised with mincore(), see mostly_resident).  Timing the direct path at run time and falling back was tried and dropped
 * (profiles/r01_abi_transfer_modes.log): one slow sample costs as much as forty staged calls, a
 * one-megabyte trial does not show the effect, and two samples are not enough to tell the cases apart.
 */
#define DIRECT_PAGEABLE_MIN_BYTES ((size_t)1 << 20)

/* already pinned by an earlier call?  (checked before dladdr(), which walks the link map) */
static int is_registered(const void *p, size_t bytes)
{
    const char *lo = (const char *)p, *hi = lo + bytes;
    for (int r = 0; r < G.nreg; ++r)
        if (lo >= G.reg[r].lo && hi <= G.reg[r].hi)
            return 1;
    return 0;
}

/*
 * Untouched memory (a calloc'ed vector nobody has written yet: the BFS driver's frontier at the first
 * level of every source) has no pages behind it; the runtime's pageable path would have to fault every one
 * of them in while pinning -- 27 ms for 8 MB -- whereas memcpy into the staging buffer reads the shared zero
 * page.  mincore() tells: one system call per large vector (about 20 us).
 */
static int mostly_resident(const void *p, size_t bytes)
{
    static unsigned char *vec;
    static size_t vec_n;
    const uintptr_t pg = (uintptr_t)G.page;
    const uintptr_t lo = (uintptr_t)p / pg * pg, hi = ((uintptr_t)p + bytes + pg - 1) / pg * pg;
    const size_t n = (size_t)((hi - lo) / pg);
    if (n > vec_n) {
        free(vec);
        vec = (unsigned char *)malloc(n);
        vec_n = vec ? n : 0;
        if (!vec)
            return 1;
    }
    if (mincore((void *)lo, (size_t)(hi - lo), vec) != 0)
        return 1; /* cannot tell: behave like the reference */
    size_t in = 0;
    for (size_t i = 0; i < n; ++i)
        in += vec[i] & 1u;
    return in * 16 >= n * 15;
}

static int use_in_place(const void *p, size_t bytes)
{
    if (G.pin == PIN_PAGEABLE)
        return 1;
    if (G.pin != PIN_STAGING && is_registered(p, bytes))
        return 1;
    if (G.pin == PIN_REGISTER || (G.pin == PIN_AUTO && has_static_storage(p, bytes)))
        return ensure_registered(p, bytes);
    if (G.pin == PIN_AUTO && bytes >= DIRECT_PAGEABLE_MIN_BYTES)
        return mostly_resident(p, bytes);
    return 0;
}

/* Is [p, p+bytes) pinned where it lies (static storage under auto, anything under register)? */
static int pinned_in_place(const void *p, size_t bytes)
{
    if (G.pin != PIN_STAGING && G.pin != PIN_PAGEABLE && is_registered(p, bytes))
        return 1;
    if (G.pin == PIN_REGISTER || (G.pin == PIN_AUTO && has_static_storage(p, bytes)))
        return ensure_registered(p, bytes);
    return 0;
}

/*
 * Small vectors: the DMA path costs two engine set-ups and 35 us for a class-A-sized round trip; a
 * copy kernel that reads x out of pinned host memory and kernels that write y straight into pinned
 * host memory do it in 19 us (profiles/r01_xferbench_pcie_round_trip.log).  Through the whole ABI:
 * NPB class A 38.5 -> 26.9 us per call, class B (600-KB vectors) -6 %, class C (1.2 MB) no gain, and
 * from 4 MB up the DMA engines win: the limit is 1 MB, above it the regular path below is used.  Pinned = the caller's static vector registered
 * in place, else our staging buffer.  Returns 0 when the call was served this way.
 */
#define ZERO_COPY_MAX_BYTES ((size_t)1 << 20)
#define Y_OVERLAP_MIN_BYTES ((size_t)4 << 20) /* from here a y travels in row chunks beside the kernels */
#define Y_OVERLAP_CHUNKS 8

static int harness_zero_copy(slot *sl, void *ov, const void *iv, size_t xbytes, size_t ybytes)
{
    if (!G.zero_copy || sl->nparts != 1 || G.pin == PIN_PAGEABLE || G.stats >= 2 || xbytes > ZERO_COPY_MAX_BYTES ||
        ybytes > ZERO_COPY_MAX_BYTES || (xbytes & 3) || ((uintptr_t)iv & 3))
        return -1;
    void *xsrc = (void *)(uintptr_t)iv;
    if (!pinned_in_place(iv, xbytes)) {
        xsrc = stage_buf(&G.xstage, &G.xstage_n, xbytes);
        memcpy(xsrc, iv, xbytes);
    }
    void *ydst = ov;
    int y_staged = 0;
    if (!pinned_in_place(ov, ybytes)) {
        ydst = stage_buf(&G.ystage, &G.ystage_n, ybytes);
        y_staged = 1;
    }
    void *xdev = spmvd_host_devptr(xsrc), *ydev = spmvd_host_devptr(ydst);
    if (!xdev || !ydev)
        return -1; /* not mapped into the device's address space: the DMA path takes the call */
    const int dev = sl->part_dev[0];
    CHECK(spmvd_copy_words_async(dev, sl->dx[0], xdev, xbytes));
    CHECK(spmv_hip_spmv_device(sl->part[0], sl->dx[0], ydev, NULL));
    CHECK(spmvd_stream_sync(dev));
    if (y_staged)
        memcpy(ov, ydst, ybytes);
    return 0;
}

/* ---------------------------------------------------------------- the call */

static void *harness(void *ov, const void *a, const void *iv, const int *rowstr, const int *colidx,
                     const int *rows_p, int elem)
{
    pthread_mutex_lock(&g_mu);
    const double t_in = spmv_now();
    const int rows = *rows_p;
    if (rows <= 0) {
        pthread_mutex_unlock(&g_mu);
        return NULL;
    }
    const int64_t nnz = (int64_t)rowstr[rows] - (int64_t)rowstr[0];
    if (nnz <= 0) { /* every row is empty: y = 0 (native-impl.c:6,9), nothing to multiply */
        memset(ov, 0, (size_t)rows * elem);
        pthread_mutex_unlock(&g_mu);
        return NULL;
    }
    lazy_init();
    slot *sl = slot_find(a, rowstr, colidx, rows, nnz, elem);
    if (dispatch_host(a, rowstr, colidx, rows, nnz, elem, sl != NULL)) {
        host_product(ov, a, iv, rowstr, colidx, rows, nnz, elem, t_in);
        pthread_mutex_unlock(&g_mu);
        return NULL;
    }
    device_init();

    int refresh = 0; /* the sparsity pattern of the cached copy stands, its values must be uploaded again */
    if (sl) {
        int full = G.cache == CACHE_OFF || sl->uncacheable || sl->first != rowstr[0];
        int gd[3] = {0, 0, 0};
        if (!full && G.cache != CACHE_POINTER)
            for (int k = 0; k < 3; ++k)
                gd[k] = sl->g[k].dirty ? (int)sl->g[k].dirty : (guard_fragments_changed(&sl->g[k]) ? GUARD_WRITTEN : 0);
        if (sl->values_volatile)
            gd[0] = GUARD_WRITTEN; /* not watched: taken to have changed */
        /* the caller wrote into the arrays -- as opposed to: a neighbour's write made their pages writable, after which
         * nothing is known about them; that costs an upload but says nothing about how long this matrix's copies live */
        const int own = gd[0] == GUARD_WRITTEN || gd[1] == GUARD_WRITTEN || gd[2] == GUARD_WRITTEN || sl->first != rowstr[0];
        if (gd[1] || gd[2] || gd[0] == GUARD_EXPOSED)
            full = 1;
        if (!full && gd[0])
            refresh = 1;
        else if (!full && sl->probing) {
            /* the values were left alone since the look: they have stopped changing; back to the slot's own watch */
            sl->probing = 0;
            guard_free(&sl->g[0]);
            slot_arm_guard(sl, 0, sl->protect);
            if (sl->uncacheable)
                full = 1;
            else if (G.plan == SPMV_HIP_PLAN_AUTO) /* the plan the matrix would have got had it never been volatile */
                for (int p = 0; p < sl->nparts; ++p)
                    if (spmv_hip_matrix_set_plan(sl->part[p], SPMV_HIP_PLAN_AUTO, 0) != 0)
                        die("re-planning a matrix whose values have settled failed: %s", spmv_hip_last_error());
        }
        if (full) {
            if (G.cache != CACHE_OFF && own) {
                G.st.invalidations++;
                churn_entry *e = churn_find(a, rowstr, colidx, 1);
                e->short_lived = sl->uses < SHORT_LIVED_USES ? e->short_lived + 1 : 0;
            }
            if (G.verbose && G.cache != CACHE_OFF)
                fprintf(stderr, "libhip-spmv: %s, dropping the device copy\n",
                        own ? "host arrays changed" : "pages shared with another cached matrix went writable");
            const double td = spmv_now();
            slot_drop(sl);
            if (G.verbose >= 2)
                fprintf(stderr, "libhip-spmv: dropped the stale copy in %.3f ms\n", 1e3 * (spmv_now() - td));
            sl = NULL;
            refresh = 0;
        }
    }
    if (sl && refresh && G.dispatch != DISPATCH_DEVICE && G.nparts == 1 &&
        host_estimate(rows, nnz) <= refresh_estimate(nnz, elem) + device_estimate(rows, nnz, elem)) {
        /* new values every call and the host multiplies sooner than they cross PCIe: the copy keeps its pattern and
         * waits for a call that finds the values as it left them, or for the next refresh */
        host_product(ov, a, iv, rowstr, colidx, rows, nnz, elem, t_in);
        pthread_mutex_unlock(&g_mu);
        return NULL;
    }
    if (!sl && G.dispatch != DISPATCH_DEVICE && G.nparts == 1) {
        /* a churning matrix: the host loop, unless another upload is the faster way or it is time to look again */
        churn_entry *e = churn_find(a, rowstr, colidx, 0);
        if (e && e->short_lived >= 2 && e->host_streak < CHURN_PROBE_CALLS &&
            host_estimate(rows, nnz) <= e->last_upload_s + device_estimate(rows, nnz, elem)) {
            e->host_streak++;
            host_product(ov, a, iv, rowstr, colidx, rows, nnz, elem, t_in);
            pthread_mutex_unlock(&g_mu);
            return NULL;
        }
    }
    if (!sl) {
        const double tu = spmv_now();
        sl = slot_victim();
        slot_upload(sl, a, rowstr, colidx, rows, nnz, elem);
        if (G.verbose >= 2)
            fprintf(stderr, "libhip-spmv: victim + upload + arm in %.3f ms\n", 1e3 * (spmv_now() - tu));
    } else if (refresh) {
        refresh_values(sl, a, colidx);
    }
    sl->last_use = ++G.tick;
    sl->uses++;
    sl->uses_since_refresh++;
    if (sl->uses == SHORT_LIVED_USES) { /* this copy has lived: whatever churned before has stopped */
        churn_entry *e = churn_find(a, rowstr, colidx, 0);
        if (e)
            e->short_lived = 0;
    }
    G.last = sl;

    const size_t xbytes = (size_t)sl->cols * elem;
    const size_t ybytes = (size_t)rows * elem;
    const int detailed = G.stats >= 2;
    double t0 = spmv_now(), t1;

    if (harness_zero_copy(sl, ov, iv, xbytes, ybytes) == 0) {
        G.st.calls++;
        G.st.t_total += spmv_now() - t_in;
        pthread_mutex_unlock(&g_mu);
        return NULL;
    }

    /*
     * x: host -> every part's device (libspmv/gpu.c:264).  One part: a large heap vector is handed to the runtime as
     * it is (use_in_place).  Several parts on several GPUs: the runtime's pageable path holds the host thread for the
     * length of every copy, so N devices would receive x one after the other; instead x goes into the pinned staging
     * buffer once (a few host threads) and from there to all devices at the same time, each over its own PCIe link.
     */
    const int multi = sl->nparts > 1 && G.pin != PIN_PAGEABLE;
    const void *xsrc = iv;
    if (multi ? !pinned_in_place(iv, xbytes) : !use_in_place(iv, xbytes)) {
        void *xs = stage_buf(&G.xstage, &G.xstage_n, xbytes);
        par_memcpy(xs, iv, xbytes);
        xsrc = xs;
    }
    for (int p = 0; p < sl->nparts; ++p)
        CHECK(spmvd_copy_h2d_async(sl->part_dev[p], sl->dx[p], xsrc, xbytes));
    if (detailed) {
        for (int p = 0; p < sl->nparts; ++p)
            CHECK(spmvd_stream_sync(sl->part_dev[p]));
        t1 = spmv_now();
        G.st.t_x_h2d += t1 - t0;
        t0 = t1;
    }

    /* a long y of one part: in row chunks, each on its way to the host while the next are computed (the y D2H of
     * SparseBench 170^3 takes longer than its kernel) */
    /* ... when y is pinned where it lies (static storage under SPMV_HIP_PIN=auto: the Fortran callers' COMMON vectors; any
     * vector under =register): then the chunks are true DMA transfers beside the kernels (SparseBench 170^3, 201 calls:
     * 0.70 -> 0.63 s).  Into pageable memory the runtime's copy path holds the host thread and the chunks cost more than
     * they hide (0.57 -> 0.69 s; profiles/r03_sb170_y_overlap_by_pin_mode.log), so a heap vector gets one copy as before. */
    if (sl->nparts == 1 && !detailed && G.y_overlap && ybytes >= Y_OVERLAP_MIN_BYTES && pinned_in_place(ov, ybytes)) {
        char *yd = (char *)ov;
        const int rc = spmvd_spmv_y_overlapped(sl->part[0], sl->dx[0], sl->dy[0], yd, Y_OVERLAP_CHUNKS);
        if (rc < 0)
            die("spmvd_spmv_y_overlapped: %s", spmv_hip_last_error());
        if (rc == 1) {
            G.st.calls++;
            G.st.t_total += spmv_now() - t_in;
            pthread_mutex_unlock(&g_mu);
            return NULL;
        }
    }

    /* y_part = A_part * x, one launch sequence per part, each on its device's stream */
    for (int p = 0; p < sl->nparts; ++p)
        CHECK(spmv_hip_spmv_device(sl->part[p], sl->dx[p], sl->dy[p], NULL));
    if (detailed) {
        for (int p = 0; p < sl->nparts; ++p)
            CHECK(spmvd_stream_sync(sl->part_dev[p]));
        t1 = spmv_now();
        G.st.t_kernel += t1 - t0;
        t0 = t1;
    }

    /* y: each part's slice straight to its place in ov -- the host gather (libspmv/gpu.c:285) */
    char *ydst = (char *)ov;
    int y_staged = 0;
    if (multi ? !pinned_in_place(ov, ybytes) : !use_in_place(ov, ybytes)) {
        ydst = (char *)stage_buf(&G.ystage, &G.ystage_n, ybytes);
        y_staged = 1;
    }
    for (int p = 0; p < sl->nparts; ++p) {
        size_t r0 = (size_t)sl->bounds[p], r1 = (size_t)sl->bounds[p + 1];
        if (r1 > r0)
            CHECK(spmvd_copy_d2h_async(sl->part_dev[p], ydst + r0 * elem, sl->dy[p], (r1 - r0) * elem));
    }
    for (int p = 0; p < sl->nparts; ++p)
        CHECK(spmvd_stream_sync(sl->part_dev[p]));
    if (y_staged)
        par_memcpy(ov, ydst, ybytes);
    if (detailed)
        G.st.t_y_d2h += spmv_now() - t0;

    G.st.calls++;
    G.st.t_total += spmv_now() - t_in;
    pthread_mutex_unlock(&g_mu);
    return NULL;
}

void *spmv_harness_(double *ov, double *a, double *iv, int *rowstr, int *colidx, int *rows)
{
    return harness(ov, a, iv, rowstr, colidx, rows, 8);
}

void *f_spmv_harness_(float *ov, float *a, float *iv, int *rowstr, int *colidx, int *rows)
{
    return harness(ov, a, iv, rowstr, colidx, rows, 4);
}

/* ---------------------------------------------------------------- introspection */

void spmv_hip_get_stats(spmv_hip_stats *out)
{
    pthread_mutex_lock(&g_mu);
    *out = G.st;
    pthread_mutex_unlock(&g_mu);
}

void spmv_hip_reset_stats(void)
{
    pthread_mutex_lock(&g_mu);
    double keep = G.st.t_init;
    memset(&G.st, 0, sizeof G.st);
    G.st.t_init = keep;
    pthread_mutex_unlock(&g_mu);
}

int spmv_hip_abi_matrix_info(spmv_hip_info *out)
{
    int rc = -1;
    pthread_mutex_lock(&g_mu);
    if (G.last && G.last->used && G.last->part[0]) {
        rc = spmv_hip_matrix_info(G.last->part[0], out);
        if (rc == 0 && G.last->nparts > 1) { /* report the whole matrix, not part 0 */
            out->rows = G.last->rows;
            out->nnz = G.last->nnz;
            out->algorithmic_bytes = G.last->nnz * (int64_t)(G.last->elem + 4) +
                                     ((int64_t)G.last->rows + 1) * 4 +
                                     (int64_t)G.last->cols * G.last->elem +
                                     (int64_t)G.last->rows * G.last->elem;
        }
    }
    pthread_mutex_unlock(&g_mu);
    return rc;
}
