/* TEST INFRASTRUCTURE, not product: a stand-in for librccl.so that lets 2-4 processes which share ONE GPU
 * drive the library's own ncclAllReduce carrier (calib_rccl_load(path) -> calib_rccl_init_deadline ->
 * calib_rccl_selftest -> calib_lm_run_sharded's ncclAllReduce branch, csrc/calib_lm.hip). The real RCCL refuses two
 * ranks on one device ("Duplicate GPU detected"), and a GPU box has one device.
 *
 * Only the five entry points the library resolves, with rccl.h's calling convention: ncclUniqueId is 128 opaque
 * bytes passed BY VALUE, ncclFloat64 = 8, ncclSum = 0, ncclSuccess = 0. Semantics kept from the real thing:
 *   - ncclCommInitRank blocks until every rank of the id has joined (here: against CALIB_STANDIN_JOIN_TIMEOUT);
 *   - ncclAllReduce is ASYNCHRONOUS and stream-ordered: D2H copy -> host function -> H2D copy, all enqueued on the
 *     caller's stream, the host never waits; the host function meets the other ranks in a POSIX shared-memory
 *     segment named by the unique id and adds the contributions in rank order (every rank gets the same bits);
 *   - a rank that does not show up within CALIB_STANDIN_OP_TIMEOUT poisons the result with NaN and makes the NEXT
 *     call return ncclSystemError; ncclCommAbort releases everybody who is still waiting.
 * Fault injection for the tests: CALIB_STANDIN_CORRUPT="rank:seq" adds 1.0 to element 0 of that rank's result of its
 * seq-th all-reduce (1 = the library's start-up self-test), so exactly one rank sees wrong sums.
 *
 * Built by tests/fake_rccl/Makefile with gcc against the HIP host API; no device code. */
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

enum { STANDIN_MAX_RANKS = 8, STANDIN_MAX_COUNT = 1024 };
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
       ncclInvalidUsage = 5 };

typedef struct { char internal[128]; } ncclUniqueId;

typedef struct {
    _Atomic uint32_t joined;                       /* ranks that have mapped the segment */
    _Atomic uint32_t aborted;                      /* somebody gave up: every wait ends */
    _Atomic uint64_t arrive[STANDIN_MAX_RANKS];    /* sequence number of the last contribution each rank published */
    double data[2][STANDIN_MAX_RANKS][STANDIN_MAX_COUNT];
} Shared;

typedef struct {
    Shared* sh;
    int nranks, rank;
    uint64_t seq;                                  /* all-reduces enqueued so far (host order = stream order) */
    double* staging;                               /* pinned; one buffer is enough: the three steps of call k+1 are
                                                      ordered behind those of call k on the stream */
    _Atomic int failed;                            /* an exchange timed out or the communicator was aborted */
    double opTimeout;
    int corruptRank;
    uint64_t corruptSeq;
    char name[128];
} Comm;

typedef struct { Comm* c; size_t count; uint64_t seq; } Op;

static double nowSeconds(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static double envSeconds(const char* name, double dflt) {
    const char* s = getenv(name);
    if (!s || !*s) return dflt;
    const double v = atof(s);
    return v > 0 ? v : dflt;
}

static void nap(void) {
    struct timespec t = {0, 20000};
    sched_yield();
    nanosleep(&t, NULL);
}

const char* ncclGetErrorString(int rc) {
    switch (rc) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled hip error (stand-in)";
        case ncclSystemError: return "unhandled system error (stand-in: a rank did not arrive)";
        case ncclInvalidArgument: return "invalid argument (stand-in: fp64 sum of at most 1024 elements only)";
        case ncclInvalidUsage: return "invalid usage (stand-in)";
        default: return "internal error (stand-in)";
    }
}

int ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    static _Atomic unsigned counter;
    struct timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    memset(id->internal, 0, sizeof(id->internal));
    snprintf(id->internal, sizeof(id->internal), "/calib_rccl_standin_%d_%lx_%u", (int)getpid(),
             (unsigned long)t.tv_nsec ^ ((unsigned long)t.tv_sec << 20), atomic_fetch_add(&counter, 1u));
    return ncclSuccess;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > STANDIN_MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    if (id.internal[0] != '/') return ncclInvalidArgument;
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, (off_t)sizeof(Shared)) != 0) { close(fd); return ncclSystemError; }   /* new pages read as zero */
    Shared* sh = (Shared*)mmap(NULL, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (sh == MAP_FAILED) return ncclSystemError;
    atomic_fetch_add(&sh->joined, 1u);
    const double deadline = nowSeconds() + envSeconds("CALIB_STANDIN_JOIN_TIMEOUT", 300.0);
    int ok = 1;
    while (atomic_load(&sh->joined) < (uint32_t)nranks) {
        if (atomic_load(&sh->aborted) || nowSeconds() > deadline) { ok = 0; break; }
        nap();
    }
    if (!ok || rank == 0) shm_unlink(id.internal);      /* everybody holds a mapping now (or nobody ever will) */
    if (!ok) {
        atomic_store(&sh->aborted, 1u);
        munmap(sh, sizeof(Shared));
        return ncclSystemError;
    }
    Comm* c = (Comm*)calloc(1, sizeof(Comm));
    if (!c) { munmap(sh, sizeof(Shared)); return ncclSystemError; }
    if (hipHostMalloc((void**)&c->staging, STANDIN_MAX_COUNT * sizeof(double), hipHostMallocDefault) != hipSuccess) {
        free(c);
        munmap(sh, sizeof(Shared));
        return ncclUnhandledCudaError;
    }
    c->sh = sh; c->nranks = nranks; c->rank = rank;
    c->opTimeout = envSeconds("CALIB_STANDIN_OP_TIMEOUT", 30.0);
    c->corruptRank = -1;
    const char* cor = getenv("CALIB_STANDIN_CORRUPT");
    if (cor && *cor) {
        int r = -1; unsigned long s = 0;
        if (sscanf(cor, "%d:%lu", &r, &s) == 2) { c->corruptRank = r; c->corruptSeq = s; }
    }
    memcpy(c->name, id.internal, sizeof(c->name));
    *comm = c;
    return ncclSuccess;
}

/* runs on a HIP runtime thread, in stream order; must not call into HIP */
static void exchange(void* p) {
    Op* op = (Op*)p;
    Comm* c = op->c;
    Shared* sh = c->sh;
    const int par = (int)(op->seq & 1u);
    /* two parities are enough: nobody can publish call k+2 before everybody has read call k (publishing k+1 comes
       after reading k, and completing k+1 needs everybody's k+1) */
    memcpy(sh->data[par][c->rank], c->staging, op->count * sizeof(double));
    atomic_store_explicit(&sh->arrive[c->rank], op->seq, memory_order_release);
    const double deadline = nowSeconds() + c->opTimeout;
    int ok = 1;
    for (int r = 0; r < c->nranks && ok; ++r)
        while (atomic_load_explicit(&sh->arrive[r], memory_order_acquire) < op->seq) {
            if (atomic_load(&sh->aborted) || atomic_load(&c->failed) || nowSeconds() > deadline) { ok = 0; break; }
            nap();
        }
    if (ok) {
        for (size_t i = 0; i < op->count; ++i) {
            double s = 0.0;
            for (int r = 0; r < c->nranks; ++r) s += sh->data[par][r][i];    /* rank order: same bits everywhere */
            c->staging[i] = s;
        }
        if (c->rank == c->corruptRank && op->seq == c->corruptSeq) c->staging[0] += 1.0;
    } else {
        atomic_store(&c->failed, 1);
        atomic_store(&sh->aborted, 1u);
        for (size_t i = 0; i < op->count; ++i) c->staging[i] = NAN;
    }
    free(op);
}

int ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (!c || !sendbuff || !recvbuff) return ncclInvalidArgument;
    if (datatype != 8 || op != 0 || count == 0 || count > STANDIN_MAX_COUNT) return ncclInvalidArgument;
    if (atomic_load(&c->failed)) return ncclSystemError;
    Op* o = (Op*)malloc(sizeof(Op));
    if (!o) return ncclSystemError;
    o->c = c; o->count = count; o->seq = ++c->seq;
    if (hipMemcpyAsync(c->staging, sendbuff, count * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess) { free(o); return ncclUnhandledCudaError; }
    if (hipLaunchHostFunc(stream, exchange, o) != hipSuccess) { free(o); return ncclUnhandledCudaError; }
    if (hipMemcpyAsync(recvbuff, c->staging, count * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

/* A host function of this communicator may still be queued or waiting: the communicator's memory is left in place
   (a few KB per communicator, test processes only) and only marked, so a late host function finds valid memory and
   returns at once with NaN. */
int ncclCommAbort(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclInvalidArgument;
    atomic_store(&c->failed, 1);
    atomic_store(&c->sh->aborted, 1u);
    return ncclSuccess;
}

/* the caller has drained its stream (calib_rccl_shutdown synchronises first) */
int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclInvalidArgument;
    atomic_store(&c->failed, 1);
    (void)hipHostFree(c->staging);
    munmap(c->sh, sizeof(Shared));
    free(c);
    return ncclSuccess;
}
