// Direct RCCL gradient all-reduce over xGMI (host-only code; no kernels here).
//
// Replaces the NCCL calls of the reference's nine DistributedDataParallel reducers (one per trainable module,
// vendor/speechbrain/speechbrain/core.py:1464-1484; bucket all-reduces fired from autograd hooks during backward, :1585-1615 no_sync)
// with plain ncclAllReduce launches on a caller-owned HIP stream: one communicator per process (one process per GPU), buckets of the
// flat gradient arena reduced in place as soon as they are complete. Going through librccl directly - not through torch.distributed's
// ProcessGroupNCCL - is what makes the collectives CAPTURABLE into the step's hipGraph: ProcessGroupNCCL's watchdog thread queries
// the events of captured collectives and aborts the process ("operation not permitted on an event last recorded in a capturing
// stream", PyTorch 2.10 + RCCL 2.26 on ROCm 7); a bare ncclAllReduce on a capturing stream simply becomes graph nodes.
// librccl is dlopen()ed at run time (the copy PyTorch ships and has already loaded, so that one RCCL lives in the process);
// the unique id is created on rank 0 and handed to the other ranks by the caller (any out-of-band channel: torch.distributed's store).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <string.h>

#include "../../include/tsasr_hip.h"

void tsasr_set_error(const char *fmt, ...);

namespace {
typedef void *comm_t;
struct unique_id { char internal[128]; };   // rccl.h: NCCL_UNIQUE_ID_BYTES
typedef int (*get_unique_id_fn)(unique_id *);
typedef int (*comm_init_rank_fn)(comm_t *, int, unique_id, int);
typedef int (*comm_destroy_fn)(comm_t);
typedef int (*all_reduce_fn)(const void *, void *, size_t, int, int, comm_t, hipStream_t);
typedef const char *(*error_string_fn)(int);
enum { kFloat32 = 7, kBfloat16 = 9, kSum = 0, kAvg = 4 };   // rccl.h: ncclDataType_t / ncclRedOp_t

void *g_lib = nullptr;
get_unique_id_fn p_get_unique_id = nullptr;
comm_init_rank_fn p_comm_init_rank = nullptr;
comm_destroy_fn p_comm_destroy = nullptr;
all_reduce_fn p_all_reduce = nullptr;
error_string_fn p_error_string = nullptr;
comm_t g_comm = nullptr;
int g_nranks = 0;

int fail(const char *what, int rc) {
    tsasr_set_error("%s: RCCL error %d (%s)", what, rc, p_error_string ? p_error_string(rc) : "?");
    return TSASR_E_LAUNCH;
}
}  // namespace

extern "C" {

/* dlopen librccl (path may be NULL: "librccl.so" through the loader's search path) and resolve the five entry points used. */
int tsasr_allreduce_load(const char *librccl_path) {
    if (g_lib) return 0;
    g_lib = dlopen(librccl_path ? librccl_path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!g_lib) {
        tsasr_set_error("tsasr_allreduce_load: %s", dlerror());
        return TSASR_E_INVALID;
    }
    p_get_unique_id = (get_unique_id_fn)dlsym(g_lib, "ncclGetUniqueId");
    p_comm_init_rank = (comm_init_rank_fn)dlsym(g_lib, "ncclCommInitRank");
    p_comm_destroy = (comm_destroy_fn)dlsym(g_lib, "ncclCommDestroy");
    p_all_reduce = (all_reduce_fn)dlsym(g_lib, "ncclAllReduce");
    p_error_string = (error_string_fn)dlsym(g_lib, "ncclGetErrorString");
    if (!p_get_unique_id || !p_comm_init_rank || !p_comm_destroy || !p_all_reduce) {
        tsasr_set_error("tsasr_allreduce_load: librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
        dlclose(g_lib);
        g_lib = nullptr;
        return TSASR_E_INVALID;
    }
    return 0;
}

/* Rank 0: 128 opaque bytes every rank must pass to tsasr_allreduce_init (host memory). */
int tsasr_allreduce_unique_id(void *host_id128) {
    if (!g_lib || !host_id128) {
        tsasr_set_error("tsasr_allreduce_unique_id: call tsasr_allreduce_load first");
        return TSASR_E_INVALID;
    }
    unique_id id;
    const int rc = p_get_unique_id(&id);
    if (rc) return fail("ncclGetUniqueId", rc);
    memcpy(host_id128, &id, sizeof(id));
    return 0;
}

/* Collective over all ranks: creates this process's communicator on the CURRENT device (one process per GPU). */
int tsasr_allreduce_init(const void *host_id128, int nranks, int rank) {
    if (!g_lib || !host_id128 || nranks < 1 || rank < 0 || rank >= nranks) {
        tsasr_set_error("tsasr_allreduce_init: bad arguments (loaded=%d nranks=%d rank=%d)", g_lib != nullptr, nranks, rank);
        return TSASR_E_INVALID;
    }
    if (g_comm) return 0;
    unique_id id;
    memcpy(&id, host_id128, sizeof(id));
    const int rc = p_comm_init_rank(&g_comm, nranks, id, rank);
    if (rc) {
        g_comm = nullptr;
        return fail("ncclCommInitRank", rc);
    }
    g_nranks = nranks;
    return 0;
}

int tsasr_allreduce_ready(void) { return g_comm != nullptr ? g_nranks : 0; }

/* In-place all-reduce of `count` elements of `buf` (TSASR_F32 or TSASR_BF16) over all ranks on `stream`: sum, or average when
 * `average` != 0. Asynchronous; ordering is the stream's (the caller joins `stream` before it reads `buf`). Capturable. Every rank
 * must issue the same sequence of calls. */
int tsasr_allreduce_bucket(void *buf, size_t count, int dtype, int average, void *stream) {
    if (!g_comm || !buf || (dtype != TSASR_F32 && dtype != TSASR_BF16)) {
        tsasr_set_error("tsasr_allreduce_bucket: no communicator / bad arguments");
        return TSASR_E_INVALID;
    }
    if (count == 0) return 0;
    const int rc = p_all_reduce(buf, buf, count, dtype == TSASR_F32 ? kFloat32 : kBfloat16, average ? kAvg : kSum, g_comm, (hipStream_t)stream);
    return rc ? fail("ncclAllReduce", rc) : 0;
}

int tsasr_allreduce_destroy(void) {
    if (g_comm) {
        const int rc = p_comm_destroy(g_comm);
        g_comm = nullptr;
        g_nranks = 0;
        if (rc) return fail("ncclCommDestroy", rc);
    }
    return 0;
}

}  // extern "C"
