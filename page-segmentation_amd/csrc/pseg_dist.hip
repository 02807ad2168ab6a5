// pseg_dist.hip -- the one collective of the build: data-parallel training all-reduces the flat float32
// gradient buffer (all parameters in weight-table order + the metric slots: 673 017 floats, 2.7 MB for fcn_skip) over RCCL,
// one process per GPU.  The reference has no distributed code at all (lib/network.py:235-241 is a single-process fit);
// these entries let a C caller train data-parallel without Python -- the Python mirror can use torch.distributed on the same
// buffer instead (pseg_amd/parallel.py).
//
// RCCL is bound at run time (dlopen): libpseg.so links nothing but the HIP runtime, a process that never trains
// data-parallel never loads the library, and a Python process that already holds torch's copy shares it.
#include <dlfcn.h>

#include <cstring>

#include "pseg_common.h"

// The library is bound by dlopen, so the handful of ABI facts used below (the id's size, two enum values, five prototypes) are
// restated here by hand -- and pinned against RCCL's own header wherever that header exists at build time (it does in the ROCm
// image): a change of any of them stops the build instead of surfacing as a wrong reduction on the first N > 1 run.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#include <type_traits>
static_assert(NCCL_UNIQUE_ID_BYTES == 128 && sizeof(ncclUniqueId) == 128, "pseg_allreduce_unique_id hands out 128 bytes");
static_assert(ncclSuccess == 0, "a zero return is success");
static_assert(ncclFloat32 == 7 && ncclSum == 0, "pseg_train_allreduce passes ncclFloat32 / ncclSum as 7 / 0");
static_assert(sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int) && sizeof(ncclResult_t) == sizeof(int), "enums travel as int");
static_assert(std::is_pointer<ncclComm_t>::value, "the communicator is kept as a void*");
static_assert(std::is_same<decltype(&ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId*)>::value, "ncclGetUniqueId");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>::value, "ncclCommInitRank");
static_assert(std::is_same<decltype(&ncclAllReduce), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)>::value, "ncclAllReduce");
static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value, "ncclCommDestroy");
static_assert(std::is_same<decltype(&ncclGetErrorString), const char* (*)(ncclResult_t)>::value, "ncclGetErrorString");
#define PSEG_RCCL_ABI_PINNED 1
#else
#define PSEG_RCCL_ABI_PINNED 0
#endif

namespace pseg {

namespace {
typedef struct { char internal[128]; } nccl_uid;          // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(void**, int, nccl_uid, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);
struct Rccl {
    void* lib = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_allreduce allreduce = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.lib) return PSEG_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;        // a copy this process already holds (torch's)
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(PSEG_EUNSUPPORTED, "RCCL is not available: %s", dlerror());
    Rccl r;
    r.lib = h;
    r.get_uid = (fn_get_uid)dlsym(h, "ncclGetUniqueId");
    r.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    r.allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce");
    r.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    r.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    if (!r.get_uid || !r.init_rank || !r.allreduce || !r.destroy) return fail(PSEG_EUNSUPPORTED, "librccl lacks an expected symbol");
    g_rccl = r;
    return PSEG_OK;
}
int rccl_fail(const char* what, int rc) {
    return fail(PSEG_EHIP, "%s failed: %s", what, g_rccl.errstr ? g_rccl.errstr(rc) : "RCCL error");
}
}  // namespace

struct DistState {
    void* comm = nullptr;
    int rank = 0, world = 1;
};

void dist_free(Engine& e) {
    auto* d = (DistState*)e.dist;
    if (!d) return;
    if (d->comm && g_rccl.destroy) (void)g_rccl.destroy(d->comm);
    delete d;
    e.dist = nullptr;
}

}  // namespace pseg

using namespace pseg;

extern "C" {

int pseg_allreduce_unique_id(uint8_t id[128]) {
    if (!id) return fail(PSEG_EINVAL, "NULL argument");
    PSEG_TRY(rccl_load());
    nccl_uid u;
    const int rc = g_rccl.get_uid(&u);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id, u.internal, 128);
    return PSEG_OK;
}

int pseg_allreduce_init(pseg_engine* h, int rank, int world, const uint8_t id[128]) {
    if (!h || !id) return fail(PSEG_EINVAL, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(PSEG_EINVAL, "bad rank %d / world %d", rank, world);
    Engine& e = h->e;
    PSEG_TRY(rccl_load());
    PSEG_HIP(hipSetDevice(e.device));
    dist_free(e);
    auto* d = new DistState();
    d->rank = rank;
    d->world = world;
    nccl_uid u;
    memcpy(u.internal, id, 128);
    const int rc = g_rccl.init_rank(&d->comm, world, u, rank);
    if (rc != 0) { delete d; return rccl_fail("ncclCommInitRank", rc); }
    e.dist = d;
    return PSEG_OK;
}

int pseg_train_allreduce(pseg_engine* h) {
    if (!h || !h->e.dist) return fail(PSEG_EINVAL, "pseg_allreduce_init has not been called");
    Engine& e = h->e;
    auto* d = (DistState*)e.dist;
    float* g = nullptr;
    int64_t n = 0;
    PSEG_TRY(pseg_train_grad_buffer(h, &g, &n));
    PSEG_HIP(hipSetDevice(e.device));
    // enqueued on the engine's stream: behind the backward kernels that wrote the buffer, in front of the clip + optimizer
    // kernels of pseg_train_apply -- no host or device synchronisation
    const int rc = g_rccl.allreduce(g, g, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, d->comm, e.stream);
    if (rc != 0) return rccl_fail("ncclAllReduce", rc);
    return PSEG_OK;
}

/* 1 when this build checked its hand-written RCCL declarations against <rccl/rccl.h> (static_asserts above), else 0 */
int pseg_rccl_abi_pinned(void) { return PSEG_RCCL_ABI_PINNED; }

int pseg_allreduce_destroy(pseg_engine* h) {
    if (!h) return fail(PSEG_EINVAL, "NULL engine");
    dist_free(h->e);
    return PSEG_OK;
}

}  // extern "C"
