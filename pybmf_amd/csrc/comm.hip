// The exchange of the row-sharded loop, issued from C (SURVEY 8b "bmf_allreduce", 8e): one communicator object per rank that
// owns an RCCL communicator, a side stream for the collectives and the events that fence it against the compute stream.
//
// RCCL is resolved at run time (dlopen of "librccl.so.1"): inside a PyTorch process that is the copy torch has already loaded
// -- the dynamic loader matches by soname -- so the process keeps ONE RCCL and ONE HIP runtime; a single-GPU user never loads it.
//
// Two kinds of communicator:
//   BMF_COMM_RCCL  ncclCommInitRank on the current device; collectives are ncclAllReduce(sum, in place) on the side stream.
//   BMF_COMM_HOST  the all-reduce is a caller-supplied, stream-ordered host function (the tests run the C loop with two ranks
//                  on ONE GPU this way, over torch.distributed's gloo group: RCCL refuses two ranks on one device).  The
//                  sequencing code -- streams, events, block order -- is the same for both kinds.
#include "common.h"

#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

#include "comm.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};

// nullptr + bmf_last_error() when RCCL cannot be loaded
const RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    static bool ok = false;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) return;
#define BMF_SYM(field, sym) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, sym))
        BMF_SYM(GetUniqueId, "ncclGetUniqueId");
        BMF_SYM(CommInitRank, "ncclCommInitRank");
        BMF_SYM(CommDestroy, "ncclCommDestroy");
        BMF_SYM(AllReduce, "ncclAllReduce");
        BMF_SYM(GroupStart, "ncclGroupStart");
        BMF_SYM(GroupEnd, "ncclGroupEnd");
        BMF_SYM(GetErrorString, "ncclGetErrorString");
        BMF_SYM(GetVersion, "ncclGetVersion");
#undef BMF_SYM
        ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.GroupStart && api.GroupEnd && api.GetErrorString;
    });
    if (!ok) {
        bmf_set_error("RCCL is not available: dlopen(\"librccl.so.1\") failed or lacks a symbol (%s)", api.handle ? "symbol lookup" : dlerror());
        return nullptr;
    }
    return &api;
}

#define BMF_NCCL_CHECK(api, expr)                                                                             \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess) {                                                                              \
            bmf_set_error("%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(r_), __FILE__, __LINE__);     \
            return BMF_ERR_COMM;                                                                              \
        }                                                                                                     \
    } while (0)

}  // namespace

// 1 when RCCL can be loaded in this process (dlopen + every symbol), 0 otherwise (bmf_last_error says why).  Touches no GPU and
// contacts no other rank: the ranks of a job call it and AGREE on the answer before any of them enters ncclCommInitRank, which
// is a collective -- a rank that returned early from bmf_comm_create would leave the others blocked inside it.
extern "C" int bmf_comm_available(void) { return rccl() ? 1 : 0; }

extern "C" int bmf_comm_unique_id(void* id_host) {
    BMF_REQUIRE(id_host, "bmf_comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == BMF_COMM_ID_BYTES, "BMF_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    const RcclApi* api = rccl();
    if (!api) return BMF_ERR_COMM;
    ncclUniqueId id;
    BMF_NCCL_CHECK(api, api->GetUniqueId(&id));
    memcpy(id_host, &id, sizeof(id));
    return BMF_OK;
}

static int comm_common_init(bmf_comm* c) {
    BMF_HIP_CHECK(hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking));
    for (hipEvent_t& e : c->ev) BMF_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return BMF_OK;
}

extern "C" int bmf_comm_create(const void* id_host, int32_t world, int32_t rank, bmf_comm** out) {
    BMF_REQUIRE(id_host && out, "bmf_comm_create: null pointer");
    BMF_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bmf_comm_create: rank %d outside a world of %d", rank, world);
    const RcclApi* api = rccl();
    if (!api) return BMF_ERR_COMM;
    bmf_comm* c = new (std::nothrow) bmf_comm();
    BMF_REQUIRE(c, "bmf_comm_create: out of memory");
    c->kind = BMF_COMM_RCCL;
    c->world = world;
    c->rank = rank;
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof(id));
    ncclComm_t nc = nullptr;
    ncclResult_t r = api->CommInitRank(&nc, world, id, rank);
    if (r != ncclSuccess) {
        bmf_set_error("ncclCommInitRank(world=%d, rank=%d) failed: %s", world, rank, api->GetErrorString(r));
        delete c;
        return BMF_ERR_COMM;
    }
    c->nccl = nc;
    int rc = comm_common_init(c);
    if (rc != BMF_OK) {
        (void)api->CommDestroy(nc);
        delete c;
        return rc;
    }
    *out = c;
    return BMF_OK;
}

extern "C" int bmf_comm_create_host(bmf_allreduce_fn fn, void* user, int32_t world, int32_t rank, bmf_comm** out) {
    BMF_REQUIRE(fn && out, "bmf_comm_create_host: null pointer");
    BMF_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bmf_comm_create_host: rank %d outside a world of %d", rank, world);
    bmf_comm* c = new (std::nothrow) bmf_comm();
    BMF_REQUIRE(c, "bmf_comm_create_host: out of memory");
    c->kind = BMF_COMM_HOST;
    c->world = world;
    c->rank = rank;
    c->fn = fn;
    c->user = user;
    int rc = comm_common_init(c);
    if (rc != BMF_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return BMF_OK;
}

extern "C" int bmf_comm_destroy(bmf_comm* c) {
    if (!c) return BMF_OK;
    (void)bmf_comm_timing(c, 0);
    if (c->cs) (void)hipStreamSynchronize(c->cs);
    if (c->kind == BMF_COMM_RCCL && c->nccl) {
        const RcclApi* api = rccl();
        if (api) (void)api->CommDestroy((ncclComm_t)c->nccl);
    }
    for (hipEvent_t& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->cs) (void)hipStreamDestroy(c->cs);
    delete c;
    return BMF_OK;
}

extern "C" int bmf_comm_info(const bmf_comm* c, int32_t* kind, int32_t* world, int32_t* rank, int32_t* rccl_version) {
    BMF_REQUIRE(c, "bmf_comm_info: null communicator");
    if (kind) *kind = c->kind;
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    if (rccl_version) {
        *rccl_version = 0;
        if (c->kind == BMF_COMM_RCCL) {
            const RcclApi* api = rccl();
            int v = 0;
            if (api && api->GetVersion && api->GetVersion(&v) == ncclSuccess) *rccl_version = v;
        }
    }
    return BMF_OK;
}

// ---- internal: the primitives the sharded loop (api.hip) is written in ------------------------------------------------

int bmf_comm_group_begin(bmf_comm* c) {
    if (c->kind != BMF_COMM_RCCL) return BMF_OK;
    const RcclApi* api = rccl();
    if (!api) return BMF_ERR_COMM;
    BMF_NCCL_CHECK(api, api->GroupStart());
    return BMF_OK;
}

int bmf_comm_group_end(bmf_comm* c) {
    if (c->kind != BMF_COMM_RCCL) return BMF_OK;
    const RcclApi* api = rccl();
    if (!api) return BMF_ERR_COMM;
    BMF_NCCL_CHECK(api, api->GroupEnd());
    return BMF_OK;
}

// sum over the ranks, in place, ordered on stream s; dtype: BMF_DTYPE_F32 / BMF_DTYPE_F64
int bmf_comm_allreduce_on(bmf_comm* c, void* buf, int64_t count, int dtype, hipStream_t s) {
    if (count <= 0) return BMF_OK;
    if (c->kind == BMF_COMM_HOST) {
        const int rc = c->fn(c->user, buf, count, dtype, (void*)s);
        if (rc != 0) {
            bmf_set_error("the host all-reduce callback failed with code %d", rc);
            return BMF_ERR_COMM;
        }
        return BMF_OK;
    }
    const RcclApi* api = rccl();
    if (!api) return BMF_ERR_COMM;
    BMF_NCCL_CHECK(api, api->AllReduce(buf, buf, (size_t)count, dtype == BMF_DTYPE_F64 ? ncclFloat64 : ncclFloat32, ncclSum, (ncclComm_t)c->nccl, s));
    return BMF_OK;
}

// ---- public: the fused exchange as one call (bmf_allreduce(h, f32_buf, n, ..., stream)) ------------------------

extern "C" int bmf_allreduce(bmf_comm* c, float* f32_buf, int64_t n32, double* f64_buf, int64_t n64, void* stream) {
    BMF_REQUIRE(c, "bmf_allreduce: null communicator");
    BMF_REQUIRE(n32 >= 0 && n64 >= 0 && (n32 == 0 || f32_buf) && (n64 == 0 || f64_buf), "bmf_allreduce: bad buffer description");
    hipStream_t s = (hipStream_t)stream;
    int rc = bmf_comm_group_begin(c);
    if (rc != BMF_OK) return rc;
    int rc1 = bmf_comm_allreduce_on(c, f32_buf, n32, BMF_DTYPE_F32, s);
    int rc2 = rc1 == BMF_OK ? bmf_comm_allreduce_on(c, f64_buf, n64, BMF_DTYPE_F64, s) : rc1;
    rc = bmf_comm_group_end(c);
    return rc2 != BMF_OK ? rc2 : rc;
}

// ---- timing of the exchange (bench.py `distributed` block) ---------------------------------------------------------------

extern "C" int bmf_comm_timing(bmf_comm* c, int32_t max_steps) {
    BMF_REQUIRE(c, "bmf_comm_timing: null communicator");
    BMF_REQUIRE(max_steps >= 0 && max_steps <= (1 << 16), "bmf_comm_timing: max_steps out of range");
    for (hipEvent_t e : c->tev) (void)hipEventDestroy(e);
    c->tev.clear();
    c->t_used = 0;
    c->t_cap = 0;
    if (max_steps == 0) return BMF_OK;
    c->tev.resize(3 * (size_t)max_steps);
    for (hipEvent_t& e : c->tev) BMF_HIP_CHECK(hipEventCreate(&e));
    c->t_cap = max_steps;
    return BMF_OK;
}

extern "C" int bmf_comm_timing_read(bmf_comm* c, int32_t* steps, double* exposed_ms, double* span_ms) {
    BMF_REQUIRE(c && steps && exposed_ms && span_ms, "bmf_comm_timing_read: null pointer");
    double ex = 0.0, sp = 0.0;
    for (int i = 0; i < c->t_used; ++i) {
        BMF_HIP_CHECK(hipEventSynchronize(c->tev[3 * i + 2]));
        float a = 0.f, b = 0.f;
        BMF_HIP_CHECK(hipEventElapsedTime(&a, c->tev[3 * i + 1], c->tev[3 * i + 2]));
        BMF_HIP_CHECK(hipEventElapsedTime(&b, c->tev[3 * i], c->tev[3 * i + 2]));
        ex += a;
        sp += b;
    }
    *steps = c->t_used;
    *exposed_ms = ex;
    *span_ms = sp;
    c->t_used = 0;
    return BMF_OK;
}
