// Data-parallel collectives of the C ABI (section 7 of include/bullet_amd.h): RCCL over xGMI, one communicator rank
// per process and GPU. The reference is single-device (crates/trainer/src/run.rs:84 takes one Device; no collective
// library anywhere in the tree), so there is nothing to mirror: this is what SURVEY 8e adds.
// RCCL is bound at run time (dlopen): the library stays loadable on a host without RCCL, and inside a PyTorch
// process it picks up the librccl PyTorch has already mapped instead of a second copy.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace {

typedef struct { char internal[128]; } nccl_unique_id;  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void *nccl_comm_t;
enum { NCCL_FLOAT32 = 7, NCCL_SUM = 0 };  // ncclDataType_t / ncclRedOp_t values of rccl.h

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(nccl_unique_id *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

int rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return BT_OK;
    static const char *const names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)  // a copy the process has mapped already (PyTorch's) first
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); i++) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    BT_REQUIRE(h, "bt_comm: librccl not found (%s)", dlerror());
    Rccl r;
    r.handle = h;
#define SYM(field, name)                                                        \
    do {                                                                        \
        *reinterpret_cast<void **>(&r.field) = dlsym(h, name);                  \
        BT_REQUIRE(r.field, "bt_comm: librccl has no symbol %s", name);         \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GetErrorString, "ncclGetErrorString");
    SYM(AllReduce, "ncclAllReduce");
    SYM(ReduceScatter, "ncclReduceScatter");
    SYM(AllGather, "ncclAllGather");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
#undef SYM
    g_rccl = r;
    return BT_OK;
}

#define BT_NCCL(call)                                                        \
    do {                                                                     \
        const int r_ = (call);                                               \
        if (r_ != 0) {                                                       \
            bt_set_error("RCCL: %s", g_rccl.GetErrorString(r_));             \
            return BT_ERR_HIP;                                               \
        }                                                                    \
    } while (0)

int hook_rc(int rc, const char *what) {
    if (rc == 0) return BT_OK;
    bt_set_error("bt_comm: the caller's %s failed (%d)", what, rc);
    return BT_ERR_STATE;
}

}  // namespace

extern "C" {

int bt_comm_unique_id(void *id128) {
    BT_REQUIRE(id128, "bt_comm_unique_id: null argument");
    if (int rc = rccl_load()) return rc;
    BT_NCCL(g_rccl.GetUniqueId(static_cast<nccl_unique_id *>(id128)));
    return BT_OK;
}

int bt_comm_init_rank(const void *id128, int rank, int world, bt_comm **out) {
    BT_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world, "bt_comm_init_rank: bad arguments");
    if (int rc = rccl_load()) return rc;
    nccl_unique_id id;
    memcpy(&id, id128, sizeof(id));
    bt_comm *c = new bt_comm();
    c->rank = rank, c->world = world;
    nccl_comm_t nc = nullptr;
    const int r = g_rccl.CommInitRank(&nc, world, id, rank);  // on the calling thread's current device
    if (r != 0) {
        bt_set_error("RCCL: %s", g_rccl.GetErrorString(r));
        delete c;
        return BT_ERR_HIP;
    }
    c->rccl = nc;
    // the collectives' own stream and the events that fence it against the caller's (bt_net_dp_step)
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    for (auto &e : c->ev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        bt_set_error("bt_comm_init_rank: could not create the collective stream / events");
        bt_comm_destroy(c);
        return BT_ERR_HIP;
    }
    *out = c;
    return BT_OK;
}

int bt_comm_from_hooks(const bt_comm_hooks *hooks, int rank, int world, bt_comm **out) {
    BT_REQUIRE(hooks && out && hooks->all_reduce && world >= 1 && rank >= 0 && rank < world,
               "bt_comm_from_hooks: bad arguments");
    BT_REQUIRE(!hooks->reduce_scatter == !hooks->all_gather, "bt_comm_from_hooks: reduce_scatter and all_gather go together");
    bt_comm *c = new bt_comm();
    c->rank = rank, c->world = world, c->hooks = *hooks, c->hooked = true;
    *out = c;
    return BT_OK;
}

int bt_comm_available(void) { return rccl_load() == BT_OK ? 1 : 0; }

int bt_comm_use_own_stream(bt_comm *c) {
    BT_REQUIRE(c, "bt_comm_use_own_stream: null communicator");
    if (c->stream) return BT_OK;  // (a native RCCL communicator has one from the start)
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    for (auto &e : c->ev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    BT_REQUIRE(ok, "bt_comm_use_own_stream: could not create the collective stream / events");
    return BT_OK;
}

int bt_comm_destroy(bt_comm *c) {
    if (!c) return BT_OK;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    int rc = BT_OK;
    if (c->rccl && g_rccl.CommDestroy(c->rccl) != 0) bt_set_error("RCCL: ncclCommDestroy failed"), rc = BT_ERR_HIP;
    delete c;
    return rc;
}

int bt_comm_rank(const bt_comm *c) { return c ? c->rank : 0; }
int bt_comm_world(const bt_comm *c) { return c ? c->world : 1; }

int bt_allreduce_sum_f32(bt_comm *c, void *stream, float *buf, size_t n) {
    BT_REQUIRE(c && buf, "bt_allreduce_sum_f32: null argument");
    if (n == 0) return BT_OK;
    if (c->hooked) return hook_rc(c->hooks.all_reduce(c->hooks.user, buf, n, stream), "all_reduce");
    BT_NCCL(g_rccl.AllReduce(buf, buf, n, NCCL_FLOAT32, NCCL_SUM, static_cast<nccl_comm_t>(c->rccl), bt_stream(stream)));
    return BT_OK;
}

int bt_reduce_scatter_sum_f32(bt_comm *c, void *stream, const float *send, float *recv, size_t n_per_rank) {
    BT_REQUIRE(c && send && recv, "bt_reduce_scatter_sum_f32: null argument");
    if (n_per_rank == 0) return BT_OK;
    if (c->hooked) {
        BT_REQUIRE(c->hooks.reduce_scatter, "bt_reduce_scatter_sum_f32: the communicator has no reduce_scatter hook");
        return hook_rc(c->hooks.reduce_scatter(c->hooks.user, send, recv, n_per_rank, stream), "reduce_scatter");
    }
    BT_NCCL(g_rccl.ReduceScatter(send, recv, n_per_rank, NCCL_FLOAT32, NCCL_SUM, static_cast<nccl_comm_t>(c->rccl),
                                 bt_stream(stream)));
    return BT_OK;
}

int bt_allgather_f32(bt_comm *c, void *stream, const float *send, float *recv, size_t n_per_rank) {
    BT_REQUIRE(c && send && recv, "bt_allgather_f32: null argument");
    if (n_per_rank == 0) return BT_OK;
    if (c->hooked) {
        BT_REQUIRE(c->hooks.all_gather, "bt_allgather_f32: the communicator has no all_gather hook");
        return hook_rc(c->hooks.all_gather(c->hooks.user, send, recv, n_per_rank, stream), "all_gather");
    }
    BT_NCCL(g_rccl.AllGather(send, recv, n_per_rank, NCCL_FLOAT32, static_cast<nccl_comm_t>(c->rccl), bt_stream(stream)));
    return BT_OK;
}

}  // extern "C"
