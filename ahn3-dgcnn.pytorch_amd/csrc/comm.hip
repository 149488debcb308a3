// comm.hip -- the exchange step behind the C ABI: one RCCL communicator per process (one process per
// GPU), one all-reduce per training step of the flat fp32 gradient buffer, a broadcast for the initial parameters.
// The reference has no distributed code (train.py:110-111 only prints the device count).
//
// RCCL is bound at RUN time (dlopen), not at link time: a PyTorch process already holds a librccl (torch/lib/librccl.so,
// SONAME librccl.so.1) and two copies of the library in one process would each keep their own state; dlopen by SONAME
// returns the copy that is already mapped, and falls back to the system's ROCm copy in a process without PyTorch.
// Only the handful of entry points used here are resolved; the types come from <rccl/rccl.h>.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "common.h"
#include "dgcnn_hip.h"

struct dgcnn_comm {
    ncclComm_t comm;
    int world, rank, device;
};

namespace dgcnn {
namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
};

// 0 = ok; the library stays mapped for the life of the process.  (Called from the thread that drives the GPU: one process
// per GPU, one thread per process is the deployment model; the first call is not guarded against a concurrent one.)
int rccl_api(const RcclApi **out)
{
    static RcclApi api;
    static int state = 0;  // 0 = not tried, 1 = ok, -1 = failed
    static char why[256] = "";  // the FIRST failure's message (dlerror() hands its string over once), reported on every later call
    if (state == 0) {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // the copy the process already holds (PyTorch's)
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        bool ok = h != nullptr;
        if (!ok) {
            const char *e = dlerror();
            snprintf(why, sizeof(why), "%s", e ? e : "dlopen failed");
        } else {
            api.lib = h;
#define DG_SYM(field, name)                                                                      \
    do {                                                                                         \
        if (ok && (*(void **)(&api.field) = dlsym(h, name)) == nullptr) {                        \
            ok = false;                                                                          \
            snprintf(why, sizeof(why), "symbol %s is missing", name);                            \
        }                                                                                        \
    } while (0)
            DG_SYM(GetUniqueId, "ncclGetUniqueId");
            DG_SYM(CommInitRank, "ncclCommInitRank");
            DG_SYM(CommDestroy, "ncclCommDestroy");
            DG_SYM(AllReduce, "ncclAllReduce");
            DG_SYM(Broadcast, "ncclBroadcast");
            DG_SYM(GetErrorString, "ncclGetErrorString");
            DG_SYM(GetVersion, "ncclGetVersion");
#undef DG_SYM
            if (!ok) {
                dlclose(h);
                api.lib = nullptr;
            }
        }
        state = ok ? 1 : -1;
    }
    if (state != 1) DG_FAIL("dgcnn_comm: librccl.so.1 could not be loaded (%s)", why);
    *out = &api;
    return 0;
}

}  // namespace
}  // namespace dgcnn

using namespace dgcnn;

#define DG_RCCL(call, what)                                                          \
    do {                                                                             \
        ncclResult_t r_ = (call);                                                    \
        if (r_ != ncclSuccess) DG_FAIL("%s: %s", what, api->GetErrorString(r_));     \
    } while (0)

extern "C" int dgcnn_comm_rccl_version(void)
{
    const RcclApi *api;
    if (rccl_api(&api)) return -1;
    int v = 0;
    return api->GetVersion(&v) == ncclSuccess ? v : -1;
}

extern "C" int dgcnn_comm_unique_id(uint8_t *id)
{
    DG_REQUIRE(id, "dgcnn_comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == DGCNN_COMM_ID_BYTES, "ncclUniqueId size");
    const RcclApi *api;
    if (rccl_api(&api)) return 1;
    ncclUniqueId u;
    DG_RCCL(api->GetUniqueId(&u), "ncclGetUniqueId");
    memcpy(id, &u, sizeof(u));
    return 0;
}

extern "C" int dgcnn_comm_init(dgcnn_comm **comm, const uint8_t *id, int world, int rank)
{
    DG_REQUIRE(comm && id, "dgcnn_comm_init: null pointer");
    DG_REQUIRE(world >= 1 && rank >= 0 && rank < world, "dgcnn_comm_init: bad rank %d of %d", rank, world);
    const RcclApi *api;
    if (rccl_api(&api)) return 1;
    int dev = -1;
    DG_REQUIRE(hipGetDevice(&dev) == hipSuccess, "dgcnn_comm_init: no current HIP device");
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    DG_RCCL(api->CommInitRank(&c, world, u, rank), "ncclCommInitRank");  // (collective: every rank calls it with the same id)
    dgcnn_comm *out = new dgcnn_comm{c, world, rank, dev};
    *comm = out;
    return 0;
}

extern "C" int dgcnn_comm_destroy(dgcnn_comm *comm)
{
    if (!comm) return 0;
    const RcclApi *api;
    if (rccl_api(&api)) return 1;
    const ncclResult_t r = api->CommDestroy(comm->comm);
    delete comm;
    if (r != ncclSuccess) DG_FAIL("ncclCommDestroy: %s", api->GetErrorString(r));
    return 0;
}

extern "C" int dgcnn_comm_world(const dgcnn_comm *comm) { return comm ? comm->world : 0; }
extern "C" int dgcnn_comm_rank(const dgcnn_comm *comm) { return comm ? comm->rank : -1; }

extern "C" int dgcnn_comm_allreduce_f32(dgcnn_comm *comm, float *buf, size_t count, int average, void *stream)
{
    DG_REQUIRE(comm && buf, "dgcnn_comm_allreduce_f32: null pointer");
    const RcclApi *api;
    if (rccl_api(&api)) return 1;
    if (count == 0) return 0;
    DG_RCCL(api->AllReduce(buf, buf, count, ncclFloat32, average ? ncclAvg : ncclSum, comm->comm, (hipStream_t)stream),
            "ncclAllReduce");
    return 0;
}

extern "C" int dgcnn_comm_broadcast_f32(dgcnn_comm *comm, float *buf, size_t count, int root, void *stream)
{
    DG_REQUIRE(comm && buf, "dgcnn_comm_broadcast_f32: null pointer");
    DG_REQUIRE(root >= 0 && root < comm->world, "dgcnn_comm_broadcast_f32: root %d outside the communicator", root);
    const RcclApi *api;
    if (rccl_api(&api)) return 1;
    if (count == 0) return 0;
    DG_RCCL(api->Broadcast(buf, buf, count, ncclFloat32, root, comm->comm, (hipStream_t)stream), "ncclBroadcast");
    return 0;
}
