// ws_rccl.cpp -- a ws_transport made of RCCL calls issued by the library itself (no host-language callbacks in
// the step): ncclSend / ncclRecv groups with the two x-neighbours over xGMI, ncclAllToAll for the per-destination far
// messages (whose headers are the status words), ncclAllGather for the host's collective reads.
// RCCL is loaded at run time (dlopen), so the library has no link-time dependency on it and single-GPU use
// never touches it.  The host distributes rank 0's unique id (128 bytes) by whatever means it has.
#include "wsfluid.h"
#include "ws_devhooks.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>

namespace {

struct NcclUniqueId {
    char internal[WS_RCCL_UNIQUE_ID_BYTES];
};
typedef void *NcclComm;
enum { NCCL_UINT8 = 1 };

struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*CommSplit)(NcclComm, int, int, NcclComm *, void *) = nullptr;  // optional (RCCL >= 2.18)
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*AllToAll)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;  // RCCL's own (not in NCCL)
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};

RcclApi g_api;
std::once_flag g_once;

template <class F>
bool sym(F &fn, const char *name)
{
    fn = reinterpret_cast<F>(dlsym(g_api.lib, name));
    if (!fn) g_api.error = std::string("librccl: missing symbol ") + name;
    return fn != nullptr;
}

bool load_rccl()
{
    std::call_once(g_once, [] {
        // Developer builds only (-DWS_DEV_HOOKS): WS_RCCL_LIBRARY names another library with RCCL's interface, by path
        // -- the test suite's one-process stand-in, tests/fake_rccl/.  The product library looks for librccl alone: no
        // environment variable can put another collective library under a production process.
        const char *override_path = WS_DEV_ENV("WS_RCCL_LIBRARY");
        if (override_path && *override_path) {
            g_api.lib = dlopen(override_path, RTLD_NOW | RTLD_GLOBAL);
        } else {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                g_api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (g_api.lib) break;
            }
        }
        if (!g_api.lib) {
            g_api.error = std::string("cannot load librccl: ") + dlerror();
            return;
        }
        const bool ok = sym(g_api.GetUniqueId, "ncclGetUniqueId") && sym(g_api.CommInitRank, "ncclCommInitRank") &&
                        sym(g_api.CommDestroy, "ncclCommDestroy") && sym(g_api.GroupStart, "ncclGroupStart") &&
                        sym(g_api.GroupEnd, "ncclGroupEnd") && sym(g_api.Send, "ncclSend") && sym(g_api.Recv, "ncclRecv") &&
                        sym(g_api.AllGather, "ncclAllGather") && sym(g_api.AllToAll, "ncclAllToAll") &&
                        sym(g_api.GetErrorString, "ncclGetErrorString");
        if (!ok) {
            dlclose(g_api.lib);
            g_api.lib = nullptr;
            return;
        }
        g_api.CommSplit = reinterpret_cast<decltype(g_api.CommSplit)>(dlsym(g_api.lib, "ncclCommSplit"));
    });
    return g_api.lib != nullptr;
}

// The slab step drives the transport from TWO streams: migration + the small all-gather (and the host's global reads)
// on the handle's stream, the two halos on its communication stream, concurrently.  Operations of ONE communicator
// issued on two streams would depend on RCCL serialising them in host issue order on every rank alike, and would
// couple the streams again (the next step's migration queuing behind halo B).  So the transport owns two
// communicators -- the second one split off the first (ncclCommSplit: no second unique-id exchange) -- and every
// stream keeps to its own: the first stream it sees gets comm[0], any other stream comm[1].
// An RCCL without ncclCommSplit (or, in developer builds, WS_RCCL_SINGLE_COMM=1) keeps everything on comm[0], the pre-round-3 behaviour.
struct RcclTransport {
    NcclComm comm[2] = {nullptr, nullptr};
    hipStream_t first_stream = nullptr;
    bool have_first = false;
    int rank = 0, world = 1, device = 0;
    std::string error;
    NcclComm on(hipStream_t s)
    {
        if (!have_first) {
            first_stream = s;
            have_first = true;
        }
        return (s == first_stream || !comm[1]) ? comm[0] : comm[1];
    }
};

int check(RcclTransport *t, int rc, const char *what)
{
    if (rc == 0) return 0;
    t->error = std::string(what) + ": " + g_api.GetErrorString(rc);
    fprintf(stderr, "wsfluid rccl transport (rank %d): %s\n", t->rank, t->error.c_str());
    return 1;
}

// ws_transport::sendrecv: segment k, direction d (0 = rank - 1, 1 = rank + 1) at index 2k + d; one RCCL group
int rccl_sendrecv(void *ctx, uint32_t nseg, void *const send_ptr[], const uint64_t send_bytes[], void *const recv_ptr[],
                  const uint64_t recv_bytes[], void *stream)
{
    RcclTransport *t = static_cast<RcclTransport *>(ctx);
    hipStream_t s = static_cast<hipStream_t>(stream);
    bool any = false;
    for (uint32_t i = 0; i < 2 * nseg; i++) any = any || send_bytes[i] || recv_bytes[i];
    if (!any) return 0;
    NcclComm comm = t->on(s);
    if (check(t, g_api.GroupStart(), "ncclGroupStart")) return 1;
    int rc = 0;
    for (uint32_t i = 0; i < 2 * nseg && !rc; i++) {
        const int peer = (i % 2 == 0) ? t->rank - 1 : t->rank + 1;
        if (send_bytes[i]) rc = g_api.Send(send_ptr[i], (size_t)send_bytes[i], NCCL_UINT8, peer, comm, s);
        if (!rc && recv_bytes[i]) rc = g_api.Recv(recv_ptr[i], (size_t)recv_bytes[i], NCCL_UINT8, peer, comm, s);
    }
    const int rc_end = g_api.GroupEnd();
    if (check(t, rc, "ncclSend/ncclRecv")) return 1;
    return check(t, rc_end, "ncclGroupEnd");
}

int rccl_allgather(void *ctx, const void *send_ptr, void *recv_ptr, uint64_t bytes_each, void *stream)
{
    RcclTransport *t = static_cast<RcclTransport *>(ctx);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return check(t, g_api.AllGather(send_ptr, recv_ptr, (size_t)bytes_each, NCCL_UINT8, t->on(s), s), "ncclAllGather");
}

// ws_transport::alltoall_dev: segment r of the send buffer to rank r, its segment for this rank into segment r here
int rccl_alltoall(void *ctx, const void *send_ptr, void *recv_ptr, uint64_t bytes_each, void *stream)
{
    RcclTransport *t = static_cast<RcclTransport *>(ctx);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return check(t, g_api.AllToAll(send_ptr, recv_ptr, (size_t)bytes_each, NCCL_UINT8, t->on(s), s), "ncclAllToAll");
}

}  // namespace

extern "C" {

ws_status ws_rccl_unique_id(void *out128)
{
    if (!out128) return WS_ERR_INVALID_ARG;
    if (!load_rccl()) return WS_ERR_COMM;
    NcclUniqueId id;
    if (g_api.GetUniqueId(&id) != 0) return WS_ERR_COMM;
    memcpy(out128, id.internal, sizeof id.internal);
    return WS_OK;
}

ws_status ws_rccl_transport_create(const void *unique_id, uint32_t rank, uint32_t world_size, int32_t device,
                                   ws_transport *out)
{
    if (!unique_id || !out || world_size == 0 || rank >= world_size) return WS_ERR_INVALID_ARG;
    if (!load_rccl()) return WS_ERR_COMM;
    {
        // (this may be the process's first HIP call, right after ncclGetUniqueId: say what HIP says if it refuses)
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e == hipSuccess && device >= 0 && device < ndev) e = hipSetDevice(device);
        else if (e == hipSuccess) e = hipErrorInvalidDevice;
        if (e != hipSuccess) {
            g_api.error = std::string("hipSetDevice(") + std::to_string(device) + ") of " + std::to_string(ndev) + " devices: " + hipGetErrorString(e);
            return WS_ERR_NO_DEVICE;
        }
    }
    RcclTransport *t = new RcclTransport();
    t->rank = (int)rank;
    t->world = (int)world_size;
    t->device = device;
    NcclUniqueId id;
    memcpy(id.internal, unique_id, sizeof id.internal);
    if (check(t, g_api.CommInitRank(&t->comm[0], t->world, id, t->rank), "ncclCommInitRank")) {
        g_api.error = t->error;
        delete t;
        return WS_ERR_COMM;
    }
    const char *single = WS_DEV_ENV("WS_RCCL_SINGLE_COMM");  // (developer builds: both streams on one communicator)
    if (g_api.CommSplit && !(single && atoi(single) != 0)) {
        // collective over the first communicator; same colour everywhere, ranks keep their order
        // (not fatal: without a second communicator both streams share the first one, as before round 3)
        if (check(t, g_api.CommSplit(t->comm[0], 0, t->rank, &t->comm[1], nullptr), "ncclCommSplit")) {
            g_api.error = t->error;
            t->comm[1] = nullptr;
        }
    }
    out->struct_size = sizeof(ws_transport);
    out->ctx = t;
    out->sendrecv = rccl_sendrecv;
    out->allgather_dev = rccl_allgather;
    out->alltoall_dev = rccl_alltoall;
    return WS_OK;
}

void ws_rccl_transport_destroy(ws_transport *t)
{
    if (!t || !t->ctx || t->sendrecv != rccl_sendrecv) return;
    RcclTransport *r = static_cast<RcclTransport *>(t->ctx);
    hipSetDevice(r->device);
    if (r->comm[1]) g_api.CommDestroy(r->comm[1]);
    if (r->comm[0]) g_api.CommDestroy(r->comm[0]);
    delete r;
    t->ctx = nullptr;
}

const char *ws_rccl_last_error(void) { return g_api.error.c_str(); }

// The handle created with this transport names its own (compute) stream: that one keeps to communicator 0, its
// communication stream to communicator 1 -- whatever streams an earlier handle on the same transport used.
void ws_rccl_transport_bind_stream(const ws_transport *t, void *stream)
{
    if (!t || !t->ctx || t->sendrecv != rccl_sendrecv) return;
    RcclTransport *r = static_cast<RcclTransport *>(t->ctx);
    r->first_stream = static_cast<hipStream_t>(stream);
    r->have_first = true;
}

// how many communicators the transport drives (2 = one per stream of the slab step); diagnostics / tests
uint32_t ws_rccl_transport_communicators(const ws_transport *t)
{
    if (!t || !t->ctx || t->sendrecv != rccl_sendrecv) return 0;
    const RcclTransport *r = static_cast<const RcclTransport *>(t->ctx);
    return r->comm[1] ? 2u : 1u;
}

}  // extern "C"
