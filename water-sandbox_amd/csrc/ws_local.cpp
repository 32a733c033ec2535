// ws_local.cpp -- a ws_transport for several slabs inside ONE process (one host thread per slab, one or several GPUs):
// the "single-process multi-device first" tier of SURVEY.md section 4, and what lets a one-GPU box drive the whole slab
// protocol -- migration, halos, the global reads -- through the C ABI from C++ (host/frame_loop.cpp).  Plain
// device-to-device copies between the slabs' message buffers, rendezvous on a host barrier.  Not a performance path:
// every call synchronises its stream (so a step using it cannot be captured into a hipGraph; the RCCL transport can).
#include "wsfluid.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <vector>

namespace {

struct LocalHub {
    uint32_t world = 1;
    std::mutex m;
    std::condition_variable cv;
    uint32_t arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    // what each rank offers in the call being made
    struct Offer {
        void *const *send_ptr = nullptr;
        const uint64_t *send_bytes = nullptr;
        uint32_t nseg = 0;
        const void *gather_ptr = nullptr;
    };
    std::vector<Offer> offer;

    // all `world` threads meet here; false = somebody did not arrive in time (the hub is then broken for good)
    bool barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
            return true;
        }
        if (!cv.wait_for(lk, std::chrono::seconds(300), [&] { return generation != gen || broken; })) {
            broken = true;  // a rank left the collective alone: fail everybody instead of hanging
            cv.notify_all();
            return false;
        }
        return !broken;
    }
    // a rank that cannot go on (a HIP error of its own) breaks the hub for everybody at once: its peers fail at their
    // next barrier instead of waiting out the time limit for a rank that will never arrive
    void fail_all()
    {
        std::lock_guard<std::mutex> lk(m);
        broken = true;
        cv.notify_all();
    }
};

struct LocalTransport {
    LocalHub *hub;
    uint32_t rank;
};

int local_sendrecv(void *ctx, uint32_t nseg, void *const send_ptr[], const uint64_t send_bytes[], void *const recv_ptr[],
                   const uint64_t recv_bytes[], void *stream)
{
    LocalTransport *t = static_cast<LocalTransport *>(ctx);
    LocalHub *hub = t->hub;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipStreamSynchronize(s) != hipSuccess) {  // my boundary data is final before anybody reads it
        hub->fail_all();
        return 1;
    }
    hub->offer[t->rank].send_ptr = send_ptr;
    hub->offer[t->rank].send_bytes = send_bytes;
    hub->offer[t->rank].nseg = nseg;
    if (!hub->barrier()) return 2;
    int rc = 0;
    for (uint32_t i = 0; i < 2 * nseg && !rc; i++) {
        if (!recv_bytes[i]) continue;
        const uint32_t d = i % 2;  // my left neighbour's right-going segment is entry i + 1 of its lists, and vice versa
        const uint32_t peer = d == 0 ? t->rank - 1 : t->rank + 1;
        const uint32_t j = d == 0 ? i + 1 : i - 1;
        if (peer >= hub->world) {  // (rank 0's left peer is UINT32_MAX: checked BEFORE the offer table is indexed)
            rc = 3;
            break;
        }
        const LocalHub::Offer &o = hub->offer[peer];
        if (o.nseg != nseg || o.send_bytes[j] != recv_bytes[i]) {
            rc = 3;  // the two ends disagree about a message size
            break;
        }
        if (hipMemcpyAsync(recv_ptr[i], o.send_ptr[j], (size_t)recv_bytes[i], hipMemcpyDeviceToDevice, s) != hipSuccess) rc = 4;
    }
    if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = 5;
    if (!hub->barrier()) return 2;  // nobody reuses a send range before every reader is done
    return rc;
}

int local_allgather(void *ctx, const void *send_ptr, void *recv_ptr, uint64_t bytes_each, void *stream)
{
    LocalTransport *t = static_cast<LocalTransport *>(ctx);
    LocalHub *hub = t->hub;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipStreamSynchronize(s) != hipSuccess) {
        hub->fail_all();
        return 1;
    }
    hub->offer[t->rank].gather_ptr = send_ptr;
    if (!hub->barrier()) return 2;
    int rc = 0;
    for (uint32_t r = 0; r < hub->world && !rc; r++)
        if (hipMemcpyAsync(static_cast<char *>(recv_ptr) + (size_t)r * bytes_each, hub->offer[r].gather_ptr, (size_t)bytes_each,
                           hipMemcpyDeviceToDevice, s) != hipSuccess)
            rc = 4;
    if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = 5;
    if (!hub->barrier()) return 2;
    return rc;
}

// segment r of every rank's send buffer is what that rank addresses to rank r
int local_alltoall(void *ctx, const void *send_ptr, void *recv_ptr, uint64_t bytes_each, void *stream)
{
    LocalTransport *t = static_cast<LocalTransport *>(ctx);
    LocalHub *hub = t->hub;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipStreamSynchronize(s) != hipSuccess) {
        hub->fail_all();
        return 1;
    }
    hub->offer[t->rank].gather_ptr = send_ptr;
    if (!hub->barrier()) return 2;
    int rc = 0;
    for (uint32_t r = 0; r < hub->world && !rc; r++)
        if (hipMemcpyAsync(static_cast<char *>(recv_ptr) + (size_t)r * bytes_each,
                           static_cast<const char *>(hub->offer[r].gather_ptr) + (size_t)t->rank * bytes_each, (size_t)bytes_each,
                           hipMemcpyDeviceToDevice, s) != hipSuccess)
            rc = 4;
    if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = 5;
    if (!hub->barrier()) return 2;
    return rc;
}

}  // namespace

extern "C" {

ws_status ws_local_hub_create(uint32_t world_size, void **hub_out)
{
    if (!hub_out || world_size == 0) return WS_ERR_INVALID_ARG;
    LocalHub *hub = new (std::nothrow) LocalHub();
    if (!hub) return WS_ERR_OUT_OF_MEMORY;
    hub->world = world_size;
    hub->offer.resize(world_size);
    *hub_out = hub;
    return WS_OK;
}

void ws_local_hub_destroy(void *hub) { delete static_cast<LocalHub *>(hub); }

ws_status ws_local_transport_create(void *hub, uint32_t rank, ws_transport *out)
{
    LocalHub *h = static_cast<LocalHub *>(hub);
    if (!h || !out || rank >= h->world) return WS_ERR_INVALID_ARG;
    LocalTransport *t = new (std::nothrow) LocalTransport{h, rank};
    if (!t) return WS_ERR_OUT_OF_MEMORY;
    out->struct_size = sizeof(ws_transport);
    out->ctx = t;
    out->sendrecv = local_sendrecv;
    out->allgather_dev = local_allgather;
    out->alltoall_dev = local_alltoall;
    return WS_OK;
}

void ws_local_transport_destroy(ws_transport *t)
{
    if (!t || !t->ctx || t->sendrecv != local_sendrecv) return;
    delete static_cast<LocalTransport *>(t->ctx);
    t->ctx = nullptr;
}

}  // extern "C"
