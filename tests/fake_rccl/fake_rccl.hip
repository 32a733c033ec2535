// fake_rccl.hip -- TEST INFRASTRUCTURE, never shipped and never a default: a stand-in for librccl that lets a ONE-GPU box
// drive the library's RCCL transport (csrc/ws_rccl.cpp) with real peers.  It exports the eleven nccl* symbols that transport
// binds; the "ranks" are host threads of one process on one GPU, and a message travels through a staging buffer in
// device memory.  The transport finds it through WS_RCCL_LIBRARY (tests/test_gpu_fake_rccl.py sets it in a child
// process); without that variable the library loads the real librccl and nothing here exists for it.
//
// What it keeps from the real thing, because the tests are about exactly that:
//   * every call is STREAM-ORDERED and returns at once -- no host wait, no stream synchronisation, no allocation, so a
//     call may sit inside hipStreamBeginCapture .. hipStreamEndCapture and be replayed from a hipGraph;
//   * ranks synchronise ON THE DEVICE, by flags in memory that the copy kernels themselves wait for and raise, as RCCL's
//     kernels do -- never through events shared between two ranks' streams, which
//     two separately captured graphs could not share;
//   * every sequence number lives in device memory and is advanced by the publish kernels, so a replayed graph finds
//     the right one;
//   * ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd are issued together at the group's end (sends first);
//   * communicators are separate worlds: ncclCommSplit creates new channels, a message sent on one communicator can
//     only be received on the same one.
// A message's byte count travels with it and the receiver checks it: wrong peer arithmetic or segment order shows up as
// an error word (fk_error), not as silently exchanged bytes.  Every wait is BOUNDED (FAKE_RCCL_TIMEOUT_MS, default
// 3000): a protocol bug sets the error word and drains, it never leaves a wave spinning on the GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

enum { OK = 0, UNHANDLED = 1, SYSTEM = 2, INTERNAL = 3, INVALID_ARG = 4, INVALID_USAGE = 5 };
enum { ERR_TIMEOUT = 1u, ERR_SIZE = 2u };

// control words of one directed channel (fine-grained device memory, touched with agent-scope atomics only)
enum { C_SENT = 0, C_CONSUMED, C_SSEQ, C_RSEQ, C_BYTES, C_WGDONE, C_WGDONE_R, C_WORDS = 8 };
// control words of the all-gather
enum { A_ARRIVED = 0, A_DEPARTED, A_WORDS = 4 };

__device__ __forceinline__ uint32_t ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Every nccl operation is ONE kernel of NB workgroups (all resident at once; every wait in them is bounded): the
// workgroups wait for the flag themselves, copy, and the last one to finish publishes -- so the only ordering the
// stand-in asks of the runtime is that of whole kernels in one stream.  (Earlier versions used separate wait / copy /
// publish kernels, or hipMemcpyAsync for the payload, and delivered stale bytes now and then: the flags of two ranks
// order their copies with no event or barrier packet the runtime knows of, so nothing made one kernel's stores reach
// memory before the next kernel of the stream raised the flag.)
#define NB 64
#define NT 256

// thread 0: spin until (int)(*flag - target) >= 0; false after a time-out (error word set) or if the error word is set
__device__ bool fk_spin(const uint32_t *flag, uint32_t target, uint32_t *err, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        if ((int32_t)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) return true;
        if (ld(err)) return false;
        if (wall_clock64() - t0 > ticks) {
            __hip_atomic_fetch_or(err, ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}
// every wave's stores performed at agent scope, then: is this the last workgroup of the kernel to get here?
__device__ bool fk_last(uint32_t *wg_done)
{
    __shared__ uint32_t s_last;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t done = __hip_atomic_fetch_add(wg_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = done + 1u == gridDim.x ? 1u : 0u;
        if (s_last) st(wg_done, 0u);
    }
    __syncthreads();
    return s_last != 0u;
}
__device__ void fk_copy(uint32_t *dst, const uint32_t *src, size_t nwords, bool dst_is_staging)
{
    // EVERY wave acquires at agent scope before it reads the staging buffer (thread 0's acquire on the flag invalidates
    // its own wave's view only): without it a wave now and then got the bytes of the same address from two rounds ago
    // out of its XCD's L2 -- the product's look-back scan survives such reads because it retries, a copy does not
    if (!dst_is_staging) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (size_t i = blockIdx.x * (size_t)NT + threadIdx.x; i < nwords; i += (size_t)NB * NT) {
        if (dst_is_staging) st(dst + i, src[i]);
        else dst[i] = ld(src + i);
    }
}

// ncclSend: the staging buffer is free once the receiver has consumed everything sent so far
__global__ void __launch_bounds__(NT) fk_send(const uint32_t *__restrict__ src, uint32_t *staging, size_t nwords, uint32_t *ctl, uint32_t bytes,
                                              uint32_t *err, unsigned long long ticks)
{
    __shared__ uint32_t s_ok;
    if (threadIdx.x == 0) s_ok = fk_spin(ctl + C_CONSUMED, ld(ctl + C_SSEQ), err, ticks) ? 1u : 0u;
    __syncthreads();
    if (s_ok) fk_copy(staging, src, nwords, true);
    if (!fk_last(ctl + C_WGDONE) || threadIdx.x != 0) return;
    const uint32_t sq = ld(ctl + C_SSEQ) + 1u;
    st(ctl + C_SSEQ, sq);
    st(ctl + C_BYTES, bytes);
    __threadfence();
    __hip_atomic_store(ctl + C_SENT, sq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// ncclRecv
__global__ void __launch_bounds__(NT) fk_recv(uint32_t *__restrict__ dst, const uint32_t *staging, size_t nwords, uint32_t *ctl, uint32_t bytes,
                                              uint32_t *err, unsigned long long ticks)
{
    __shared__ uint32_t s_ok;
    if (threadIdx.x == 0) {
        uint32_t ok = fk_spin(ctl + C_SENT, ld(ctl + C_RSEQ) + 1u, err, ticks) ? 1u : 0u;
        if (ok && ld(ctl + C_BYTES) != bytes) __hip_atomic_fetch_or(err, ERR_SIZE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ok = ok;
    }
    __syncthreads();
    if (s_ok) fk_copy(dst, staging, nwords, false);
    if (!fk_last(ctl + C_WGDONE_R) || threadIdx.x != 0) return;
    const uint32_t rq = ld(ctl + C_RSEQ) + 1u;
    st(ctl + C_RSEQ, rq);
    __threadfence();
    __hip_atomic_store(ctl + C_CONSUMED, rq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// ncclAllGather, round k of this rank (k = *seq): every rank has left round k - 1 (departed >= W k) -> my part in ->
// arrived += 1 -> all W have arrived (arrived >= W (k + 1)) and brought the same byte count -> everything out ->
// departed += 1, *seq = k + 1
__global__ void __launch_bounds__(NT) fk_allgather(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, uint32_t *staging, size_t nwords_each,
                                                   uint32_t rank, uint32_t W, uint32_t *ag_ctl, uint32_t *seq, uint32_t *notes, uint32_t *wg_in,
                                                   uint32_t *wg_out, uint32_t bytes, uint32_t *err, unsigned long long ticks)
{
    __shared__ uint32_t s_ok, s_round;
    if (threadIdx.x == 0) {
        s_round = ld(seq);
        s_ok = fk_spin(ag_ctl + A_DEPARTED, s_round * W, err, ticks) ? 1u : 0u;
    }
    __syncthreads();
    const uint32_t round = s_round;
    if (s_ok) fk_copy(staging + (size_t)rank * nwords_each, src, nwords_each, true);
    if (fk_last(wg_in) && threadIdx.x == 0) {
        st(notes + rank, bytes);
        __threadfence();
        __hip_atomic_fetch_add(ag_ctl + A_ARRIVED, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
        uint32_t ok = fk_spin(ag_ctl + A_ARRIVED, round * W + W, err, ticks) ? 1u : 0u;
        for (uint32_t r = 0; ok && r < W; r++)
            if (ld(notes + r) != bytes) __hip_atomic_fetch_or(err, ERR_SIZE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ok = ok;
    }
    __syncthreads();
    if (s_ok) fk_copy(dst, staging, nwords_each * W, false);
    if (!fk_last(wg_out) || threadIdx.x != 0) return;
    st(seq, round + 1u);
    __threadfence();
    __hip_atomic_fetch_add(ag_ctl + A_DEPARTED, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ncclAllToAll (RCCL's own): the all-gather's rounds and counters (both are collectives of the same communicator, issued
// in the same order by every rank), with W x W segments in the staging buffer: rank r's whole send buffer goes in at
// [r W, (r + 1) W), segment (q W + r) comes out as what rank q addressed to rank r.  The byte count is noted with its
// top bit set: a rank that calls the other collective in the same round is a size mismatch.
__global__ void __launch_bounds__(NT) fk_alltoall(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, uint32_t *staging, size_t nwords_each,
                                                  uint32_t rank, uint32_t W, uint32_t *ag_ctl, uint32_t *seq, uint32_t *notes, uint32_t *wg_in,
                                                  uint32_t *wg_out, uint32_t bytes, uint32_t *err, unsigned long long ticks)
{
    __shared__ uint32_t s_ok, s_round;
    const uint32_t note = bytes | 0x80000000u;
    if (threadIdx.x == 0) {
        s_round = ld(seq);
        s_ok = fk_spin(ag_ctl + A_DEPARTED, s_round * W, err, ticks) ? 1u : 0u;
    }
    __syncthreads();
    const uint32_t round = s_round;
    if (s_ok) fk_copy(staging + (size_t)rank * W * nwords_each, src, nwords_each * W, true);
    if (fk_last(wg_in) && threadIdx.x == 0) {
        st(notes + rank, note);
        __threadfence();
        __hip_atomic_fetch_add(ag_ctl + A_ARRIVED, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
        uint32_t ok = fk_spin(ag_ctl + A_ARRIVED, round * W + W, err, ticks) ? 1u : 0u;
        for (uint32_t r = 0; ok && r < W; r++)
            if (ld(notes + r) != note) __hip_atomic_fetch_or(err, ERR_SIZE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ok = ok;
    }
    __syncthreads();
    if (s_ok)
        for (uint32_t q = 0; q < W; q++) fk_copy(dst + (size_t)q * nwords_each, staging + ((size_t)q * W + rank) * nwords_each, nwords_each, false);
    if (!fk_last(wg_out) || threadIdx.x != 0) return;
    st(seq, round + 1u);
    __threadfence();
    __hip_atomic_fetch_add(ag_ctl + A_DEPARTED, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

struct Channel {
    uint32_t *ctl = nullptr;
    char *staging = nullptr;
};

struct Group {
    int world = 0, joined = 0, left = 0;
    std::vector<Channel> to_right, to_left;  // [r]: r -> r + 1 / r -> r - 1
    uint32_t *ag_ctl = nullptr, *ag_seq = nullptr;  // all-gather: arrival / departure counters, one round counter per rank
    uint32_t *ag_cnt = nullptr;                    // ... and the byte count every rank brought to the current round
    uint32_t *ag_wg = nullptr, *ag_wg2 = nullptr;  // ... and workgroup counters per rank (the copy kernels' last-workgroup test)
    char *ag_staging = nullptr;
    uint32_t *err = nullptr;
    size_t chan_cap = 0, ag_cap = 0;
    unsigned long long ticks = 0;
    std::vector<int> splits;                   // per rank: ncclCommSplit calls made on this communicator
    std::map<int, Group *> children;           // by split index
    bool ready = false, failed = false;
};

struct Comm {
    Group *g;
    int rank;
};

std::mutex g_mu;
std::condition_variable g_cv;
std::map<std::string, Group *> g_groups;
std::vector<Group *> g_live;  // every world with a rank in it (top-level and split-off), for the error hook
uint64_t g_next_id = 1;

void *fine(size_t bytes)
{
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    }
    hipMemset(p, 0, bytes);
    return p;
}

size_t env_bytes(const char *name, size_t dflt)
{
    const char *v = getenv(name);
    return v ? (size_t)strtoull(v, nullptr, 10) : dflt;
}

bool group_alloc(Group *g, int world)
{
    g->world = world;
    g->chan_cap = env_bytes("FAKE_RCCL_CHANNEL_BYTES", (size_t)16 << 20);
    g->ag_cap = env_bytes("FAKE_RCCL_ALLGATHER_BYTES", (size_t)512 << 20);  // (an all-to-all stages world x world segments)
    g->ticks = (unsigned long long)env_bytes("FAKE_RCCL_TIMEOUT_MS", 3000) * 100000ull;
    g->to_right.resize(world);
    g->to_left.resize(world);
    g->splits.assign(world, 0);
    g->err = (uint32_t *)fine(64);
    g->ag_ctl = (uint32_t *)fine(A_WORDS * 4);
    g->ag_seq = (uint32_t *)fine((size_t)world * 4);
    g->ag_cnt = (uint32_t *)fine((size_t)world * 4);
    g->ag_wg = (uint32_t *)fine((size_t)world * 4);
    g->ag_wg2 = (uint32_t *)fine((size_t)world * 4);
    if (!g->err || !g->ag_ctl || !g->ag_seq || !g->ag_cnt || !g->ag_wg || !g->ag_wg2 || !(g->ag_staging = (char *)fine(g->ag_cap))) return false;
    for (int r = 0; r < world; r++)
        for (Channel *c : {r + 1 < world ? &g->to_right[r] : nullptr, r > 0 ? &g->to_left[r] : nullptr}) {
            if (!c) continue;
            c->ctl = (uint32_t *)fine(C_WORDS * 4);
            if (!c->ctl || !(c->staging = (char *)fine(g->chan_cap))) return false;
        }
    return hipDeviceSynchronize() == hipSuccess;
}

void group_free(Group *g)
{
    hipDeviceSynchronize();
    for (auto &v : {&g->to_right, &g->to_left})
        for (Channel &c : *v) {
            hipFree(c.ctl);
            hipFree(c.staging);
        }
    hipFree(g->ag_ctl);
    hipFree(g->ag_seq);
    hipFree(g->ag_cnt);
    hipFree(g->ag_wg);
    hipFree(g->ag_wg2);
    hipFree(g->ag_staging);
    hipFree(g->err);
    delete g;
}

// every rank of `g` calls this once: the first one allocates, all leave together (host rendezvous, 60 s)
int group_join(std::unique_lock<std::mutex> &lk, Group *g, int world)
{
    if (g->joined == 0) {
        g->failed = !group_alloc(g, world);
        g->ready = true;
        g_live.push_back(g);
    }
    if (g->world != world) return INVALID_ARG;
    g->joined++;
    g_cv.notify_all();
    if (!g_cv.wait_for(lk, std::chrono::seconds(60), [&] { return g->joined >= g->world; })) return SYSTEM;
    return g->failed ? SYSTEM : OK;
}

struct PendingOp {
    bool send;
    void *ptr;
    size_t bytes;
    int peer;
    Comm *comm;
    hipStream_t stream;
};
thread_local int t_depth = 0;
thread_local std::vector<PendingOp> t_ops;

Channel *channel(Group *g, int from, int to)
{
    if (to == from + 1 && to < g->world) return &g->to_right[from];
    if (to == from - 1 && to >= 0) return &g->to_left[from];
    return nullptr;  // the slab protocol talks to x-neighbours only
}

unsigned long long *calls();
int issue(const PendingOp &op)
{
    {
        std::lock_guard<std::mutex> lk(g_mu);
        calls()[0]++;
    }
    Group *g = op.comm->g;
    const int me = op.comm->rank;
    Channel *c = op.send ? channel(g, me, op.peer) : channel(g, op.peer, me);
    if (!c || op.bytes > g->chan_cap || op.bytes >= ((size_t)1 << 32) || (op.bytes & 3u)) return INVALID_ARG;
    hipStream_t s = op.stream;
    if (op.send)
        hipLaunchKernelGGL(fk_send, dim3(NB), dim3(NT), 0, s, (const uint32_t *)op.ptr, (uint32_t *)c->staging, op.bytes / 4, c->ctl,
                           (uint32_t)op.bytes, g->err, g->ticks);
    else
        hipLaunchKernelGGL(fk_recv, dim3(NB), dim3(NT), 0, s, (uint32_t *)op.ptr, (const uint32_t *)c->staging, op.bytes / 4, c->ctl,
                           (uint32_t)op.bytes, g->err, g->ticks);
    return hipGetLastError() == hipSuccess ? OK : UNHANDLED;
}

int flush_ops()
{
    int rc = OK;
    for (int pass = 0; pass < 2 && rc == OK; pass++)  // sends first: a send never waits for this step's peer
        for (const PendingOp &op : t_ops)
            if (op.send == (pass == 0) && rc == OK) rc = issue(op);
    t_ops.clear();
    return rc;
}

unsigned long long g_calls[4];  // send / recv operations, all-gathers, communicators, all-to-alls
unsigned long long *calls() { return g_calls; }

}  // namespace

extern "C" {

int ncclGetUniqueId(void *out128)
{
    std::lock_guard<std::mutex> lk(g_mu);
    memset(out128, 0, 128);
    snprintf((char *)out128, 128, "fake-rccl-%llu", (unsigned long long)g_next_id++);
    return OK;
}

struct FakeUniqueId {
    char internal[128];
};

int ncclCommInitRank(void **comm, int nranks, FakeUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return INVALID_ARG;
    std::unique_lock<std::mutex> lk(g_mu);
    Group *&g = g_groups[std::string(id.internal, sizeof id.internal)];
    if (!g) g = new Group();
    Group *mine = g;
    const int rc = group_join(lk, mine, nranks);
    if (rc != OK) return rc;
    *comm = new Comm{mine, rank};
    calls()[2]++;
    return OK;
}

// same colour everywhere, ranks keep their order (all the transport asks for): a new world of channels
int ncclCommSplit(void *comm, int color, int key, void **newcomm, void *config)
{
    (void)config;
    Comm *c = static_cast<Comm *>(comm);
    if (!c || !newcomm || color != 0 || key != c->rank) return INVALID_ARG;
    std::unique_lock<std::mutex> lk(g_mu);
    const int idx = c->g->splits[c->rank]++;
    Group *&child = c->g->children[idx];
    if (!child) child = new Group();
    Group *mine = child;
    const int rc = group_join(lk, mine, c->g->world);
    if (rc != OK) return rc;
    *newcomm = new Comm{mine, c->rank};
    calls()[2]++;
    return OK;
}

int ncclCommDestroy(void *comm)
{
    Comm *c = static_cast<Comm *>(comm);
    if (!c) return INVALID_ARG;
    std::unique_lock<std::mutex> lk(g_mu);
    Group *g = c->g;
    delete c;
    if (++g->left == g->world) {
        for (auto it = g_live.begin(); it != g_live.end(); ++it)
            if (*it == g) {
                g_live.erase(it);
                break;
            }
        for (auto it = g_groups.begin(); it != g_groups.end(); ++it)
            if (it->second == g) {
                g_groups.erase(it);
                break;
            }
        // (a split's parent keeps the child in its map: children are freed when they are left themselves)
        for (auto &kv : g_groups)
            for (auto it = kv.second->children.begin(); it != kv.second->children.end(); ++it)
                if (it->second == g) {
                    kv.second->children.erase(it);
                    break;
                }
        lk.unlock();
        group_free(g);
    }
    return OK;
}

int ncclGroupStart(void)
{
    t_depth++;
    return OK;
}

int ncclGroupEnd(void)
{
    if (t_depth <= 0) return INVALID_USAGE;
    if (--t_depth > 0) return OK;
    return flush_ops();
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    if (dtype != 1 || !comm) return INVALID_ARG;  // ncclUint8: all the transport uses
    t_ops.push_back({true, const_cast<void *>(buf), count, peer, static_cast<Comm *>(comm), stream});
    return t_depth > 0 ? OK : flush_ops();
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    if (dtype != 1 || !comm) return INVALID_ARG;
    t_ops.push_back({false, buf, count, peer, static_cast<Comm *>(comm), stream});
    return t_depth > 0 ? OK : flush_ops();
}

int ncclAllGather(const void *sendbuf, void *recvbuf, size_t count, int dtype, void *comm, hipStream_t s)
{
    Comm *c = static_cast<Comm *>(comm);
    if (dtype != 1 || !c) return INVALID_ARG;
    Group *g = c->g;
    const uint32_t W = (uint32_t)g->world;
    if (count * W > g->ag_cap || (count & 3u)) return INVALID_ARG;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        calls()[1]++;
    }
    hipLaunchKernelGGL(fk_allgather, dim3(NB), dim3(NT), 0, s, (const uint32_t *)sendbuf, (uint32_t *)recvbuf, (uint32_t *)g->ag_staging, count / 4,
                       (uint32_t)c->rank, W, g->ag_ctl, g->ag_seq + c->rank, g->ag_cnt, g->ag_wg + c->rank, g->ag_wg2 + c->rank, (uint32_t)count,
                       g->err, g->ticks);
    return hipGetLastError() == hipSuccess ? OK : UNHANDLED;
}

int ncclAllToAll(const void *sendbuf, void *recvbuf, size_t count, int dtype, void *comm, hipStream_t s)
{
    Comm *c = static_cast<Comm *>(comm);
    if (dtype != 1 || !c) return INVALID_ARG;
    Group *g = c->g;
    const uint32_t W = (uint32_t)g->world;
    if (count * W * W > g->ag_cap || (count & 3u) || count >= ((size_t)1 << 31)) return INVALID_ARG;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        calls()[3]++;
    }
    hipLaunchKernelGGL(fk_alltoall, dim3(NB), dim3(NT), 0, s, (const uint32_t *)sendbuf, (uint32_t *)recvbuf, (uint32_t *)g->ag_staging, count / 4,
                       (uint32_t)c->rank, W, g->ag_ctl, g->ag_seq + c->rank, g->ag_cnt, g->ag_wg + c->rank, g->ag_wg2 + c->rank, (uint32_t)count,
                       g->err, g->ticks);
    return hipGetLastError() == hipSuccess ? OK : UNHANDLED;
}

const char *ncclGetErrorString(int rc)
{
    switch (rc) {
    case OK: return "no error";
    case INVALID_ARG: return "invalid argument (fake rccl: not an x-neighbour, message larger than the staging buffer, ...)";
    case INVALID_USAGE: return "invalid usage";
    case SYSTEM: return "system error (fake rccl: rendezvous timed out or allocation failed)";
    default: return "unhandled error";
    }
}

// test hook (not an nccl symbol): the error words of every live world OR-ed together -- 0, or ERR_TIMEOUT (1) /
// ERR_SIZE (2) bits.  Synchronises the device.
uint32_t fake_rccl_errors(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    hipDeviceSynchronize();
    uint32_t all = 0;
    for (Group *g : g_live) {
        uint32_t v = 0xFFFFFFFFu;
        if (g->err) hipMemcpy(&v, g->err, 4, hipMemcpyDeviceToHost);
        all |= v;
    }
    return all;
}

// test hook: calls served since the library was loaded (proof that the transport really went through here)
void fake_rccl_calls(unsigned long long out[4])
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < 4; i++) out[i] = g_calls[i];
}

}  // extern "C"
