"""Stand-alone exercise of tests/libfakerccl.so (no product code): `world` host threads, one torch stream each, all-gathers
of alternating sizes, all-to-alls and neighbour send / recv groups with per-rank delays; every result is checked on the host.
    python3 tests/fake_rccl/selftest.py [world] [rounds]
With four or more ranks set GPU_MAX_HW_QUEUES=8: the runtime multiplexes the streams of one priority onto four hardware
queues, and two ranks' torch streams that share one wait for each other's spinning kernel until the stand-in's time limit
(world 4: 3 000 bad results with the default, none with eight queues).  The product's own slab streams (two per rank, of
different priorities, created by the library) have not collided in the 4-rank cases of tests/test_gpu_fake_rccl.py."""
import ctypes as C
import os
import random
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(os.path.join(ROOT, "tests", "libfakerccl.so"))


class Uid(C.Structure):
    _fields_ = [("b", C.c_char * 128)]


L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
L.ncclAllToAll.argtypes = L.ncclAllGather.argtypes
L.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
L.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
L.fake_rccl_errors.restype = C.c_uint32
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
MODE = os.environ.get("SELFTEST_MODE", "torch")  # "raw": hipMalloc / hipFree around every round, a raw non-blocking stream


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    uid = Uid()
    L.ncclGetUniqueId(C.byref(uid))
    bad = []

    def body(r):
        rng = random.Random(r)
        comm = C.c_void_p()
        assert L.ncclCommInitRank(C.byref(comm), world, uid, r) == 0
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for k in range(rounds):
                n = 4 if k % 2 == 0 else (1 << 18) + 64 * (k % 7)  # words per rank: 16 bytes, then ~1 MB
                src = torch.full((n,), r * 1000003 + k, dtype=torch.int32, device="cuda")
                src[n - 1] = k
                dst = torch.zeros((n * world,), dtype=torch.int32, device="cuda")
                if rng.random() < 0.3:
                    time.sleep(rng.random() * 0.003)
                if MODE == "raw":
                    # as the product's collective reads do: buffers freed and allocated between rounds (hipFree synchronises
                    # the device), the result read back with an asynchronous copy on the same stream
                    import numpy as np
                    raw_dst = C.c_void_p()
                    assert hip.hipMalloc(C.byref(raw_dst), n * 4 * world) == 0
                    assert L.ncclAllGather(src.data_ptr(), raw_dst, n * 4, 1, comm, st.cuda_stream) == 0
                    host = np.empty(n * world, np.int32)
                    assert hip.hipMemcpyAsync(host.ctypes.data, raw_dst, n * 4 * world, 2, st.cuda_stream) == 0
                    assert hip.hipStreamSynchronize(st.cuda_stream) == 0
                    hh = host.reshape(world, n)
                    for q in range(world):
                        if not ((hh[q, : n - 1] == q * 1000003 + k).all() and hh[q, n - 1] == k):
                            bad.append(("raw allgather", r, k, q, int(hh[q, 0]), int(hh[q, n - 1])))
                    assert hip.hipFree(raw_dst) == 0
                assert L.ncclAllGather(src.data_ptr(), dst.data_ptr(), n * 4, 1, comm, st.cuda_stream) == 0
                # all-to-all: segment q of the send buffer is addressed to rank q
                a = 64 + 16 * (k % 3)
                a_src = (torch.arange(world, dtype=torch.int32, device="cuda").repeat_interleave(a) * 131 + r * 10007 + k)
                a_dst = torch.zeros((a * world,), dtype=torch.int32, device="cuda")
                assert L.ncclAllToAll(a_src.data_ptr(), a_dst.data_ptr(), a * 4, 1, comm, st.cuda_stream) == 0
                # neighbour exchange, as the slab step does it (one group, both directions)
                m = 1024 + 16 * (k % 5)
                sl = torch.full((m,), r * 7 + k, dtype=torch.int32, device="cuda")
                sr = torch.full((m,), r * 11 + k, dtype=torch.int32, device="cuda")
                rl = torch.zeros((m,), dtype=torch.int32, device="cuda")
                rr = torch.zeros((m,), dtype=torch.int32, device="cuda")
                L.ncclGroupStart()
                if r > 0:
                    L.ncclSend(sl.data_ptr(), m * 4, 1, r - 1, comm, st.cuda_stream)
                    L.ncclRecv(rl.data_ptr(), m * 4, 1, r - 1, comm, st.cuda_stream)
                if r + 1 < world:
                    L.ncclSend(sr.data_ptr(), m * 4, 1, r + 1, comm, st.cuda_stream)
                    L.ncclRecv(rr.data_ptr(), m * 4, 1, r + 1, comm, st.cuda_stream)
                assert L.ncclGroupEnd() == 0
                if k % 10 == 9 or k == rounds - 1:  # (checked now and then: the host runs ahead of the device otherwise)
                    st.synchronize()
                h = dst.cpu().view(world, n)
                for q in range(world):
                    if not (bool((h[q, : n - 1] == q * 1000003 + k).all()) and int(h[q, n - 1]) == k):
                        bad.append(("allgather", r, k, q, int(h[q, 0]), int(h[q, n - 1])))
                ha = a_dst.cpu().view(world, a)
                for q in range(world):  # what rank q addressed to me
                    if not bool((ha[q] == r * 131 + q * 10007 + k).all()):
                        bad.append(("alltoall", r, k, q, int(ha[q, 0])))
                if r > 0 and not bool((rl.cpu() == (r - 1) * 11 + k).all()):
                    bad.append(("recv from left", r, k, int(rl[0])))
                if r + 1 < world and not bool((rr.cpu() == (r + 1) * 7 + k).all()):
                    bad.append(("recv from right", r, k, int(rr[0])))
        st.synchronize()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    print("world %d, %d rounds: %d bad results, error word %d" % (world, rounds, len(bad), L.fake_rccl_errors()))
    for b in bad[:12]:
        print("  ", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
