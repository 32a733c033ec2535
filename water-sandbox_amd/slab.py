"""Multi-GPU slabs over the C ABI (include/wsfluid.h, "multi-GPU" section).

Host-side plumbing only: the slab protocol itself (migration -> sort -> halo A -> K4 -> halo B ->
K5+K6, fixed-capacity messages with their counts in headers, no host round trip) lives in the
library (csrc/ws_slab.inc); this module supplies the transports it calls and a thin worker class.

Transports:
  NativeRcclTransport the library's own RCCL transport (csrc/ws_rccl.cpp): ncclSend / ncclRecv groups
                      with the two x-neighbours + ncclAllToAll / ncclAllGather, issued from C++ -- no Python in the
                      step.  This is what bench.py uses.
  TorchDistTransport  the same three callbacks on torch.distributed (RCCL or, in the tests, gloo).
  LoopbackHub         several slabs inside ONE process on ONE GPU, one host thread per slab,
                      device-to-device copies.  Lets the one-GPU test box exercise the whole slab
                      protocol (ghost indexing, migration, bit-exactness against one GPU).
"""
import contextlib
import ctypes as C
import os
import threading

import numpy as np

from . import fluid

SENDRECV_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                         C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p)
ALLGATHER_DEV_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


class WsTransport(C.Structure):
    _fields_ = [("struct_size", C.c_uint64), ("ctx", C.c_void_p), ("sendrecv", SENDRECV_T), ("allgather_dev", ALLGATHER_DEV_T),
                ("alltoall_dev", ALLGATHER_DEV_T)]  # (the same signature)


def assign(params, positions, world_size, library=None):
    """ws_slab_assign: the slab (rank) that owns each position at t = 0."""
    L = library if library is not None else fluid.load_library()
    positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    out = np.empty(positions.shape[0], np.uint32)
    L.ws_slab_assign.argtypes = [C.POINTER(fluid.WsParams), C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    st = L.ws_slab_assign(C.byref(params), positions.ctypes.data, positions.shape[0], world_size, out.ctypes.data)
    if st != 0:
        raise fluid.WsError(st, (L.ws_last_error(None) or b"").decode())
    return out


class _TransportBase:
    """Wraps two Python methods as the C callback table; keeps the thunks alive."""

    def __init__(self):
        self.error = None
        self._thunks = (SENDRECV_T(self._c_sendrecv), ALLGATHER_DEV_T(self._c_allgather_dev), ALLGATHER_DEV_T(self._c_alltoall_dev))
        self.struct = WsTransport(C.sizeof(WsTransport), None, *self._thunks)

    def _guard(self, fn, *a):
        try:
            fn(*a)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            self.error = e
            return 1

    def _c_sendrecv(self, ctx, nseg, sp, sb, rp, rb, stream):
        m = 2 * nseg  # entry 2k + d: segment k, direction d (0 = left neighbour, 1 = right neighbour)
        return self._guard(self.sendrecv, [sp[i] for i in range(m)], [sb[i] for i in range(m)],
                           [rp[i] for i in range(m)], [rb[i] for i in range(m)], stream)

    def _c_allgather_dev(self, ctx, sp, rp, nbytes, stream):
        return self._guard(self.allgather_dev, sp, rp, nbytes, stream)

    def _c_alltoall_dev(self, ctx, sp, rp, nbytes, stream):
        return self._guard(self.alltoall_dev, sp, rp, nbytes, stream)


# ------------------------------------------------------------------------------------------
class _DevMem:
    """A device byte range exposed through __cuda_array_interface__ so torch can alias it."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


class TorchDistTransport(_TransportBase):
    """One process per GPU.  data_group: backend "nccl" (RCCL over xGMI) for device buffers;
    ctrl_group: a gloo group for the handful of host control words per step."""

    def __init__(self, rank, world, device, data_group=None, ctrl_group=None):
        super().__init__()
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", device)
        self.data_group, self.ctrl_group = data_group, ctrl_group
        # RCCL ("nccl") calls are ordered on the current stream; gloo (tests) is not: fence by hand
        self.stream_ordered = dist.get_backend(data_group) == "nccl"
        self._ext = {}
        # torch's current stream when the transport was made: the stream the handle is created on
        self._home_stream = torch.cuda.current_stream(self.device).cuda_stream if torch.cuda.is_available() else 0

    def _fence(self):
        if not self.stream_ordered:
            self.torch.cuda.synchronize(self.device)

    def _on(self, stream):
        """Context in which torch's current stream is the HIP stream the library names: RCCL calls are ordered on
        the current stream, and the library issues its halos on a second stream while the first one computes."""
        if not (self.stream_ordered and stream) or stream == self._home_stream:
            return contextlib.nullcontext()  # gloo (fenced by hand), or already the current stream
        ext = self._ext.get(stream)
        if ext is None:
            ext = self._ext[stream] = self.torch.cuda.ExternalStream(stream, device=self.device)
        return self.torch.cuda.stream(ext)

    def _tensor(self, ptr, nbytes):
        return self.torch.as_tensor(_DevMem(ptr, nbytes), device=self.device)

    def sendrecv(self, sp, sb, rp, rb, stream):
        dist = self.dist
        ops, keep = [], []
        for i in range(len(sb)):  # segments in order; every pair of neighbours lists them identically
            peer = self.rank - 1 if i % 2 == 0 else self.rank + 1
            if sb[i]:
                t = self._tensor(sp[i], sb[i]); keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, peer, group=self.data_group))
            if rb[i]:
                t = self._tensor(rp[i], rb[i]); keep.append(t)
                ops.append(dist.P2POp(dist.irecv, t, peer, group=self.data_group))
        if ops:
            self._fence()
            with self._on(stream):
                for req in dist.batch_isend_irecv(ops):
                    req.wait()  # stream-ordered on the current stream for the nccl backend
            self._fence()

    def allgather_dev(self, sp, rp, nbytes, stream):
        src = self._tensor(sp, nbytes)
        dst = self._tensor(rp, nbytes * self.world)
        if self.stream_ordered:
            with self._on(stream):
                self.dist.all_gather_into_tensor(dst, src, group=self.data_group)
        else:  # gloo (tests): list form, fenced
            self._fence()
            self.dist.all_gather(list(dst.view(self.world, nbytes).unbind(0)), src, group=self.data_group)
            self._fence()

    def alltoall_dev(self, sp, rp, nbytes, stream):
        src = self._tensor(sp, nbytes * self.world)
        dst = self._tensor(rp, nbytes * self.world)
        if self.stream_ordered:
            with self._on(stream):
                self.dist.all_to_all_single(dst, src, group=self.data_group)
        else:  # gloo (tests): through host memory, fenced
            self._fence()
            host = self.torch.empty(nbytes * self.world, dtype=self.torch.uint8)
            self.dist.all_to_all_single(host, src.cpu(), group=self.data_group)
            dst.copy_(host)
            self._fence()


class NativeRcclTransport:
    """The transport inside the library (csrc/ws_rccl.cpp): RCCL send/recv groups, all-to-alls and all-gathers issued by the
    C++ side itself -- no Python in the step.  The host only moves rank 0's 128-byte unique id to every rank."""

    def __init__(self, unique_id, rank, world, device, library=None):
        L = self._L = library if library is not None else fluid.load_library()
        L.ws_rccl_transport_create.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(WsTransport)]
        L.ws_rccl_transport_destroy.argtypes = [C.POINTER(WsTransport)]
        L.ws_rccl_last_error.restype = C.c_char_p
        assert len(unique_id) == 128
        self.error = None
        self.struct = WsTransport()
        st = L.ws_rccl_transport_create(unique_id, rank, world, device, C.byref(self.struct))
        if st != 0:
            raise fluid.WsError(st, "ws_rccl_transport_create: " + (L.ws_rccl_last_error() or b"").decode())

    @staticmethod
    def unique_id(library=None):
        L = library if library is not None else fluid.load_library()
        L.ws_rccl_unique_id.argtypes = [C.c_char_p]
        L.ws_rccl_last_error.restype = C.c_char_p
        buf = C.create_string_buffer(128)
        st = L.ws_rccl_unique_id(buf)
        if st != 0:
            raise fluid.WsError(st, "ws_rccl_unique_id: " + (L.ws_rccl_last_error() or b"").decode())
        return buf.raw

    def communicators(self):
        """2 = each of the slab step's two streams has a communicator of its own."""
        self._L.ws_rccl_transport_communicators.argtypes = [C.POINTER(WsTransport)]
        self._L.ws_rccl_transport_communicators.restype = C.c_uint32
        return int(self._L.ws_rccl_transport_communicators(C.byref(self.struct)))

    def close(self):
        if self.struct.ctx:
            self._L.ws_rccl_transport_destroy(C.byref(self.struct))


def make_dist_workload(ws, block, size, dist_name, rank, world, seed=0, chunk=1 << 22):
    """This rank's share of an N-GPU benchmark workload: lattice block `block` (ni, nj, nk) or a uniform cloud of
    ni * nj * nk particles (counter-based generator, `seed`) in a container of `size`, cut into `world` x-slabs.
    Returns (positions_local, ids_local, n_global, params).  Generated in chunks so that no rank ever holds all
    positions at once; the cloud is bit-identical to workloads.uniform_cloud of the whole domain."""
    from . import workloads

    params = fluid.make_params(container_size=size)
    n_global = block[0] * block[1] * block[2]
    pos_parts, id_parts = [], []
    if dist_name == "lattice":
        # cube_fluid order: i (x) outermost -> generate x-planes in chunks of whole planes
        plane = block[1] * block[2]
        planes_per_chunk = max(1, chunk // plane)
        full_off = np.float32(0.1) - np.float32(block[0]) * np.float32(0.1)
        for i0 in range(0, block[0], planes_per_chunk):
            ni = min(planes_per_chunk, block[0] - i0)
            sub = fluid.cube_fluid(ni, block[1], block[2])  # x = i * diam + (r - ni r): re-base x exactly
            xi = (np.arange(i0, i0 + ni, dtype=np.float32) * (np.float32(0.1) * np.float32(2.0)) + full_off)
            sub = sub.reshape(ni, plane, 3)
            sub[:, :, 0] = xi[:, None]
            sub = sub.reshape(-1, 3)
            ids = np.arange(i0 * plane, (i0 + ni) * plane, dtype=np.uint32)
            own = assign(params, sub, world) == rank
            pos_parts.append(sub[own]); id_parts.append(ids[own])
    else:
        for s0 in range(0, n_global, chunk):
            m = min(chunk, n_global - s0)
            sub = workloads.uniform_cloud(m, seed, list(params.ext_min), list(params.ext_max), start=s0)
            ids = np.arange(s0, s0 + m, dtype=np.uint32)
            own = assign(params, sub, world) == rank
            pos_parts.append(sub[own]); id_parts.append(ids[own])
    return np.concatenate(pos_parts), np.concatenate(id_parts), n_global, params


# ------------------------------------------------------------------------------------------
class LoopbackHub:
    """Rendezvous point for `world` slabs living in one process (one host thread each)."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=float(os.environ.get("WS_LOOPBACK_TIMEOUT", "300")))  # a rank left
        # alone in a collective (a protocol bug) breaks the barrier instead of hanging the process
        self.slots = [None] * world
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]

    def copy(self, dst, src, nbytes, stream):
        """Device-to-device copy ordered on the READER's stream (a blocking hipMemcpy on the null stream is ordered
        with neither of the two non-blocking streams involved)."""
        if nbytes and self.hip.hipMemcpyAsync(dst, src, nbytes, 3, stream) != 0:  # hipMemcpyDeviceToDevice
            raise RuntimeError("hipMemcpyAsync failed")

    def done(self, stream):
        """The reader's copies have landed: only then may the writers reuse their send ranges."""
        if self.hip.hipStreamSynchronize(stream) != 0:
            raise RuntimeError("hipStreamSynchronize failed")

    def transport(self, rank):
        return _LoopbackTransport(self, rank)


class _LoopbackTransport(_TransportBase):
    def __init__(self, hub, rank):
        super().__init__()
        self.hub, self.rank, self.world = hub, rank, hub.world

    def sendrecv(self, sp, sb, rp, rb, stream):
        hub = self.hub
        hub.hip.hipStreamSynchronize(stream)  # my boundary data is final before anybody reads it
        hub.slots[self.rank] = (sp, sb)
        hub.barrier.wait()
        for i in range(len(rb)):
            if not rb[i]:
                continue
            d = i % 2  # my left neighbour's right-going segment is entry i + 1 of its lists, and vice versa
            psp, psb = hub.slots[self.rank - 1 if d == 0 else self.rank + 1]
            j = i + 1 if d == 0 else i - 1
            assert psb[j] == rb[i], "neighbour sends %d bytes, I expect %d" % (psb[j], rb[i])
            hub.copy(rp[i], psp[j], rb[i], stream)
        hub.done(stream)
        hub.barrier.wait()  # nobody reuses a send range before every reader is done

    def allgather_dev(self, sp, rp, nbytes, stream):
        hub = self.hub
        hub.hip.hipStreamSynchronize(stream)
        hub.slots[self.rank] = sp
        hub.barrier.wait()
        for r in range(self.world):
            hub.copy(rp + r * nbytes, hub.slots[r], nbytes, stream)
        hub.done(stream)
        hub.barrier.wait()

    def alltoall_dev(self, sp, rp, nbytes, stream):
        hub = self.hub
        hub.hip.hipStreamSynchronize(stream)
        hub.slots[self.rank] = sp
        hub.barrier.wait()
        for r in range(self.world):  # what rank r addressed to me
            hub.copy(rp + r * nbytes, hub.slots[r] + self.rank * nbytes, nbytes, stream)
        hub.done(stream)
        hub.barrier.wait()


# ------------------------------------------------------------------------------------------
class SlabWorker:
    """One x-slab of the domain on one GPU (ws_slab_create / ws_step / ws_slab_read_particles)."""

    def __init__(self, positions, ids, n_global, params, rank, world, transport, device=0, stream=None, profile=False,
                 capacity=0, ghost_capacity=0, ieee_division=False, graph=False, exact_messages=False, lagged_messages=False,
                 fixed_messages=False, overlap=True, graph_multirank=False, library=None):
        L = self._L = library if library is not None else fluid.load_library()
        L.ws_slab_create.argtypes = [C.POINTER(fluid.WsParams), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                     C.POINTER(fluid.WsDeviceCfg), C.POINTER(WsTransport), C.POINTER(C.c_void_p)]
        L.ws_slab_read_particles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.ws_slab_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        self.transport = transport  # keep the callback thunks alive
        self.rank, self.world, self.n_global = rank, world, n_global
        positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        ids = np.ascontiguousarray(ids, np.uint32)
        assert positions.shape[0] == ids.shape[0]
        cfg = fluid.WsDeviceCfg()
        cfg.device, cfg.rank, cfg.world_size = device, rank, world
        cfg.flags = ((fluid.WS_FLAG_PROFILE if profile else 0) | (fluid.WS_FLAG_IEEE_DIVISION if ieee_division else 0)
                     | (fluid.WS_FLAG_GRAPH if graph else 0) | (fluid.WS_FLAG_EXACT_MESSAGES if exact_messages else 0)
                     | (fluid.WS_FLAG_LAGGED_MESSAGES if lagged_messages else 0)
                     | (fluid.WS_FLAG_FIXED_MESSAGES if fixed_messages else 0) | (0 if overlap else fluid.WS_FLAG_NO_OVERLAP)
                     | (fluid.WS_FLAG_GRAPH_MULTIRANK if graph_multirank else 0))
        cfg.capacity, cfg.ghost_capacity = capacity, ghost_capacity
        cfg.stream = stream
        self.params = params
        self._h = C.c_void_p()
        st = L.ws_slab_create(C.byref(params), positions.ctypes.data, ids.ctypes.data, positions.shape[0], n_global,
                              C.byref(cfg), C.byref(transport.struct), C.byref(self._h))
        if st != 0:
            raise fluid.WsError(st, (L.ws_last_error(None) or b"").decode())

    def _check(self, st):
        if st != 0:
            msg = (self._L.ws_last_error(self._h) or b"").decode()
            if self.transport.error is not None:
                msg += " [transport: %r]" % (self.transport.error,)
            raise fluid.WsError(st, msg)

    def run(self, steps=1):
        for _ in range(steps):
            self._check(self._L.ws_step(self._h))

    def sync(self):
        self._check(self._L.ws_sync(self._h))

    def num_owned(self):
        """Particles this slab owns after the steps enqueued so far (the count lives on the device: waits for them)."""
        self.sync()
        return int(self._L.ws_num_particles(self._h))

    def set_params(self, params):
        """ws_set_params on a slab handle.  A new smoothing radius / container is COLLECTIVE (every rank makes the same
        call): the cell grid, the cuts and every slab's particle set are rebuilt."""
        self._check(self._L.ws_set_params(self._h, C.byref(params)))
        self.params = params

    # ---- the host's per-frame calls on a slab handle: global, id-ordered, COLLECTIVE (every rank calls them at the
    #      same point; every rank gets the whole array) -- update() / despawn_liquid of src/fluid_compute.rs:468-525
    def read_positions(self, want=True):
        out = np.empty((self.n_global, 3), np.float32) if want else None
        self._check(self._L.ws_read_positions(self._h, out.ctypes.data if want else None))
        return out

    def read_positions_begin(self, buf):
        assert buf.dtype == np.float32 and buf.shape == (self.n_global, 3) and buf.flags.c_contiguous
        self._check(self._L.ws_read_positions_begin(self._h, buf.ctypes.data))

    def read_positions_end(self):
        self._check(self._L.ws_read_positions_end(self._h))

    def read_positions_begin_owned(self):
        """COLLECTIVE like read_positions_begin, into the library's own page-locked double buffer."""
        self._L.ws_read_positions_begin.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self._L.ws_read_positions_begin(self._h, None))

    def read_positions_view(self):
        p = C.c_void_p()
        self._L.ws_read_positions_view.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        self._check(self._L.ws_read_positions_view(self._h, C.byref(p)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(self.n_global, 3))

    def read_speeds(self):
        out = np.empty(self.n_global, np.float32)
        self._check(self._L.ws_read_speeds(self._h, out.ctypes.data))
        return out

    def read_vec(self, name="particles"):
        if name != "particles":
            raise KeyError(name)
        out = np.empty(self.n_global, fluid.PARTICLE_DTYPE)
        self._check(self._L.ws_read_particles(self._h, out.ctypes.data))
        return out

    def reset(self, positions):
        """ws_reset: `positions` = ALL n_global initial positions (the same array on every rank)."""
        positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        assert positions.shape[0] == self.n_global
        self._check(self._L.ws_reset(self._h, positions.ctypes.data))

    def write_particles(self, records):
        """ws_write_particles: `records` = ALL n_global 80-byte records in id order (the same array on every rank)."""
        records = np.ascontiguousarray(records, fluid.PARTICLE_DTYPE)
        assert records.shape[0] == self.n_global
        self._check(self._L.ws_write_particles(self._h, records.ctypes.data))

    def sort_view(self):
        keys = np.empty(self.n_global, np.uint32)
        perm = np.empty(self.n_global, np.uint32)
        off = np.empty(self.n_global, np.uint32)
        self._check(self._L.ws_read_sort_view(self._h, keys.ctypes.data, perm.ctypes.data, off.ctypes.data))
        return keys, perm, off

    def stats(self):
        out = np.zeros(16, np.uint32)
        self._check(self._L.ws_read_stats(self._h, out.ctypes.data))
        st = {"mask_overflow": int(out[0]), "cells_merged": tuple(int(x) for x in out[1:4]), "graph_steps": int(out[4])}
        if out[6]:  # a slab handle: peaks of the fixed-capacity messages against their capacities
            st.update(halo_peak=int(out[5]), halo_capacity=int(out[6]), migration_peak=int(out[7]), migration_capacity=int(out[8]), far_peak=int(out[9]), far_capacity=int(out[10]),
                      migration_now=int(out[11]), halo_now=int(out[12]), far_now=int(out[13]), size_waits=int(out[14]))
        return st

    def rebalance(self):
        """ws_slab_rebalance (collective): re-cut the slabs to equal particle counts."""
        self._L.ws_slab_rebalance.argtypes = [C.c_void_p]
        self._check(self._L.ws_slab_rebalance(self._h))

    def counters(self):
        """Migration counters since creation (ws_slab_counters): owned now, left, arrived, left by the far route."""
        out = (C.c_uint64 * 4)()
        self._check(self._L.ws_slab_counters(self._h, out))
        return {"owned": int(out[0]), "left": int(out[1]), "arrived": int(out[2]), "far": int(out[3])}

    def read(self):
        """(records[n_owned], ids[n_owned]) of the particles this slab owns now."""
        cap = self.num_owned()
        out = np.empty(cap, fluid.PARTICLE_DTYPE)
        ids = np.empty(cap, np.uint32)
        n = C.c_uint32(0)
        self._check(self._L.ws_slab_read_particles(self._h, out.ctypes.data, ids.ctypes.data, cap, C.byref(n)))
        return out[: n.value], ids[: n.value]

    def profile(self):
        res = {}
        for name, k in fluid.KERNEL_IDS.items():
            ms, cnt = C.c_double(0), C.c_uint64(0)
            self._check(self._L.ws_profile_read(self._h, k, C.byref(ms), C.byref(cnt)))
            res[name] = (ms.value, int(cnt.value))
        return res

    def profile_reset(self):
        self._check(self._L.ws_profile_reset(self._h))

    def profile_select(self, mask):
        self._check(self._L.ws_profile_select(self._h, mask & 0xFFFFFFFF))

    def close(self):
        if self._h:
            self._L.ws_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_loopback_program(positions, params, world, program, device=0, **kw):
    """`world` slabs of one domain inside this process (one thread per slab); every thread runs `program(worker, rank)`
    -- the host's frame loop, written once against the worker interface -- and the results of all ranks are returned
    ([program's return value per rank]).  The loopback transport's barrier breaks (instead of hanging) when a rank
    leaves a collective alone."""
    positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    n = positions.shape[0]
    owner = assign(params, positions, world, kw.get("library"))
    hub = LoopbackHub(world)
    results, errors = [None] * world, []

    def body(r):
        try:
            sel = np.flatnonzero(owner == r).astype(np.uint32)
            w = SlabWorker(positions[sel], sel, n, params, r, world, hub.transport(r), device=device, **kw)
            results[r] = program(w, r)
            w.close()
        except Exception as e:  # pragma: no cover - surfaced by the caller
            errors.append((r, e))
            hub.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise RuntimeError("slab thread failed: %r" % (errors,))
    return results


def run_loopback(positions, params, world, steps, device=0, ieee_division=False, capacity=0, ghost_capacity=0,
                 collect_errors=False, counters=None, change_params=None, sync_every_step=False, state=None,
                 exact_messages=False, lagged_messages=False, **worker_kw):
    """Step `world` slabs of one domain inside this process (one thread per slab) and return the
    particles of all slabs merged into original-id order.  Test helper for one-GPU boxes.
    collect_errors: instead of raising, return {rank: (steps completed, exception)} for the slabs whose ws_step
    failed (the capacity tests expect every rank to fail alike).
    counters: a dict that receives {rank: SlabWorker.counters()} taken after the last step.
    change_params: (step, params) -- every slab calls ws_set_params(params) after `step` steps.
    sync_every_step (with collect_errors): ws_sync after every ws_step, as a frame loop that reads every frame does;
    what ws_sync reports is recorded under the key (rank, "sync") and the rank goes on stepping.
    state: 80-byte records (original-id order) to start from instead of positions at rest."""
    positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    n = positions.shape[0]
    owner = assign(params, positions, world, worker_kw.get("library"))
    hub = LoopbackHub(world)
    out = np.zeros(n, fluid.PARTICLE_DTYPE)
    seen = np.zeros(n, np.int32)
    errors = []
    owned_counts = [None] * world

    def body(r):
        try:
            sel = np.flatnonzero(owner == r).astype(np.uint32)
            w = SlabWorker(positions[sel], sel, n, params, r, world, hub.transport(r), device=device,
                           ieee_division=ieee_division, capacity=capacity, ghost_capacity=ghost_capacity,
                           exact_messages=exact_messages, lagged_messages=lagged_messages, **worker_kw)
            if state is not None:
                w.write_particles(state)
            if collect_errors:
                for k in range(steps):
                    try:
                        w.run(1)
                    except fluid.WsError as e:
                        errors.append((r, (k, e)))
                        w.close()
                        return
                    if sync_every_step:
                        try:
                            w.sync()
                        except fluid.WsError as e:
                            errors.append(((r, "sync"), (k, e)))
            elif change_params is not None:
                w.run(change_params[0])
                w.set_params(change_params[1])
                w.run(steps - change_params[0])
            else:
                w.run(steps)
            rec, ids = w.read()
            out[ids] = rec
            np.add.at(seen, ids, 1)
            owned_counts[r] = len(ids)
            if counters is not None:
                counters[r] = dict(w.counters(), **{k: v for k, v in w.stats().items() if k.endswith("_peak")})
            w.close()
        except Exception as e:  # pragma: no cover - surfaced by the caller
            errors.append((r, e))
            hub.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if collect_errors:
        return dict(errors)
    if errors:
        raise RuntimeError("slab thread failed: %r" % (errors,))
    assert np.all(seen == 1), "every particle must be owned by exactly one slab"
    return out, owned_counts
