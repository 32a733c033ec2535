"""Rank body of tests/test_slab_gloo_cpu.py: a CPU model of the slab protocol (x-slabs on cell cuts,
halo A = boundary-layer predicted positions, halo B = their densities, migration by the predicted
position's x cell) with the ORACLE as the compute engine and gloo as the transport.  It checks, on
CPU and with world_size 2, the host logic the GPU path relies on: ws_slab_assign's cuts, that a
one-cell halo is sufficient, that K5 needs the owners' densities for the ghosts (halo B), and that
migration conserves particles."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import water_sandbox_amd as ws  # noqa: E402
from oracle import oracle as O  # noqa: E402
from util import oracle_from_params  # noqa: E402

H = np.float32(0.25)


def xcell(params, pred, nx_org):
    return np.floor(pred[:, 0] / H).astype(np.int64) - nx_org


HDR = 4  # message header words: records, sender's sticky error bits, sender's owned count, step (csrc WS_HDR_WORDS)
ERR_MIGRATION, ERR_HALO = 1, 4


class Protocol:
    """The slab step's message discipline (csrc/ws_slab.inc) on gloo: every message has a FIXED capacity known to
    both ends and carries its record count in a header; nothing is sized by a count only the sender knows.  An
    overrun clamps and sets a sticky error bit, which reaches every rank with the per-step all-to-all and is acted on
    LAG steps later by all ranks alike."""
    LAG = 2

    def __init__(self, rank, world, halo_cap, mig_cap, far_cap):
        self.rank, self.world = rank, world
        self.halo_cap, self.mig_cap, self.far_cap = halo_cap, mig_cap, far_cap
        self.err = 0
        self.tables = {}  # step -> all ranks' headers

    def _message(self, payload, cap, width, n_owned, step, errbit):
        msg = np.zeros((1 + cap, max(width, HDR)), np.float64)
        cnt = payload.shape[0]
        if cnt > cap:
            self.err |= errbit
            cnt = cap
        msg[0, :HDR] = (cnt, self.err, n_owned, step)
        msg[1:1 + cnt, :width] = payload[:cnt]
        return torch.from_numpy(msg)

    def exchange(self, to_left, to_right, cap, n_owned, step, errbit):
        """One fixed-size message to / from each x-neighbour."""
        got = {"left": None, "right": None}
        width = to_left.shape[1]
        reqs, bufs = [], {}
        for peer, payload, key in ((self.rank - 1, to_left, "left"), (self.rank + 1, to_right, "right")):
            if not 0 <= peer < self.world:
                continue
            bufs[key] = torch.zeros((1 + cap, max(width, HDR)), dtype=torch.float64)
            reqs += [dist.isend(self._message(payload, cap, width, n_owned, step, errbit), peer), dist.irecv(bufs[key], peer)]
        for r in reqs:
            r.wait()
        for key, buf in bufs.items():
            a = buf.numpy()
            got[key] = a[1:1 + int(a[0, 0]), :width]
        return got

    def alltoall_far(self, payload, dest, n_owned, step):
        """The messages for particles that cross more than one slab -- one per destination rank, exchanged all-to-all -- and,
        in their headers, every rank's owned count and error bits of this step (csrc/ws_slab.inc: alltoall_dev)."""
        width = payload.shape[1]
        mine = torch.stack([self._message(payload[dest == r], self.far_cap, width, n_owned, step, ERR_MIGRATION) for r in range(self.world)])
        got = torch.zeros_like(mine)
        dist.all_to_all_single(got, mine)
        got = got.numpy()
        self.tables[step] = np.stack([got[q, 0, :HDR] for q in range(self.world)])
        return [got[q, 1:1 + int(got[q, 0, 0]), :width] for q in range(self.world)]

    def gather_by_id(self, ids, payload, n_global):
        """The host's global reads on slab handles (csrc/ws_slab.inc slab_gather): ONE all-gather of the owned counts, ONE
        of {id, payload} records sized by the largest count, then a scatter by id -- every rank ends with the whole,
        id-ordered array.  Returns (array[n_global, width], counts)."""
        mine = torch.tensor([len(ids)], dtype=torch.int64)
        counts = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        dist.all_gather(counts, mine)
        counts = [int(c.item()) for c in counts]
        assert sum(counts) == n_global, "particle count not conserved across slabs: %r" % (counts,)
        cap, width = max(counts), payload.shape[1]
        rec = torch.zeros((cap, 1 + width), dtype=torch.float64)
        rec[: len(ids), 0] = torch.from_numpy(ids.astype(np.float64))
        rec[: len(ids), 1:] = torch.from_numpy(payload.astype(np.float64))
        allr = [torch.zeros_like(rec) for _ in range(self.world)]
        dist.all_gather(allr, rec)
        out = np.full((n_global, width), np.nan)
        for r, a in enumerate(allr):
            a = a.numpy()[: counts[r]]
            out[a[:, 0].astype(np.int64)] = a[:, 1:]
        assert not np.isnan(out).any()
        return out, counts

    def check(self, step):
        """What ws_step does first: the table of step - LAG; any rank's error bit fails every rank at this step."""
        t = self.tables.get(step - self.LAG)
        if t is not None and int(t[:, 1].max()) != 0:
            raise OverflowError("step %d: rank(s) %s overran a message capacity at step %d"
                                % (step, np.flatnonzero(t[:, 1]).tolist(), step - self.LAG))


def main():
    out_path, steps = sys.argv[1], int(sys.argv[2])
    halo_cap = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    recut_at = int(sys.argv[4]) if len(sys.argv) > 4 else -1  # step before which the slabs are re-cut (ws_slab_rebalance)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    proto = Protocol(rank, world, halo_cap=halo_cap, mig_cap=512, far_cap=64)
    params = ws.make_params(container_size=(8.0, 5.0, 5.0), gravity=(5.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(4096, 11, list(params.ext_min), list(params.ext_max))
    n = pos.shape[0]
    owner = ws.slab.assign(params, pos, world)
    # the cuts, recovered from the assignment: global x layer = floor(x/h) - org, org = floor(ext_min/h) - 2
    org = int(np.floor(np.float32(params.ext_min[0]) / H)) - 2
    nx = int(np.floor(np.float32(params.ext_max[0]) / H)) + 2 - org + 1
    cuts = [r * nx // world for r in range(world + 1)]
    gx0 = np.clip(xcell(params, pos, org), 0, nx - 1)
    assert np.array_equal(owner, np.searchsorted(cuts, gx0, side="right") - 1)

    # state of the particles this rank owns: id, position, velocity, predicted (float32 records)
    mine = np.flatnonzero(owner == rank)
    ids = mine.astype(np.int64)
    state = np.zeros(len(mine), O.PARTICLE_DTYPE)
    state["position"][:, :3] = pos[mine]
    state["predicted_position"][:, :3] = pos[mine]
    lo, hi = cuts[rank], cuts[rank + 1]
    max_err = 0.0
    failed_at = -1
    frames = []
    for step in range(steps):
        try:
            proto.check(step)
        except OverflowError:
            failed_at = step  # every rank raises here, at the same step, outside any collective
            break
        # update(): every rank reads the id-ordered positions of ALL particles (src/fluid_compute.rs:478-485)
        allpos, counts = proto.gather_by_id(ids, state["position"][:, :3], n)
        frames.append(allpos.astype(np.float32))
        if step == recut_at:
            # ws_slab_rebalance: gather the full state, the x-layer histogram of all predicted positions (identical on
            # every rank), cuts of equal particle counts from the library's host function, every rank keeps its share
            full, _ = proto.gather_by_id(ids, np.c_[state["position"][:, :3], state["velocity"][:, :3],
                                                    state["predicted_position"][:, :3]], n)
            gxa = np.clip(np.floor(full[:, 6].astype(np.float32) / H).astype(np.int64) - org, 0, nx - 1)
            hist = np.bincount(gxa, minlength=nx).astype(np.uint32)
            newcuts = np.zeros(world + 1, np.uint32)
            lib = ws.load_library()
            import ctypes as C
            lib.ws_slab_balanced_cuts.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
            assert lib.ws_slab_balanced_cuts(hist.ctypes.data, nx, world, newcuts.ctypes.data) == 0
            cuts = [int(c) for c in newcuts]
            lo, hi = cuts[rank], cuts[rank + 1]
            sel = np.flatnonzero((gxa >= lo) & (gxa < hi))
            ids = sel.astype(np.int64)
            state = np.zeros(len(sel), O.PARTICLE_DTYPE)
            state["position"][:, :3] = full[sel, 0:3]; state["velocity"][:, :3] = full[sel, 3:6]
            state["predicted_position"][:, :3] = full[sel, 6:9]
            owned_after_recut = len(sel)
        gx = np.clip(xcell(params, state["predicted_position"], org), 0, nx - 1)
        assert np.all((gx >= lo) & (gx < hi)) or proto.err, "ownership invariant"
        # halo A: predicted positions (+ velocity, ids) of my first / last owned layer
        def pack(mask):
            return np.c_[ids[mask], state["predicted_position"][mask, :3], state["velocity"][mask, :3]].astype(np.float64)
        ghosts = proto.exchange(pack(gx == lo), pack(gx == hi - 1), proto.halo_cap, len(state), step, ERR_HALO)
        g = [a for a in (ghosts["left"], ghosts["right"]) if a is not None and len(a)]
        g = np.concatenate(g) if g else np.zeros((0, 7))
        # (padded to a power of two with far-away dummies: the oracle hashes into as many buckets as it has particles, and
        # for an arbitrary count two cells of one 27-stencil can share a bucket -- the reference's table then counts a
        # neighbour twice; the slab sizes of this model are arbitrary, the GPU path uses the global N)
        real = len(state) + len(g)
        padded = 1 << max(12, int(real - 1).bit_length())
        local = np.zeros(padded, O.PARTICLE_DTYPE)
        local["predicted_position"][real:, 0] = 1.0e4 + 10.0 * np.arange(padded - real, dtype=np.float32)
        local["predicted_position"][real:, 1:3] = 1.0e4
        local["position"][real:] = local["predicted_position"][real:]
        local[: len(state)] = state
        local["predicted_position"][len(state):real, :3] = g[:, 1:4].astype(np.float32)
        local["velocity"][len(state):real, :3] = g[:, 4:7].astype(np.float32)
        local["position"][len(state):real] = local["predicted_position"][len(state):real]
        gid = g[:, 0].astype(np.int64)
        orc = oracle_from_params(O, local["position"][:, :3].copy(), params)
        orc.set_particles(local)
        orc.hash_particles(); orc.sort(O.SORT_FAST); orc.calculate_cell_offsets(); orc.update_density()
        # halo B: the ghosts' densities come from their owners (a ghost's own neighbours are not all here)
        def packd(mask):
            return np.c_[ids[mask], orc.particles["density"][: len(state)][mask], orc.particles["pressure"][: len(state)][mask]].astype(np.float64)
        dens = proto.exchange(packd(gx == lo), packd(gx == hi - 1), proto.halo_cap, len(state), step, ERR_HALO)
        d = [a for a in (dens["left"], dens["right"]) if a is not None and len(a)]
        if d:
            d = np.concatenate(d)
            assert np.array_equal(d[:, 0].astype(np.int64), gid) or proto.err  # same order as halo A
            wrong = np.abs(orc.particles["density"][len(state):real, 0] - d[:, 1]).max() if len(d) else 0.0
            orc.particles["density"][len(state):real] = d[:, 1:3].astype(np.float32)
            orc.particles["pressure"][len(state):real] = d[:, 3:5].astype(np.float32)
            if step == 0 and len(d):
                # without halo B the ghosts' locally computed densities are wrong (missing neighbours)
                assert wrong > 1.0
        orc.update_pressure_force()
        orc.integrate()
        state = orc.particles[: len(state)].copy()
        # migration: hand particles whose predicted x cell left [lo, hi) to the neighbour
        gx = np.clip(xcell(params, state["predicted_position"], org), 0, nx - 1)
        dest = np.searchsorted(cuts, gx, side="right") - 1  # the slab each particle belongs to now
        def packm(mask):
            return np.c_[ids[mask], state["position"][mask, :3], state["velocity"][mask, :3],
                         state["predicted_position"][mask, :3], state["density"][mask], state["acceleration"][mask, :3]].astype(np.float64)
        # direct neighbours by send/recv; anything further in the message addressed to its owner (all-to-all)
        arr = proto.exchange(packm(dest == rank - 1), packm(dest == rank + 1), proto.mig_cap, len(state), step, ERR_MIGRATION)
        far = np.abs(dest - rank) > 1
        far_all = proto.alltoall_far(packm(far), dest[far], len(state), step)
        keep = dest == rank
        state, ids = state[keep], ids[keep]
        arrivals = [arr["left"], arr["right"]] + [a for q, a in enumerate(far_all) if q != rank]
        for a in arrivals:
            if a is not None and len(a):
                add = np.zeros(len(a), O.PARTICLE_DTYPE)
                add["position"][:, :3] = a[:, 1:4]; add["velocity"][:, :3] = a[:, 4:7]
                add["predicted_position"][:, :3] = a[:, 7:10]; add["density"] = a[:, 10:12]
                add["acceleration"][:, :3] = a[:, 12:15]
                state = np.concatenate([state, add]); ids = np.concatenate([ids, a[:, 0].astype(np.int64)])
    np.savez(out_path % rank, ids=ids, state=state, failed_at=failed_at, frames=np.array(frames),
             cuts=np.array(cuts), owned_after_recut=locals().get("owned_after_recut", -1))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
