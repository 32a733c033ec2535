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


def exchange(rank, world, to_left, to_right):
    """Send arrays to the x-neighbours, receive theirs (variable length: sizes first)."""
    got = {}
    for peer, payload, key in ((rank - 1, to_left, "left"), (rank + 1, to_right, "right")):
        if not 0 <= peer < world:
            got[key] = None
            continue
        size = torch.tensor([payload.shape[0]], dtype=torch.int64)
        other = torch.zeros(1, dtype=torch.int64)
        reqs = [dist.isend(size, peer), dist.irecv(other, peer)]
        for r in reqs:
            r.wait()
        buf = torch.zeros((int(other), payload.shape[1]), dtype=torch.float64)
        reqs = [dist.isend(torch.from_numpy(payload.astype(np.float64)), peer), dist.irecv(buf, peer)]
        for r in reqs:
            r.wait()
        got[key] = buf.numpy()
    return got


def main():
    out_path, steps = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
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
    for step in range(steps):
        gx = np.clip(xcell(params, state["predicted_position"], org), 0, nx - 1)
        assert np.all((gx >= lo) & (gx < hi)), "ownership invariant"
        # halo A: predicted positions (+ velocity, ids) of my first / last owned layer
        def pack(mask):
            return np.c_[ids[mask], state["predicted_position"][mask, :3], state["velocity"][mask, :3]].astype(np.float64)
        ghosts = exchange(rank, world, pack(gx == lo), pack(gx == hi - 1))
        g = [a for a in (ghosts["left"], ghosts["right"]) if a is not None and len(a)]
        g = np.concatenate(g) if g else np.zeros((0, 7))
        local = np.zeros(len(state) + len(g), O.PARTICLE_DTYPE)
        local[: len(state)] = state
        local["predicted_position"][len(state):, :3] = g[:, 1:4].astype(np.float32)
        local["velocity"][len(state):, :3] = g[:, 4:7].astype(np.float32)
        local["position"][len(state):] = local["predicted_position"][len(state):]
        gid = g[:, 0].astype(np.int64)
        orc = oracle_from_params(O, local["position"][:, :3].copy(), params)
        orc.set_particles(local)
        orc.hash_particles(); orc.sort(O.SORT_FAST); orc.calculate_cell_offsets(); orc.update_density()
        # halo B: the ghosts' densities come from their owners (a ghost's own neighbours are not all here)
        def packd(mask):
            return np.c_[ids[mask], orc.particles["density"][: len(state)][mask], orc.particles["pressure"][: len(state)][mask]].astype(np.float64)
        dens = exchange(rank, world, packd(gx == lo), packd(gx == hi - 1))
        d = [a for a in (dens["left"], dens["right"]) if a is not None and len(a)]
        if d:
            d = np.concatenate(d)
            assert np.array_equal(d[:, 0].astype(np.int64), gid)  # same order as halo A
            wrong = np.abs(orc.particles["density"][len(state):, 0] - d[:, 1]).max() if len(d) else 0.0
            orc.particles["density"][len(state):] = d[:, 1:3].astype(np.float32)
            orc.particles["pressure"][len(state):] = d[:, 3:5].astype(np.float32)
            if step == 0 and len(d):
                # without halo B the ghosts' locally computed densities are wrong (missing neighbours)
                assert wrong > 1.0
        orc.update_pressure_force()
        orc.integrate()
        state = orc.particles[: len(state)].copy()
        # migration: hand particles whose predicted x cell left [lo, hi) to the neighbour
        gx = np.clip(xcell(params, state["predicted_position"], org), 0, nx - 1)
        assert np.all((gx >= lo - 1) & (gx <= hi)), "this small test only migrates to direct neighbours"
        def packm(mask):
            return np.c_[ids[mask], state["position"][mask, :3], state["velocity"][mask, :3],
                         state["predicted_position"][mask, :3], state["density"][mask], state["acceleration"][mask, :3]].astype(np.float64)
        arr = exchange(rank, world, packm(gx < lo), packm(gx >= hi))
        keep = (gx >= lo) & (gx < hi)
        state, ids = state[keep], ids[keep]
        for a in (arr["left"], arr["right"]):
            if a is not None and len(a):
                add = np.zeros(len(a), O.PARTICLE_DTYPE)
                add["position"][:, :3] = a[:, 1:4]; add["velocity"][:, :3] = a[:, 4:7]
                add["predicted_position"][:, :3] = a[:, 7:10]; add["density"] = a[:, 10:12]
                add["acceleration"][:, :3] = a[:, 12:15]
                state = np.concatenate([state, add]); ids = np.concatenate([ids, a[:, 0].astype(np.int64)])
    np.savez(out_path % rank, ids=ids, state=state)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
