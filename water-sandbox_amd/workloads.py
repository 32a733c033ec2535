"""Synthetic inputs of SURVEY.md 8(d) / BASELINE.md: lattice (A) and uniform cloud (B) for the
five benchmark configurations.  Host-side numpy only; no physics here."""
import numpy as np

from . import fluid

# name -> (lattice block ni,nj,nk ; container size), BASELINE.md section 2.  The container
# is the block extent x (1.25, 1.40625, 1.40625), the reference's own ratio
# (src/fluid_compute.rs:15-17 vs src/fluid_container.rs:8); C1 is the planar sheet.
CONFIGS = {
    "c1": ((64, 64, 1), (16.0, 18.0, 0.2)),
    "c2": ((128, 64, 32), (32.0, 18.0, 9.0)),
    "c3": ((256, 128, 128), (64.0, 36.0, 36.0)),
    "c4": ((512, 256, 128), (128.0, 72.0, 36.0)),
    "c5": ((1024, 256, 256), (256.0, 72.0, 72.0)),
    # the reference's compiled-in default, src/fluid_compute.rs:15-17
    "ref": ((64, 32, 32), (16.0, 9.0, 9.0)),
}
CONFIG_NUMBER = {"c1": 1, "c2": 2, "c3": 3, "c4": 4, "c5": 5, "ref": 0}


def block_for(n_particles):
    """A power-of-two lattice block with ni >= nj >= nk and ni*nj*nk == n_particles (x longest)."""
    k = int(n_particles).bit_length() - 1
    if (1 << k) != n_particles:
        raise ValueError("lattice workloads need a power-of-two particle count")
    ek = k // 3
    ej = (k - ek) // 2
    ei = k - ek - ej
    return (1 << ei, 1 << ej, 1 << ek)


def container_for_block(block, r=0.1):
    ni, nj, nk = block
    d = 2.0 * r
    return (ni * d * 1.25, nj * d * 1.40625, nk * d * 1.40625)


def config_params(name, **overrides):
    block, size = CONFIGS[name]
    return block, fluid.make_params(container_size=size, **overrides)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform_cloud(n, seed, ext_min, ext_max, start=0):
    """Distribution (B): component c of particle i = ext_min[c] + u * (ext_max[c] - ext_min[c]),
    u = (splitmix64(seed ^ (3 i + c)) >> 40) * 2^-24, all in f32.  Counter-based."""
    idx = np.arange(start, start + n, dtype=np.uint64)
    out = np.empty((n, 3), np.float32)
    mn = np.asarray(ext_min, np.float32)
    mx = np.asarray(ext_max, np.float32)
    for c in range(3):
        r = _splitmix64(np.uint64(seed) ^ (np.uint64(3) * idx + np.uint64(c))) >> np.uint64(40)
        u = r.astype(np.float32) * np.float32(2.0 ** -24)
        span = np.float32(mx[c] - mn[c])
        out[:, c] = mn[c] + u * span
    return out


def cloud_seed(name):
    return 0x5EED0000 + CONFIG_NUMBER[name]


def make_workload(name, dist="cloud"):
    """(positions[n,3] f32, params) for a named config and distribution 'lattice' | 'cloud'."""
    block, params = config_params(name)
    if dist == "lattice":
        pos = fluid.cube_fluid(*block)
    elif dist == "cloud":
        n = block[0] * block[1] * block[2]
        pos = uniform_cloud(n, cloud_seed(name), list(params.ext_min), list(params.ext_max))
    else:
        raise ValueError(dist)
    return pos, params
