#!/usr/bin/env python3
"""bench.py -- simulation steps/s of the SPH fluid step on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3] [--dist cloud|lattice]

One "step" = one full pass of the hot path (cell sort -> density -> force -> integrate) over all
particles, inputs resident in HBM before the timed region, no host readback inside it.  Rank 0
prints ONE JSON line.

The workload is not stationary: a uniform cloud collapses onto the floor within
~120 steps and the step gets ~3x more expensive, so the line carries TWO windows measured in the same
run: `value` = steps W..W+K of the trajectory (what the flags ask for), and `settled` = steps
400..500 (the state the simulation lives in afterwards), each with its own roofline object.
The whole measurement is repeated --reps times (default 5) on fresh trajectories and the line reports the
repetition with the MEDIAN time, each window on its own (SURVEY 8(d) protocol); `repetitions` lists them all.
`roofline.frac` prices the dominant kernel against the HBM roofline by ALGORITHMIC bytes (the contract);
what actually limits it is in `roofline.limiter` / `roofline.secondary` (VALU issue rate, texture
addresser, from the committed counter passes of the same window) and `step_traffic` is the counter-measured
HBM traffic of all five kernels of a step over the step time.
`value_ieee` is the first window again with WS_FLAG_IEEE_DIVISION (correctly rounded sqrt / division,
the CPU restatement's arithmetic) instead of the hardware's 1-ULP forms.  `with_readback` is the host
application's frame pattern over the first window (positions read back every step, overlapped; PCIe
inclusive, never `value`).

For N > 1 the driver launches this under torch.distributed.run (one rank per GPU): N = 4 runs
BASELINE.json's config 4 (C4, 16 777 216 particles) and N = 8 config 5 (C5, 67 108 864 particles) cut
into x-slabs; N = 2 runs C3 doubled along x.  See DESIGN.md "Multi-GPU".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic bytes per particle per step, SURVEY.md 8(d): K1 24 + K2 12 + K3 12 + K4 40 + K5 64 + K6 72
B_ALG = {"K1": 24, "K2": 12, "K3": 12, "K4": 40, "K5": 64, "K6": 72}
B_ALG_STEP = sum(B_ALG.values())  # 224
# which reference passes each HIP kernel implements (DESIGN.md "Kernels")
KERNEL_ALG_BYTES = {
    "cell_scan": 0,
    "cell_scatter": B_ALG["K2"],
    "reorder": B_ALG["K3"],
    "density": B_ALG["K4"],
    "force_integrate_bin": B_ALG["K5"] + B_ALG["K6"] + B_ALG["K1"],
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0  # achievable: float4 copy measured on MI355X (MI355X_MICROARCH.md, HBM section)
C3_PARTICLES = 4194304
SETTLED_FROM = 400     # first step of the settled window
SETTLED_STEPS = 100
PROFILES_ROUND = "r05"
PROFILES = os.path.join(ROOT, "profiles", PROFILES_ROUND)
# VALU issue peak of the chip: 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (a packed-f32 or
# transcendental instruction takes more: the fraction below is a lower bound of how busy the VALU issue ports are)
VALU_PEAK_WAVE_INSTR_PER_S = 256 * 4 * 2.4e9 / 2
# the step loop's instantiations as rocprofv3 names them (prefixes): K4 <IEEE = false, CUT = false, SCHED>, K5 <IEEE =
# false, ACCEL_ONLY = false, CUT = false, SCHED> (CUT: the early launches of a slab step; SCHED: the
# cost-guided tile schedule of handles of 2^17 .. 2^20 particles); k_scan<ZERO = true, cells per thread>: 64 from 2^21
# cells on, 32 below
STEP_KERNELS = {"cell_scan": "k_scan<true", "cell_scatter": "k_place", "reorder": "k_reorder<true>",
                "density": "k_density_listed<false, false,", "force_integrate_bin": "k_force_listed<false, false, false,"}


def _by_kernel(table, name):
    """The entry of a per-kernel counter table whose rocprofv3 kernel name is `name` (or starts with it)."""
    if not table:
        return None
    if name in table:
        return table[name]
    hits = [v for k, v in table.items() if k.startswith(name)]
    return hits[0] if len(hits) == 1 else None

KERNEL_LABEL = {
    "density": "density (K4 update_density: radius sweep + accept masks)",
    "force_integrate_bin": "force_integrate_bin (K5 update_pressure_force + K6 integrate + next K1 binning)",
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default=None, help="c1..c5 | ref (BASELINE.md section 2); default: by --gpus")
    ap.add_argument("--dist", default="cloud", choices=["cloud", "lattice"])
    ap.add_argument("--replicate", action="store_true",
                    help="N GPUs: the config replicated N times along x (e.g. --config c3 --replicate: 4 194 304 "
                         "particles per GPU at every N) instead of BASELINE.json's C4 / C5 geometry")
    ap.add_argument("--copies", type=int, default=0,
                    help="the config replicated this many times along x whatever the GPU count (one GPU: the like-for-like "
                         "run of an N-GPU --replicate workload, e.g. --config c3 --copies 2 = what --gpus 2 runs)")
    ap.add_argument("--no-north-star", action="store_true",
                    help="skip the north_star leg (C4, 16.7 M particles, on this one GPU: steps 400..500)")
    ap.add_argument("--reps", type=int, default=5,
                    help="repetitions of the whole measurement (fresh trajectory each: W warm-up steps, K timed steps, the "
                         "run-up to the settled window and its 100 timed steps); the line reports the MEDIAN repetition "
                         "(SURVEY 8(d) / BASELINE.md protocol) and lists all of them")
    ap.add_argument("--graph", action="store_true",
                    help="WS_FLAG_GRAPH: replay the step from a captured hipGraph (BASELINE config 5's 'hipGraph-captured step'); "
                         "per-kernel events are not part of a captured step, so the roofline objects are omitted")
    ap.add_argument("--lagged-messages", action="store_true",
                    help="N > 1: WS_FLAG_LAGGED_MESSAGES -- slab message sizes derived from the demand of a few steps earlier (ws_step "
                         "never waits for the device; a shock front across a slab face fails the run) instead of the default: every "
                         "message at exactly its sender's count, two short host waits per step.  Implied by --graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-settled", action="store_true", help="skip the settled-state window (steps 400..500)")
    ap.add_argument("--no-readback", action="store_true",
                    help="skip the frame-pattern leg (positions read back every step; PCIe inclusive)")
    ap.add_argument("--no-ieee", action="store_true", help="skip the WS_FLAG_IEEE_DIVISION repeat of the first window")
    ap.add_argument("--cpu-steps", type=int, default=12,
                    help="CPU-restatement steps timed for cpu_baseline (~10 s of CPU work at C3 on 16 threads)")
    ap.add_argument("--breakdown", action="store_true", help="also print a per-kernel table to stderr")
    return ap.parse_args()


def cpu_baseline(pos, params, steps):
    """The CPU restatement of the reference WGSL (oracle, 'fast' sort mode: OpenMP over particles, parallel
    radix sort, parallel cell offsets) timed on this host on the SAME workload for a few steps.  Reported,
    never the target."""
    from oracle import oracle as O

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import oracle_from_params

    orc = oracle_from_params(O, pos, params)
    orc.step(O.SORT_FAST)  # warm-up (page faults, thread team)
    t0 = time.perf_counter()
    for _ in range(steps):
        orc.step(O.SORT_FAST)
    dt = time.perf_counter() - t0
    share, why = O.cpu_share()
    return {
        "value": steps / dt,
        "unit": "steps/s",
        "cores": O.default_threads(),
        "kind": "port",
        "sample": "%d full steps of the same %d-particle workload after 1 warm-up step (CPU restatement of the "
        "reference WGSL, fast sort mode; %d OpenMP threads = every core this process may use: %s)"
        % (steps, orc.n, O.default_threads(), why),
    }


def cpu_baseline_other_configs():
    """SURVEY 8(d)'s other CPU rows next to the benchmarked one: C1 with one thread in the exact sort
    mode (the bitonic network stage by stage -- the reference's own algorithm) and with every granted core in the fast
    mode, C2 in the fast mode.  About three seconds of CPU work."""
    import water_sandbox_amd as ws
    from oracle import oracle as O

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import oracle_from_params

    rows = []
    for cfg, mode, threads, steps in (("c1", O.SORT_EXACT, 1, 100), ("c1", O.SORT_FAST, O.default_threads(), 200),
                                      ("c2", O.SORT_FAST, O.default_threads(), 20)):
        O.set_threads(threads)
        p, prm = ws.workloads.make_workload(cfg, "cloud")
        orc = oracle_from_params(O, p, prm)
        orc.step(mode)
        t0 = time.perf_counter()
        for _ in range(steps):
            orc.step(mode)
        dt = time.perf_counter() - t0
        rows.append({"config": cfg, "particles": int(orc.n), "sort_mode": "exact (bitonic network)" if mode == O.SORT_EXACT else "fast (radix)",
                     "cores": threads, "value": steps / dt, "unit": "steps/s", "sample": "%d steps after 1 warm-up, uniform cloud" % steps})
    O.set_threads(O.default_threads())
    return rows


def load_traffic(config, dist, warmup, steps, kernel):
    """HBM bytes per launch of `kernel` from a committed rocprofv3 --pmc pass of EXACTLY this window
    (profiles/rNN/traffic.json, key '<config>-<dist>-w<warmup>-k<steps>'), with its provenance -- or (None, None)
    when no pass of this window is committed.  Never filled from another window."""
    path = os.path.join(PROFILES, "traffic.json")
    key = "%s-%s-w%d-k%d" % (config, dist, warmup, steps)
    try:
        with open(path) as f:
            t = json.load(f)
        v = t.get(key, {}).get("bytes_per_launch", {}).get(kernel)
        if v is None:
            return None, None
        return v, ("profiles/%s/traffic.json[%s]: separate rocprofv3 --pmc passes of this window, size-resolved read requests "
                   "(TCC_EA0_RDREQ_32B / _64B / _128B) + WRITE_SIZE" % (PROFILES_ROUND, key))
    except (OSError, ValueError):
        return None, None


def load_window_counters(config, dist, warmup, steps):
    """Everything the committed counter passes of EXACTLY this window hold (profiles/rNN/traffic.json: FETCH_SIZE /
    WRITE_SIZE per kernel; profiles/rNN/pmc_windows.json: SQ / TA counters per kernel), or {}."""
    key = "%s-%s-w%d-k%d" % (config, dist, warmup, steps)
    out = {}
    for name in ("traffic.json", "pmc_windows.json"):
        try:
            with open(os.path.join(PROFILES, name)) as f:
                out[name] = json.load(f).get(key)
        except (OSError, ValueError):
            out[name] = None
    return out


def step_traffic(counters, ms_per_step):
    """Counter-measured HBM bytes of ALL five kernels of a step over the step time, against the measured float4-copy rate:
    the measured-bytes view of the WHOLE step.  Reads from the size-resolved fabric request counters
    (32 n32 + 64 n64 + 128 n128 = 2 x FETCH_SIZE: every request is a 128-byte line, profiles/r04/traffic_calibration.json),
    writes from WRITE_SIZE.  LINES moved, not useful bytes: a scattered 16-byte gather costs a line."""
    t = counters.get("traffic.json")
    if not t or "read_resolved" not in t:
        return None
    per = {}
    for label, kname in STEP_KERNELS.items():
        f, w = _by_kernel(t["read_resolved"], kname), _by_kernel(t["write_raw"], kname)
        if f is None or w is None:
            return None
        per[label] = f + w
    total = sum(per.values())
    gbps = total / (ms_per_step * 1e-3) / 1e9
    return {"bytes_per_step": total, "bytes_per_kernel": per, "GBps": gbps, "frac_of_measured_copy_bw": gbps / HBM_COPY_GBS,
            "source": "profiles/%s/traffic.json (size-resolved read requests + WRITE_SIZE of the five step kernels) / this "
                      "run's ms_per_step" % PROFILES_ROUND}


def secondary_roofline(counters, label, avg_s):
    """What the counters say bounds the kernel: VALU issue (wave-instructions per second against the chip's issue peak)
    with the scalar instruction stream, the lane utilisation and the texture-addresser busy fraction next to it."""
    p = _by_kernel(counters.get("pmc_windows.json"), STEP_KERNELS[label])
    if not p or avg_s <= 0:
        return None
    valu = p.get("SQ_INSTS_VALU")
    if valu is None:
        return None
    out = {"resource": "valu_issue", "achieved": valu / avg_s, "peak": VALU_PEAK_WAVE_INSTR_PER_S, "unit": "wave-instr/s",
           "frac": valu / avg_s / VALU_PEAK_WAVE_INSTR_PER_S, "valu_wave_instr_per_launch": valu,
           "salu_wave_instr_per_launch": p.get("SQ_INSTS_SALU"),
           "source": "profiles/%s/pmc_windows.json (rocprofv3 --pmc passes of this window) / this run's launch time" % PROFILES_ROUND}
    if p.get("SQ_THREAD_CYCLES_VALU") and p.get("SQ_ACTIVE_INST_VALU"):
        out["lane_utilisation"] = p["SQ_THREAD_CYCLES_VALU"] / 64.0 / p["SQ_ACTIVE_INST_VALU"]
    if p.get("ta_busy_frac") is not None:
        out["texture_addresser_busy_frac"] = p["ta_busy_frac"]
    return out


NEIGHBOUR_KERNELS = ("density", "force_integrate_bin")


def dominant_kernel(*profs):
    """The neighbour kernel with the most time over ALL the timed windows of the run (in the sparse first window K4 and
    K5 are within 2 % of each other and would swap places from run to run; in the settled state K5 leads by a third)."""
    return max(NEIGHBOUR_KERNELS, key=lambda k: sum(p[k][0] / max(p[k][1], 1) * w for p, w in profs))


def roofline_of(prof, owned, config, dist, warmup, steps, distributed, dominant):
    """The roofline object of one timed window from the HIP events of the two neighbour kernels' launches, for the run's
    dominant kernel; the other neighbour kernel's figures ride along under `other`."""
    ms, cnt = prof[dominant]
    if not cnt:
        return None  # --graph: no per-kernel events inside a captured step
    avg_s = ms / max(cnt, 1) * 1e-3
    alg_bytes = KERNEL_ALG_BYTES[dominant] * owned
    achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
    traffic, source = (None, None) if distributed else load_traffic(config, dist, warmup, steps, dominant)
    counters = {} if distributed else load_window_counters(config, dist, warmup, steps)
    secondary = secondary_roofline(counters, dominant, avg_s)
    # the measured-bytes view SURVEY 8(d) asks for next to the algorithmic one: counter bytes per launch over this
    # run's launch time, against the float4-copy rate measured on MI355X (MI355X_MICROARCH.md: 6.29 TB/s)
    measured = traffic / avg_s / 1e9 if traffic and avg_s > 0 else None
    other = [k for k in NEIGHBOUR_KERNELS if k != dominant][0]
    o_ms, o_cnt = prof[other]
    o_avg = o_ms / max(o_cnt, 1) * 1e-3
    o_alg = KERNEL_ALG_BYTES[other] * owned
    return {
        "kernel": KERNEL_LABEL[dominant],
        "dominance": "most time over all timed windows of this run",
        "other": {"kernel": KERNEL_LABEL[other], "avg_launch_ms": o_avg * 1e3, "alg_bytes_per_launch": o_alg,
                  "achieved": o_alg / o_avg / 1e9 if o_avg > 0 else 0.0,
                  "frac": (o_alg / o_avg / 1e9 if o_avg > 0 else 0.0) / HBM_PEAK_GBS,
                  "traffic": (counters.get("traffic.json") or {}).get("bytes_per_launch", {}).get(other),
                  "secondary": secondary_roofline(counters, other, o_avg)},
        # `bound` names the roofline this object prices the kernel against (the contract's HBM roofline, by algorithmic
        # bytes).  It is NOT what limits the kernel: the counters say VALU issue and the texture addresser (`limiter`,
        # `secondary`), and its real HBM traffic is `traffic` (about 1 x the algorithmic bytes).
        "bound": "hbm",
        "priced_against": "hbm (algorithmic bytes / launch time against the 8 TB/s HBM roofline: the contract's pricing, not the limiter)",
        "bound_by_counters": "valu_issue + texture_addresser",
        "limiter": "valu_issue + texture_addresser (divergent 16-B gathers), not HBM: see `secondary`; DESIGN.md 5",
        "secondary": secondary,
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": source,
        "traffic_GBps": measured,
        "traffic_frac_of_measured_copy_bw": measured / HBM_COPY_GBS if measured else None,
        "alg_bytes_per_launch": alg_bytes,
        "avg_launch_ms": avg_s * 1e3,
        "launches_timed": cnt,
    }


def same_config_one_gpu(cfg_name, dist):
    """The committed one-GPU line of the configuration an N-GPU run cuts into slabs (profiles/rNN/bench_<cfg>_one_gpu.json,
    newest round first): {ms_per_step, window, settled_ms_per_step, source}, or None."""
    if dist != "cloud":
        return None
    for rnd in sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.startswith("r")), reverse=True):
        path = os.path.join(ROOT, "profiles", rnd, "bench_%s_one_gpu.json" % cfg_name)
        try:
            with open(path) as f:
                line = json.load(f)
        except (OSError, ValueError):
            continue
        return {"ms_per_step": line.get("ms_per_step"), "warmup": line.get("warmup"), "steps": line.get("steps"),
                "settled_ms_per_step": (line.get("settled") or {}).get("ms_per_step"),
                "particles": (line.get("config") or {}).get("particles"),
                "source": "profiles/%s/bench_%s_one_gpu.json (bench.py --gpus 1 --config ... on one MI355X)" % (rnd, cfg_name)}
    return None


def dist_geometry(args, world):
    """(name, lattice block, container size) of the N-GPU workload.  BASELINE.json: 4 GPUs = config 4 (C4), 8 GPUs =
    config 5 (C5); it has no 2-GPU config, so N = 2 (and every other N) is C3 replicated N times along x.
    --config picks another config; --replicate repeats the chosen config N times along x (fixed work per GPU)."""
    import water_sandbox_amd as ws

    name = args.config or {4: "c4", 8: "c5"}.get(world, "c3")
    replicate = args.replicate or (args.config is None and world not in (1, 4, 8))
    block, size = ws.workloads.CONFIGS[name]
    copies = args.copies or (world if replicate else 1)
    if copies > 1:
        return "%sx%d" % (name, copies), (block[0] * copies, block[1], block[2]), (size[0] * copies, size[1], size[2])
    return name, block, size


def main():
    args = parse_args()
    # Only the JSON line may appear on stdout: libraries underneath (RCCL prints a version banner, c10d
    # warns) write to fd 1 as well, so fd 1 points at stderr for the whole run and the result goes to the
    # saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import numpy as np  # noqa: F401
    import torch

    import water_sandbox_amd as ws

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # WS_BENCH_FORCE_SLAB=1 runs the slab / RCCL path even with one rank (rehearsal on a one-GPU box)
    distributed = world > 1 or os.environ.get("WS_BENCH_FORCE_SLAB") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if os.environ.get("WS_BENCH_BACKEND", "nccl") != "nccl":
        local_rank %= torch.cuda.device_count()  # the gloo rehearsal: several ranks share the GPUs there are
    torch.cuda.set_device(local_rank)

    transport = None
    if distributed:
        # one process per GPU; each owns one x-slab of the domain.  Halos and migrants move as RCCL
        # send/recv with the two x-neighbours, issued by the library itself (csrc/ws_rccl.cpp): no Python in the
        # step.  torch.distributed only bootstraps (rank 0's RCCL unique id -> every rank) and provides the
        # barrier / max-over-ranks of the timing contract.
        import torch.distributed as dist

        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29531")):
            os.environ.setdefault(k, v)  # the one-rank rehearsal (WS_BENCH_FORCE_SLAB=1) without a launcher
        # WS_BENCH_BACKEND=gloo: rehearsal of the multi-rank path with several ranks on ONE GPU (RCCL refuses two
        # ranks per device); the device buffers then travel through host memory, so the numbers mean nothing
        backend = os.environ.get("WS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        ctrl = dist.new_group(backend="gloo")
        torch.cuda.set_stream(torch.cuda.Stream(device=local_rank))
        cfg_name, block, size = dist_geometry(args, world)
        base_name = cfg_name.split("x")[0]
        pos, ids, n_global, params = ws.slab.make_dist_workload(ws, block, size, args.dist, rank, world,
                                                                seed=ws.workloads.cloud_seed(base_name))
        which = os.environ.get("WS_TRANSPORT", "rccl" if backend == "nccl" else "torch")
        if which == "rccl":
            box = [ws.slab.NativeRcclTransport.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=ctrl)
            transport = ws.slab.NativeRcclTransport(box[0], rank, world, local_rank)
        else:
            transport = ws.slab.TorchDistTransport(rank, world, local_rank, data_group=None, ctrl_group=ctrl)

        def make_worker(ieee=False, profile=True):
            return ws.slab.SlabWorker(pos, ids, n_global, params, rank, world, transport, device=local_rank,
                                      stream=None if which == "rccl" else torch.cuda.current_stream().cuda_stream,
                                      profile=profile and not args.graph, ieee_division=ieee, graph=args.graph,
                                      lagged_messages=args.lagged_messages)
    else:
        cfg_name, block, size = dist_geometry(args, 1)
        base_name = cfg_name.split("x")[0]
        if cfg_name == base_name:
            pos, params = ws.workloads.make_workload(cfg_name, args.dist)
        else:  # --copies: the geometry an N-GPU --replicate run cuts into slabs, on this one GPU
            pos, _, _, params = ws.slab.make_dist_workload(ws, block, size, args.dist, 0, 1, seed=ws.workloads.cloud_seed(base_name))
        n_global = pos.shape[0]

        def make_worker(ieee=False, profile=True):
            return ws.FluidWorker(pos, params, device=local_rank, profile=profile and not args.graph, ieee_division=ieee,
                                  graph=args.graph)

    # HIP events time only the two neighbour kernels inside the timed regions (the start / stop events each launch
    # carries, on the library's stream; a slab step that splits them into early / late ranges records events around
    # the group): one of them is the dominant kernel, and timing all five launches costs several per cent of a
    # sub-millisecond step
    neighbour_mask = (1 << ws.fluid.KERNEL_IDS["force_integrate_bin"]) | (1 << ws.fluid.KERNEL_IDS["density"])

    def barrier(worker):
        worker.sync()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()

    def timed_window(worker, steps):
        """Time exactly `steps` steps bracketed by barrier + synchronize on both sides; max over ranks."""
        barrier(worker)
        worker.profile_reset()
        t0 = time.perf_counter()
        worker.run(steps)
        barrier(worker)
        elapsed = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, worker.profile()

    def one_repetition():
        """A fresh trajectory: W untimed warm-up steps, EXACTLY K timed steps, then (same handle) on to step 400 and the
        100 timed steps of the settled window."""
        worker = make_worker()
        worker.profile_select(neighbour_mask)
        worker.run(args.warmup)
        elapsed, prof = timed_window(worker, args.steps)
        rep = {"elapsed": elapsed, "prof": prof, "owned": worker.num_owned() if distributed else n_global}
        done = args.warmup + args.steps
        if not args.no_settled and done <= SETTLED_FROM:
            worker.run(SETTLED_FROM - done)
            rep["s_elapsed"], rep["s_prof"] = timed_window(worker, SETTLED_STEPS)
            rep["s_owned"] = worker.num_owned() if distributed else n_global
        return worker, rep

    reps = []
    worker = None
    for _ in range(max(1, args.reps)):
        if worker is not None:
            worker.close()
        worker, rep = one_repetition()
        reps.append(rep)

    def median_of(key):
        order = sorted(range(len(reps)), key=lambda i: reps[i][key])
        return reps[order[len(order) // 2]]

    first = median_of("elapsed")  # the repetition with the median time of the first window
    elapsed, prof, owned = first["elapsed"], first["prof"], first["owned"]
    breakdown = {k: (v[0] / max(v[1], 1)) for k, v in prof.items() if v[1]}
    settled = None
    s_prof = None
    if "s_elapsed" in reps[0]:
        sm = median_of("s_elapsed")
        s_elapsed, s_prof, s_owned = sm["s_elapsed"], sm["s_prof"], sm["s_owned"]
    # one dominant kernel for the whole run: per-launch means weighted by the windows' step counts
    dominant = dominant_kernel(*([(prof, args.steps)] + ([(s_prof, SETTLED_STEPS)] if s_prof is not None else [])))
    roof = roofline_of(prof, owned, cfg_name, args.dist, args.warmup, args.steps, distributed, dominant)
    if s_prof is not None:
        settled = {
            "window": "steps %d..%d of the same trajectory" % (SETTLED_FROM, SETTLED_FROM + SETTLED_STEPS),
            "warmup": SETTLED_FROM,
            "steps": SETTLED_STEPS,
            "ms_per_step": s_elapsed / SETTLED_STEPS * 1e3,
            "global_steps_per_s": SETTLED_STEPS / s_elapsed,
            "repetitions_ms_per_step": [r["s_elapsed"] / SETTLED_STEPS * 1e3 for r in reps],
            "kernel_ms": {k: (v[0] / max(v[1], 1)) for k, v in s_prof.items() if v[1]},
            "roofline": roofline_of(s_prof, s_owned, cfg_name, args.dist, SETTLED_FROM, SETTLED_STEPS, distributed, dominant),
        }
        if not distributed:
            settled["step_traffic"] = step_traffic(load_window_counters(cfg_name, args.dist, SETTLED_FROM, SETTLED_STEPS),
                                                   settled["ms_per_step"])
    if args.breakdown and not distributed and rank == 0:
        worker.profile_select(0xFFFFFFFF)
        worker.profile_reset()
        worker.run(20)
        worker.sync()
        for k, v in sorted(worker.profile().items(), key=lambda kv: -kv[1][0]):
            if v[1]:
                print("%-22s %9.3f ms" % (k, v[0] / v[1]), file=sys.stderr)
    stats = worker.stats()
    grid = list(worker.grid_dims()) if not distributed else None
    worker.close()

    ieee = None
    if not args.no_ieee and not distributed:
        w2 = make_worker(ieee=True)
        w2.profile_select(neighbour_mask)
        w2.run(args.warmup)
        i_elapsed, i_prof = timed_window(w2, args.steps)
        ieee = {"ms_per_step": i_elapsed / args.steps * 1e3, "global_steps_per_s": args.steps / i_elapsed,
                "kernel_ms": {k: (v[0] / max(v[1], 1)) for k, v in i_prof.items() if v[1]}}
        done = args.warmup + args.steps
        if settled is not None and done <= SETTLED_FROM:
            # the settled window with the reference's expression tree as well (one repetition): the dense state is where
            # correctly rounded division costs most
            w2.run(SETTLED_FROM - done)
            is_elapsed, is_prof = timed_window(w2, SETTLED_STEPS)
            ieee["settled"] = {"ms_per_step": is_elapsed / SETTLED_STEPS * 1e3, "global_steps_per_s": SETTLED_STEPS / is_elapsed,
                               "kernel_ms": {k: (v[0] / max(v[1], 1)) for k, v in is_prof.items() if v[1]}}
        w2.close()

    # SURVEY 8(d)'s second number: the Bevy host's per-frame pattern (src/fluid_compute.rs:478: positions read back
    # every frame) -- id-ordered positions, 12 B/particle, into the library's own page-locked double buffer, the copy of
    # frame k overlapped with step k + 1 (ws_read_positions_begin(h, NULL) / ws_step / ws_read_positions_end), over the
    # same two windows.  PCIe inclusive, so by definition never `value`.  The handle has no WS_FLAG_PROFILE (no host
    # runs its frame loop with per-kernel timing on); `ratio` = frame time / max(step alone, copy alone): 1.0 = the copy
    # hides behind the step, or the step behind the copy, completely.
    with_readback = None
    if not args.no_readback and not distributed:
        legacy = os.environ.get("WS_BENCH_RB_VARIANT") == "r03"  # A/B: round 3's leg (profiled handle, caller's registered buffer)
        w3 = make_worker(profile=legacy)
        w3.run(args.warmup)
        if legacy:
            import numpy as np

            rb_buf = np.empty((n_global, 3), np.float32)
            w3.pin_host_buffer(rb_buf)
            w3.read_positions_begin_owned = lambda: w3.read_positions_begin(rb_buf)
        w3.read_positions_begin_owned()  # (allocates the two page-locked buffers and the copy stream: not part of a frame)
        w3.read_positions_end()

        def frame_leg(steps, step_ms):
            w3.sync()
            t0 = time.perf_counter()
            for _ in range(5):  # the copy by itself, from this state (the state does not move)
                w3.read_positions_begin_owned()
                w3.read_positions_end()
            copy_ms = (time.perf_counter() - t0) / 5 * 1e3
            w3.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                w3.read_positions_begin_owned()
                w3.run(1)
                w3.read_positions_end()
            w3.sync()
            frame_ms = (time.perf_counter() - t0) / steps * 1e3
            return {"steps_per_s": 1e3 / frame_ms, "ms_per_frame": frame_ms, "copy_alone_ms": copy_ms, "step_alone_ms": step_ms,
                    "ratio_to_max_of_step_and_copy": frame_ms / max(step_ms, copy_ms),
                    "bytes_per_frame": n_global * 12, "copy_GBps": n_global * 12 / copy_ms / 1e6}

        with_readback = frame_leg(args.steps, elapsed / args.steps * 1e3)
        with_readback["pattern"] = ("per frame: ws_read_positions_begin(h, NULL) (id order, 12 B/particle, the library's "
                                    "page-locked double buffer, SDMA copy on a stream of its own priority), ws_step, "
                                    "ws_read_positions_end; same window as `value`; copy_alone = gather + copy + host round trip")
        done = args.warmup + args.steps
        if settled is not None and done <= SETTLED_FROM:
            w3.run(SETTLED_FROM - done)
            settled["with_readback"] = frame_leg(SETTLED_STEPS, settled["ms_per_step"])
        w3.close()

    # BASELINE.json's single-GPU target, driver-timed: ">= 10 M particles at >= 60 simulation steps/s on one MI355X".
    # C4 (16 777 216 particles, the smallest BASELINE config above 10 M) on this one GPU, the settled window (steps
    # 400..500 after the run-up: the slowest state of the trajectory), one repetition.
    north_star = None
    if not args.no_north_star and not distributed and cfg_name == "c3" and args.dist == "cloud" and not args.graph:
        npos, nparams = ws.workloads.make_workload("c4", "cloud")
        early_reps = []
        for rep in range(3):  # the early window on three fresh trajectories (median); the last handle goes on to step 400
            if rep:
                w4.close()
            w4 = ws.FluidWorker(npos, nparams, device=local_rank, profile=True)
            w4.profile_select(neighbour_mask)
            w4.run(5)
            w4.sync()
            t0 = time.perf_counter()
            w4.run(20)
            w4.sync()
            early_reps.append((time.perf_counter() - t0) / 20)
        early = sorted(early_reps)[1]
        ns_early_traffic = step_traffic(load_window_counters("c4", "cloud", 5, 20), early * 1e3)
        w4.run(SETTLED_FROM - 25)
        w4.sync()
        w4.profile_reset()
        t0 = time.perf_counter()
        w4.run(SETTLED_STEPS)
        w4.sync()
        late = (time.perf_counter() - t0) / SETTLED_STEPS
        ns_kernel_ms = {k: v[0] / max(v[1], 1) for k, v in w4.profile().items() if v[1]}
        w4.close()
        # measured bytes at the north-star size: this round's committed counter passes of exactly this
        # window (profiles/r05/traffic.json[c4-cloud-w400-k100]) over THIS run's step time
        ns_traffic = step_traffic(load_window_counters("c4", "cloud", SETTLED_FROM, SETTLED_STEPS), late * 1e3)
        north_star = {"target": ">= 10 M particles at >= 60 steps/s on one MI355X (BASELINE.json north_star)",
                      "workload": "C4: %d particles, uniform cloud seed 0x%X, one GPU" % (npos.shape[0], ws.workloads.cloud_seed("c4")),
                      "particles": int(npos.shape[0]),
                      "steps_per_s": 1.0 / late, "ms_per_step": late * 1e3,
                      "window": "steps %d..%d (settled: the slowest state of the trajectory), one repetition" % (SETTLED_FROM, SETTLED_FROM + SETTLED_STEPS),
                      "early": {"window": "steps 5..25, median of three fresh trajectories", "steps_per_s": 1.0 / early, "ms_per_step": early * 1e3,
                                "repetitions_ms_per_step": [e * 1e3 for e in early_reps],
                                "step_traffic": ns_early_traffic,
                                "traffic_frac_of_measured_copy_bw": ns_early_traffic["frac_of_measured_copy_bw"] if ns_early_traffic else None,
                                "meets_10M_at_60_at_40_percent_of_measured_hbm_bandwidth": bool(
                                    ns_early_traffic and npos.shape[0] >= 10_000_000 and 1.0 / early >= 60.0
                                    and ns_early_traffic["frac_of_measured_copy_bw"] >= 0.4)},
                      "meets_10M_at_60": bool(npos.shape[0] >= 10_000_000 and 1.0 / late >= 60.0),
                      "algorithmic_GBps_step": B_ALG_STEP * npos.shape[0] / late / 1e9,
                      "frac_of_measured_copy_bw_algorithmic": B_ALG_STEP * npos.shape[0] / late / 1e9 / HBM_COPY_GBS,
                      "kernel_ms": ns_kernel_ms,
                      "step_traffic": ns_traffic,
                      "traffic_frac_of_measured_copy_bw": ns_traffic["frac_of_measured_copy_bw"] if ns_traffic else None,
                      "meets_40_percent_of_measured_hbm_bandwidth": bool(ns_traffic and ns_traffic["frac_of_measured_copy_bw"] >= 0.4),
                      "bandwidth_note": "north_star asks for >= 40 % of measured HBM bandwidth: the settled step is bound by VALU issue "
                                        "and divergent gathers (DESIGN.md 5), so its counter-measured traffic stays far below that; "
                                        "the early window of the same trajectory is the bandwidth-bound one"}
        del npos

    # The reference's OWN configuration (src/fluid_compute.rs:15-17,:285: the 64 x 32 x 32 lattice, 65 536 particles -- the
    # only N the real app runs): five launches of a few microseconds each, the launch-bound regime.
    # Direct launches, the captured graph, and the frame pattern, steps 10..210 and 400..600 of the lattice's trajectory.
    ref_config = None
    if not args.no_north_star and not distributed and cfg_name == "c3" and args.dist == "cloud" and not args.graph:
        rpos, rparams = ws.workloads.make_workload("ref", "lattice")

        def ref_leg(graph):
            w5 = ws.FluidWorker(rpos, rparams, device=local_rank, profile=False, graph=graph)
            out5 = {}
            done5 = 0
            for label, start5 in (("early", 10), ("settled", SETTLED_FROM)):
                w5.run(start5 - done5)
                w5.sync()
                t0 = time.perf_counter()
                w5.run(200)
                w5.sync()
                out5[label] = (time.perf_counter() - t0) / 200 * 1e3
                done5 = start5 + 200
            if not graph:
                w5.read_positions_begin_owned()
                w5.read_positions_end()
                w5.sync()
                t0 = time.perf_counter()
                for _ in range(200):
                    w5.read_positions_begin_owned()
                    w5.run(1)
                    w5.read_positions_end()
                w5.sync()
                out5["frame"] = (time.perf_counter() - t0) / 200 * 1e3
            w5.close()
            return out5

        direct, graphed = ref_leg(False), ref_leg(True)
        ref_config = {"workload": "the reference's default: %d particles, cube_fluid(64, 32, 32) lattice, container 16 x 9 x 9" % rpos.shape[0],
                      "particles": int(rpos.shape[0]),
                      "ms_per_step": direct["early"], "steps_per_s": 1e3 / direct["early"], "window": "steps 10..210",
                      "settled": {"window": "steps 400..600", "ms_per_step": direct["settled"], "steps_per_s": 1e3 / direct["settled"]},
                      "graph_replay": {"ms_per_step": graphed["early"], "settled_ms_per_step": graphed["settled"],
                                       "vs_direct": graphed["early"] / direct["early"], "vs_direct_settled": graphed["settled"] / direct["settled"]},
                      "with_readback": {"ms_per_frame": direct["frame"], "window": "200 frames from step 600", "bytes_per_frame": int(rpos.shape[0]) * 12},
                      "regime": "launch-bound: five dependent launches per step"}

    if rank == 0:
        global_steps_per_s = args.steps / elapsed
        # `value` = simulation steps/s of the configuration this N runs -- BASELINE.json's "steps/sec @ N particles"
        # (C3 on one GPU, C4 on 4, C5 on 8): one step advances ALL particles of the job, so the global step rate IS
        # the whole-job rate.  C4 on 4 GPUs holds as many particles per GPU as C3 on one (weak scaling: the step rate
        # should stay level); `c3_equivalent_steps_per_s` = value x particles / 4 194 304 is the same throughput in
        # steps of a C3-sized share, for whoever wants value(N) / (N x value(1)).
        shares = n_global / C3_PARTICLES
        dist_name = ("uniform cloud seed 0x%X" % ws.workloads.cloud_seed(base_name)) if args.dist == "cloud" \
            else "cube_fluid lattice"
        # "scaling" is a statement about a multi-GPU run: null on one GPU.  BASELINE's C4 / C5 are FIXED problems cut into
        # 4 / 8 slabs (strong: judged against the same problem on one GPU, `same_config_one_gpu`); a replicated config
        # (N = 2: C3 twice along x) keeps the work per GPU (weak).
        scaling = None if world == 1 else ("weak" if cfg_name != base_name else "strong")
        one_gpu = same_config_one_gpu(cfg_name, args.dist) if world > 1 else None
        if one_gpu:
            one_gpu["same_window"] = one_gpu["warmup"] == args.warmup and one_gpu["steps"] == args.steps
        out = {
            "metric": "simulation steps/sec @ N particles",
            "value": global_steps_per_s,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d particles, 3D, %s, reference default parameters%s; steps %d..%d of the trajectory"
                % (cfg_name.upper(), n_global, dist_name, (", %d x-slabs (one per GPU)" % world) if distributed else "",
                   args.warmup, args.warmup + args.steps),
                "particles": n_global,
                "particles_per_gpu": n_global // world,
                "distribution": args.dist,
                "container": [params.ext_min[i] for i in range(3)] + [params.ext_max[i] for i in range(3)],
                "value_definition": "global simulation steps/s at this particle count (one step advances every particle "
                                    "of the job); c3_equivalent_steps_per_s = value x particles / 4 194 304",
                "arithmetic": "f32, every operation rounded as written (no FMA contraction); sqrt / division of the pair "
                              "terms: hardware v_sqrt_f32 / v_rcp_f32 (1 ULP) -- value_ieee: correctly rounded",
                "readback_in_timed_region": False,
                "stored_per_step": "position, velocity, cell id (+ sorted copies, densities, accept masks); the predicted "
                                   "position (pos + vel / 50 of the stored values) and the acceleration field of the "
                                   "80-byte record are materialised on demand by ws_read_particles, bit-identical",
            },
            "global_steps_per_s": global_steps_per_s,
            "c3_equivalent_steps_per_s": shares * global_steps_per_s,
            "particle_steps_per_s": global_steps_per_s * n_global,
            "algorithmic_GBps_step": B_ALG_STEP * n_global * global_steps_per_s / 1e9,
            "repetitions": {"count": len(reps), "reported": "the repetition with the median time (each window on its own)",
                            "ms_per_step": [r["elapsed"] / args.steps * 1e3 for r in reps]},
            "roofline": roof,
            "graph_replay": bool(args.graph),
            "step_traffic": None if distributed else step_traffic(
                load_window_counters(cfg_name, args.dist, args.warmup, args.steps), elapsed / args.steps * 1e3),
            "kernel_ms": breakdown,
            "settled": settled,
        }
        if settled is not None:
            settled["value"] = settled["global_steps_per_s"]
            settled["algorithmic_GBps_step"] = B_ALG_STEP * n_global * settled["global_steps_per_s"] / 1e9
        if ieee is not None:
            out["value_ieee"] = ieee["global_steps_per_s"]
            out["ieee"] = ieee
        if with_readback is not None:
            out["with_readback"] = with_readback
        if north_star is not None:
            out["north_star"] = north_star
        if ref_config is not None:
            out["ref_config"] = ref_config
        if world > 1:
            # the like-for-like yard-stick of a multi-GPU run: the SAME configuration on one GPU (it fits: C5 is 27 GB),
            # from the committed one-GPU line of that configuration.  north_star's ">= 6x at 8 GPUs" is judged on
            # speedup_vs_one_gpu_same_config of the C5 run.
            out["same_config_one_gpu"] = one_gpu
            out["speedup_vs_one_gpu_same_config"] = (
                {"first_window": one_gpu["ms_per_step"] / out["ms_per_step"] if one_gpu.get("ms_per_step") and one_gpu.get("same_window") else None,
                 "settled": one_gpu["settled_ms_per_step"] / settled["ms_per_step"] if settled and one_gpu.get("settled_ms_per_step") else None}
                if one_gpu else None)
            out["c3_equivalent_note"] = ("c3_equivalent_steps_per_s compares containers of different height (C4 / C5 settle into a "
                                         "denser floor layer than C3: +36 % / +55 % per particle on one GPU) -- not a scaling efficiency")
        if transport is not None:
            out["config"]["transport"] = type(transport).__name__
        if distributed and "migration_now" in stats:
            # what rank 0's fixed-size messages carried when the last window ended (records; 32 bytes each for a migrant
            # and for a halo-A particle, 8 for halo B): sized by every rank alike from the demand of a few steps earlier
            out["messages_rank0"] = {
                "migration_records": stats["migration_now"], "migration_MB_per_direction": stats["migration_now"] * 32 / 1e6,
                "far_records_per_destination": stats["far_now"], "far_MB_per_link": stats["far_now"] * 32 / 1e6,
                "halo_records": stats["halo_now"], "halo_MB_per_direction": stats["halo_now"] * 40 / 1e6,
                "peaks_since_load": {k: stats[k] for k in ("migration_peak", "far_peak", "halo_peak")},
                "sizing": "from the demand of a few steps earlier (WS_FLAG_LAGGED_MESSAGES)" if (args.lagged_messages or args.graph) else "exact: every message at its sender's count (default)",
                "note": "sizes in force at the end of the last timed window; DESIGN.md 6"}
        out["stats"] = stats
        if grid is not None:
            out["config"]["grid_cells"] = grid
        if not args.no_cpu_baseline and not distributed:
            out["cpu_baseline"] = cpu_baseline(pos, params, args.cpu_steps)
            if cfg_name == "c3":
                out["cpu_baseline"]["other_configs"] = cpu_baseline_other_configs()
        else:
            out["cpu_baseline"] = None
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier()
        if hasattr(transport, "close"):
            transport.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
