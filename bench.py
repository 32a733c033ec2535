#!/usr/bin/env python3
"""bench.py -- simulation steps/s of the SPH fluid step on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3] [--dist cloud|lattice]

One "step" = one full pass of the hot path (cell sort -> density -> force -> integrate) over all
particles, inputs resident in HBM before the timed region, no host readback inside it.  Rank 0
prints ONE JSON line.  For N > 1 the driver launches this under torch.distributed.run (one rank per
GPU); see DESIGN.md "Multi-GPU".
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c3", help="c1..c5 | ref (BASELINE.md section 2)")
    ap.add_argument("--dist", default="cloud", choices=["cloud", "lattice"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=12,
                    help="oracle steps timed for cpu_baseline (~10 s of CPU work at C3 on 16 threads)")
    ap.add_argument("--breakdown", action="store_true", help="also print a per-kernel table to stderr")
    return ap.parse_args()


def cpu_baseline(pos, params, steps):
    """The CPU restatement of the reference WGSL (oracle, 'fast' sort mode, OpenMP) timed on
    this host on the SAME workload for a few steps.  Reported, never the target."""
    from oracle import oracle as O

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import oracle_from_params

    orc = oracle_from_params(O, pos, params)
    orc.step(O.SORT_FAST)  # warm-up (page faults, thread team)
    t0 = time.perf_counter()
    for _ in range(steps):
        orc.step(O.SORT_FAST)
    dt = time.perf_counter() - t0
    return {
        "value": steps / dt,
        "unit": "steps/s",
        "cores": O.default_threads(),
        "kind": "port",
        "sample": "%d full steps of the same %d-particle workload after 1 warm-up step (oracle, fast sort mode)"
        % (steps, orc.n),
    }


KERNEL_LABEL = {
    "density": "density (K4 update_density: radius sweep + accept masks)",
    "force_integrate_bin": "force_integrate_bin (K5 update_pressure_force + K6 integrate + next K1 binning)",
}


def load_traffic(config, dist, kernel):
    """HBM bytes per launch of `kernel` from committed rocprofv3 --pmc passes
    (profiles/traffic.json), or None if no measurement for this workload is committed."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return t.get("%s-%s" % (config, dist), {}).get("bytes_per_launch", {}).get(kernel)
    except (OSError, ValueError):
        return None


def main():
    args = parse_args()
    # Only the JSON line may appear on stdout: libraries underneath (RCCL prints a version banner, c10d
    # warns) write to fd 1 as well, so fd 1 points at stderr for the whole run and the result goes to the
    # saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch

    import water_sandbox_amd as ws

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # WS_BENCH_FORCE_SLAB=1 runs the slab / torch.distributed path even with one rank (rehearsal on a one-GPU box)
    distributed = world > 1 or os.environ.get("WS_BENCH_FORCE_SLAB") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if os.environ.get("WS_BENCH_BACKEND", "nccl") != "nccl":
        local_rank %= torch.cuda.device_count()  # the gloo rehearsal: several ranks share the GPUs there are
    torch.cuda.set_device(local_rank)

    if distributed:
        # one process per GPU; each owns one x-slab of the (world x wider) domain.  Halos and migrants
        # move through torch.distributed: backend "nccl" = RCCL over xGMI for device buffers, a gloo
        # group for the few control words per step (DESIGN.md "Multi-GPU").
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
        # the library enqueues on the stream RCCL's point-to-point calls are ordered on: an explicit,
        # non-default torch stream made current for the whole run (the default stream's handle is 0,
        # which ws_device_cfg.stream reads as "create your own")
        torch.cuda.set_stream(torch.cuda.Stream(device=local_rank))
        pos, ids, n_global, params = ws.slab.make_dist_workload(ws, args.config, args.dist, rank, world)
        if os.environ.get("WS_TRANSPORT", "torch") == "rccl":
            # the library's own RCCL transport: torch.distributed only carries rank 0's unique id to the others
            box = [ws.slab.NativeRcclTransport.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=ctrl)
            transport = ws.slab.NativeRcclTransport(box[0], rank, world, local_rank)
        else:
            transport = ws.slab.TorchDistTransport(rank, world, local_rank, data_group=None, ctrl_group=ctrl)
        worker = ws.slab.SlabWorker(pos, ids, n_global, params, rank, world, transport, device=local_rank,
                                    stream=torch.cuda.current_stream().cuda_stream, profile=True)
        n_rank = n_global // world  # particles per rank at t = 0 (the single-GPU config's count)
    else:
        pos, params = ws.workloads.make_workload(args.config, args.dist)
        n_global = n_rank = pos.shape[0]
        worker = ws.FluidWorker(pos, params, device=local_rank, profile=True)
    # HIP events bracket only the two neighbour kernels inside the timed region (4 records per step, on the
    # library's stream): one of them is the dominant kernel, and bracketing all five launches costs several
    # per cent of a sub-millisecond step
    worker.profile_select((1 << ws.fluid.KERNEL_IDS["force_integrate_bin"]) | (1 << ws.fluid.KERNEL_IDS["density"]))

    def barrier():
        worker.sync()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()

    worker.run(args.warmup)
    barrier()
    worker.profile_reset()
    t0 = time.perf_counter()
    worker.run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = worker.profile()
    dominant = max(("density", "force_integrate_bin"), key=lambda k: prof[k][0])
    force_ms, force_cnt = prof[dominant]
    breakdown = {k: (v[0] / max(v[1], 1)) for k, v in prof.items() if v[1]}
    owned = worker.num_owned() if distributed else n_rank
    if args.breakdown and not distributed and rank == 0:
        # per-kernel table from a separate short pass AFTER the timed region (later steps of the trajectory)
        worker.profile_select(0xFFFFFFFF)
        worker.profile_reset()
        worker.run(min(args.steps, 20))
        worker.sync()
        breakdown = {k: (v[0] / max(v[1], 1)) for k, v in worker.profile().items() if v[1]}

    if rank == 0:
        global_steps_per_s = args.steps / elapsed
        # unit of work = one step of one rank's 4 194 304-particle (config-sized) share; all ranks
        # together process `world` of them per global step (weak scaling), so the whole-job value is
        # world x global steps/s.  At world = 1 this is plain simulation steps/s.
        value = world * global_steps_per_s
        force_avg_s = force_ms / max(force_cnt, 1) * 1e-3
        alg_bytes = KERNEL_ALG_BYTES[dominant] * owned
        achieved = alg_bytes / force_avg_s / 1e9
        dist_name = ("uniform cloud seed 0x%X" % ws.workloads.cloud_seed(args.config)) if args.dist == "cloud" \
            else "cube_fluid lattice"
        out = {
            "metric": "simulation steps/sec @ N particles",
            "value": value,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s%s: %d particles, 3D, %s, reference default parameters; steps %d..%d of the trajectory"
                % (args.config.upper(), (" x%d along x (one x-slab per GPU)" % world) if distributed else "", n_global,
                   dist_name, args.warmup, args.warmup + args.steps),
                "particles": n_global,
                "particles_per_gpu": n_rank,
                "distribution": args.dist,
                "container": [params.ext_min[i] for i in range(3)] + [params.ext_max[i] for i in range(3)],
                "value_definition": "n_gpus x global simulation steps/s (each GPU steps a %d-particle share)" % n_rank,
                "readback_in_timed_region": False,
            },
            "global_steps_per_s": global_steps_per_s,
            "particle_steps_per_s": global_steps_per_s * n_global,
            "algorithmic_GBps_step": B_ALG_STEP * n_global * global_steps_per_s / 1e9,
            "roofline": {
                "kernel": KERNEL_LABEL[dominant],
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": load_traffic(args.config, args.dist, dominant) if not distributed else None,
                "alg_bytes_per_launch": alg_bytes,
                "avg_launch_ms": force_avg_s * 1e3,
                "launches_timed": force_cnt,
            },
            "kernel_ms": breakdown,
        }
        if not distributed:
            out["config"]["grid_cells"] = list(worker.grid_dims())
            out["stats"] = worker.stats()
        if not args.no_cpu_baseline and not distributed:
            out["cpu_baseline"] = cpu_baseline(pos, params, args.cpu_steps)
        else:
            out["cpu_baseline"] = None
        if args.breakdown:
            for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1]):
                print("%-22s %9.3f ms" % (k, v), file=sys.stderr)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    worker.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
