"""How many host threads does the CPU restatement profit from on this box?  (cpu_baseline.cores in bench.py)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import water_sandbox_amd as ws
from oracle import oracle as O
from util import oracle_from_params

print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try:
        print(p, open(p).read().strip())
    except OSError:
        pass
pos, params = ws.workloads.make_workload(sys.argv[1] if len(sys.argv) > 1 else "c3", "cloud")
orc = oracle_from_params(O, pos, params)
state = orc.particles.copy()
for t in (8, 16, 32, 64, 128, 256):
    if t > len(os.sched_getaffinity(0)):
        break
    O.set_threads(t)
    orc.set_particles(state)
    orc.step(O.SORT_FAST)
    t0 = time.perf_counter()
    for _ in range(3):
        orc.step(O.SORT_FAST)
    print("threads %3d: %.3f s/step" % (t, (time.perf_counter() - t0) / 3), flush=True)
