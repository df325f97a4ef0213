import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|MHz|NUMA node\\(s\\)' ; cat /proc/loadavg")
from oracle import oracle as orc
from harness import workload as wl
tb = wl.TablesBatch(64, 8, base_seed=42, nthreads=0)
views = []
for s in tb.scenarios:
    w = tb.world(s); views.append(orc.MapView(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d))
slot = {s: k for k, s in enumerate(tb.scenarios)}
mid = np.array([slot[s] for s in tb.scen], dtype=np.int32)
for nt in (1, 8, 32, 64, 128, 256):
    n = min(len(tb.lens), max(16, 4 * nt))
    offs = np.concatenate([[0], np.cumsum(tb.lens)])
    r = orc.optimize_batch_maps(views, mid[:n], tb.lens[:n], tb.paths[:offs[n]], nthreads=nt)
    print(f"threads {nt:4d}: {n} traj in {r['seconds']:.2f} s -> {n / r['seconds']:.1f} traj/s; thread-seconds {r['seconds_each'].sum():.1f}; per traj {r['seconds_each'].mean():.3f}")
