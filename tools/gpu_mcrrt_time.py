"""Time of topay_mcrrt_plan on a benchmark-sized batch (1024 tables scenarios x 8 chassis paths) and the CPU restatement's
time on a sample of it (one thread)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import set_map
from topay_amd import api
from harness import workload as wl

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tb = wl.TablesBatch(S, 8, base_seed=42, nthreads=0)
opt = api.MomaTrajOptBatch(device=0)
slot = {s: k for k, s in enumerate(tb.scenarios)}
for s_ in tb.scenarios:
    set_map(opt, tb.world(s_), map_id=slot[s_])
offs = np.concatenate([[0], np.cumsum(tb.lens)])
n = len(tb.lens)
car = np.c_[tb.paths[:, :3], tb.dts]
start = tb.paths[offs[:-1]]
end = tb.paths[offs[1:] - 1]
mid = np.array([slot[s] for s in tb.scen], dtype=np.int32)
prm = opt.mcrrt_params(seed=42)
for rep in range(3):
    t0 = time.perf_counter()
    wbs, stats, cmax = opt.mcrrt_plan(tb.lens, car, start, end, prm, map_ids=mid)
    dt = time.perf_counter() - t0
    print(f"run {rep}: {n} searches in {dt * 1e3:.1f} ms = {n / dt:.0f} searches/s; found {np.mean(stats[:, 0] == 1):.3f}, pool full {np.sum(stats[:, 0] == -1)}, "
          f"mean iterations {stats[:, 2].mean():.1f}, mean nodes {stats[:, 1].mean():.1f} (max {stats[:, 1].max()}), whole-body checks {stats[:, 7].astype(np.int64).sum() / 1e6:.1f} M", flush=True)
k = min(n, 256)
t0 = time.perf_counter()
for b in range(k):
    wl.mcrrt_plan(tb.world(int(tb.scen[b])), start[b], end[b], car[offs[b]:offs[b + 1]], wl.McrrtParams(seed=42), inst=b, want_nodes=False)
dt = time.perf_counter() - t0
print(f"CPU restatement, one thread: {k} searches in {dt:.2f} s = {k / dt:.0f} searches/s")
# the 2-D jump-point search (one thread per search): one (start, goal) pair per scenario
first = np.array([int(np.nonzero(tb.scen == s_)[0][0]) for s_ in tb.scenarios])
st2, en2 = tb.paths[offs[first], :2], tb.paths[offs[first + 1] - 1, :2]
mid2 = np.array([slot[tb.scen[b]] for b in first], dtype=np.int32)
for rep in range(2):
    t0 = time.perf_counter()
    jp, jst, jln = opt.plan2d_jps(st2, en2, 0.5, map_ids=mid2)
    dt = time.perf_counter() - t0
    print(f"JPS run {rep}: {len(first)} searches in {dt * 1e3:.1f} ms = {len(first) / dt:.0f} searches/s; paths {int((jln > 0).sum())}, expanded nodes mean {jst[:, 0].mean():.0f} max {jst[:, 0].max()}", flush=True)
k2 = min(len(first), 256)
t0 = time.perf_counter()
for i in range(k2):
    wl.plan2d_jps(tb.world(int(tb.scen[first[i]])), st2[i], en2[i], 0.5)
dt = time.perf_counter() - t0
print(f"JPS CPU restatement, one thread: {k2} searches in {dt:.2f} s = {k2 / dt:.0f} searches/s")
