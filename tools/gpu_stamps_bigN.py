import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harness import workload as wl
from topay_amd import api
lo, hi = int(sys.argv[1]), int(sys.argv[2])
tb = wl.TablesBatch(512, 8, base_seed=42, nthreads=0)
probe = api.MomaTrajOptBatch(device=0)
slot = {}
for k, s in enumerate(tb.scenarios):
    w = tb.world(s)
    probe.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d, map_id=k)
    slot[s] = k
map_ids = np.array([slot[s] for s in tb.scen], dtype=np.int32)
probe.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)
N = probe.n_pieces()
sel = np.nonzero((N >= lo) & (N <= hi))[0]
offs = np.concatenate([[0], np.cumsum(tb.lens)])
lens = tb.lens[sel]
paths = np.concatenate([tb.paths[offs[b]:offs[b + 1]] for b in sel])
gpu = api.MomaTrajOptBatch(device=0, lib_path="topay_amd/lib/libtopay_hip_stamps.so")
for k, s in enumerate(tb.scenarios):
    w = tb.world(s)
    gpu.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d, map_id=k)
gpu.set_init_traj(lens, paths, map_ids=map_ids[sel])
gpu.set_trace(64)
ok = gpu.optimize(); ms, nl = gpu.last_kernel_ms()
st = gpu.stats(); ev = (st[:, 2] + st[:, 5]).astype(float)
names = ["fill", "LU", "subst", "jerk", "sweep1", "between", "sweep2", "adjoint", "assemble", "lbfgs", "(twoloop)", "(s1 rounds)", "(body)", "(mani)", "(adj sweeps)", "(s1 merged round)"]
tot = np.zeros(16)
for b in range(len(lens)):
    tot += gpu.get_trace(b)[8:24].view(np.int64)[:16].astype(float)
print("B", len(lens), "N mean", gpu.n_pieces().mean(), "kernel %.1f ms" % ms, "mean bound", st[:, 7].sum() / max(1, st[:, 4].sum()))
for n, c in zip(names, tot / ev.sum()):
    print("   %-14s %10.0f cycles/eval" % (n, c))
print("   total %.0f" % (tot[:10].sum() / ev.sum()))
