"""Wide per-evaluation parity sweep HIP vs oracle: every N bucket, three kinds of points (initial guess, random
perturbation, rare-path trigger), both stages, multi-map."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from topay_amd import api
from harness import workload as wl
tb = wl.TablesBatch(256, 8, base_seed=7000, nthreads=0)
opt = api.MomaTrajOptBatch(device=0)
slot = {}
views = {}
for k, s in enumerate(tb.scenarios):
    w = tb.world(s)
    opt.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d, map_id=k)
    slot[s] = k
    views[s] = orc.MapView(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d)
map_ids = np.array([slot[s] for s in tb.scen], dtype=np.int32)
opt.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)
N = opt.n_pieces()
offs = np.concatenate([[0], np.cumsum(tb.lens)])
rng = np.random.default_rng(1)
worst_f, worst_g, cnt = 0.0, 0.0, 0
for lo, hi in ((3, 7), (8, 10), (11, 13), (14, 16), (17, 21), (22, 26), (27, 32)):
    idx = np.nonzero((N >= lo) & (N <= hi))[0]
    if len(idx) == 0:
        continue
    for b in rng.choice(idx, size=min(25, len(idx)), replace=False):
        b = int(b)
        o = orc.Oracle(views[int(tb.scen[b])])
        n = o.set_init_traj(tb.paths[offs[b]:offs[b + 1]])
        Nb = o.N
        for trial in range(3):
            x = o.get_x().copy()
            if trial == 1:
                x += 0.08 * rng.standard_normal(n)
            if trial == 2:
                x[:Nb] -= 1.5
                x[Nb - 1] += 2.2
                x[3 * Nb - 1:] += np.tile([0.0, 1.5, 0.0, 2.4, 0.0, 1.9, 0.0], Nb - 1)
            lam, rho = rng.uniform(-1, 1, 2), np.array([1e4, 3e5])
            o.set_alm(lam, rho)
            for stage in (1, 2):
                f, g = o.eval(stage, x)
                fg, gg, _ = opt.eval(stage, b, x, lam, rho)
                if not np.isfinite(f):
                    continue
                ef = abs(f - fg) / abs(f)
                eg = np.abs(g - gg).max() / max(np.abs(g).max(), 1e-300)
                worst_f, worst_g, cnt = max(worst_f, ef), max(worst_g, eg), cnt + 1
                if ef > 1e-11 or eg > 1e-10:
                    print("MISMATCH b", b, "N", Nb, "trial", trial, "stage", stage, ef, eg)
print(f"{cnt} evaluations compared; worst relative error f {worst_f:.2e}, g {worst_g:.2e}")
