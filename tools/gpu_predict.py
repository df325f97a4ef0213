"""Is the number of stage-2 evaluations predictable after stage 1?  (launch-order experiments)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from topay_amd import api
from harness import workload as wl
tb = wl.TablesBatch(512, 8, base_seed=42, nthreads=0)
opt = api.MomaTrajOptBatch(device=0)
slot = {}
for k, s in enumerate(tb.scenarios):
    w = tb.world(s)
    opt.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d, map_id=k)
    slot[s] = k
map_ids = np.array([slot[s] for s in tb.scen], dtype=np.int32)
opt.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)
opt.set_trace(96)
ok = opt.optimize()
st = opt.stats(); N = opt.n_pieces()
m = (N > 0) & (st[:, 0] > 0) & (st[:, 2] < 90)
ev2 = st[:, 5].astype(float)
f_first = np.zeros(len(N)); f_s1 = np.zeros(len(N)); f_5 = np.zeros(len(N))
for b in np.nonzero(m)[0]:
    tr = opt.get_trace(int(b))
    s1 = st[b, 2]
    f_s1[b] = tr[s1 - 1]; f_first[b] = tr[s1]; f_5[b] = tr[min(s1 + 5, 95)]
def corr(a, b): return np.corrcoef(a, b)[0, 1]
lev = np.log(ev2[m] + 1)
print("n", m.sum())
for name, v in (("N", N[m].astype(float)), ("path states", tb.lens[m].astype(float)), ("log stage-1 final cost", np.log(f_s1[m])),
                ("log first stage-2 cost", np.log(f_first[m])), ("log(first s2 / s1 cost)", np.log(f_first[m] / f_s1[m])),
                ("log cost after 5 s2 evals", np.log(np.abs(f_5[m]) + 1)), ("stage-1 iterations", st[m, 1].astype(float))):
    print(f"corr(log evals2, {name:28s}) = {corr(lev, v):+.3f}   corr(time proxy N*evals, .) = {corr(np.log(N[m]*ev2[m]+1), v):+.3f}")
# multi-feature least squares
X = np.stack([np.ones(m.sum()), N[m], np.log(f_first[m]), np.log(f_first[m] / f_s1[m]), tb.lens[m]], axis=1)
coef, *_ = np.linalg.lstsq(X, np.log(N[m] * ev2[m] + 1), rcond=None)
pred = X @ coef
print("multi-feature fit corr with log(N*evals):", corr(pred, np.log(N[m] * ev2[m] + 1)))
