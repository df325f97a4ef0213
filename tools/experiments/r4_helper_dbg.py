"""Which candidates differ between the default and the helper-wave kernels (debugging aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from conftest import serpentine_path
from harness import workload as wl
from topay_amd import api
w, lens0, paths0, scen = wl.cuboids_batch(3, 2)
offs = np.concatenate([[0], np.cumsum(lens0)])
paths = [paths0[offs[b]:offs[b + 1]] for b in range(len(lens0))] + [serpentine_path(L) for L in (14.0, 19.5, 27.0, 34.0)]
lens = np.array([len(p) for p in paths], dtype=np.int32)
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 0
res = []
for mode in (0, 2, 2):
    p = api.default_params()
    if cap:
        p.s1_lbfgs.max_iterations = cap; p.s2_lbfgs.max_iterations = cap; p.alm_max_outer = 1
    o = api.MomaTrajOptBatch(params=p, device=0)
    o.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d, 0)
    o.set_init_traj(lens, np.concatenate(paths))
    o.set_latency_mode(mode)
    o.set_trace(4096)
    ok = o.optimize()
    res.append((ok, o.stats(), np.nan_to_num(o.traj_cost), [o.get_trace(b) for b in range(len(paths))], o.n_pieces()))
a = res[0]
for name, h in (("helper", res[1]), ("helper again", res[2])):
    for b in range(len(paths)):
        same = (a[1][b] == h[1][b]).all() and a[2][b] == h[2][b]
        ta, th = a[3][b], h[3][b]
        nd = np.nonzero(ta != th)[0]
        print(name, "b", b, "N", int(a[4][b]), "same" if same else "DIFF", "first differing evaluation", (int(nd[0]) if len(nd) else None), "evals", a[1][b][2] + a[1][b][5], h[1][b][2] + h[1][b][5])
