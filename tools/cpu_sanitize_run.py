import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topay_amd import api
from harness import workload as wl
LIB = sys.argv[1]
w, lens, paths, scen = wl.cuboids_batch(3, 2)
p = api.default_params(api.load(LIB))
p.s2_lbfgs.max_iterations = 25
p.alm_max_outer = 2
opt = api.MomaTrajOptBatch(params=p, lib_path=LIB)
opt.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d, 0)
opt.optimizeTraj(lens, paths)
print("solve ok", opt.stats()[:, :3].tolist())
print("gate", opt.check_feasible())
# tables scenario with more candidates (persistent loop) and a map built by the EDT kernels
w1, _, _, lens1, paths1 = wl.tables_scenario(0, 12)
o2 = api.MomaTrajOptBatch(params=p, lib_path=LIB)
o2.build_esdf(w1.origin, w1.res, w1.dims, w1.min_b, w1.max_b, w1.occ2d, w1.occ3d)
o2.optimizeTraj(lens1, paths1)
print("tables ok", o2.stats()[:, :3].tolist())
st = np.zeros((4, 10)); st[:, 0] = np.linspace(-3, 3, 4)
print("wb", o2.whole_body_collision(st))
