import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import set_map
from topay_amd import api
from oracle import oracle as orc
from harness import workload as wl
w, lens, paths, scen = wl.cuboids_batch(3, 2)
m = orc.MapView(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d)
offs = np.concatenate([[0], np.cumsum(lens)])
gpu = api.MomaTrajOptBatch(device=0)
set_map(gpu, w)
gpu.set_init_traj(lens, paths)
for b in (0, 3, 5):
    o = orc.Oracle(m); n = o.set_init_traj(paths[offs[b]:offs[b + 1]])
    x = o.get_x()
    for lam, rho in (([0.0, 0.0], [1e4, 1e4]), ([0.3, -0.2], [1e4, 2e4]), ([0.0, 0.0], [1.0, 1.0])):
        o.set_alm(lam, rho)
        f, g = o.eval(2, x); eo = o.final_xy_error()
        for wv in (1, 2):
            fg, gg, eg = gpu.eval(2, b, x, lam, rho, waves=wv)
            bad = np.nonzero(~(np.abs(gg - g) <= 1e-9 * np.abs(g).max()))[0]
            print("cand", b, "N", o.N, "lam", lam, "rho", rho, "waves", wv, "f dev %.6f ref %.6f" % (fg, f), "xyerr dev", eg, "ref", eo, "bad g idx", bad[:12], len(bad), flush=True)
